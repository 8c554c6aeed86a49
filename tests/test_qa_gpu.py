"""Video-QA half of SURVEY.md §8 (f4) on the MI355X engine: ``BLIP2_MR(task="...QA...")`` — forward_QA (uniform sampling and the localizer
variant), the answerer's LoRA gradients, videoQA_generate — through the C ABI, against the reference-generated golden (mr_tiny_qa.npz: the
reference's own forward_QA at B = 0 LoRA) and the CPU oracle (non-zero LoRA, autograd).  Reference: blip2_mr.py:309-431, 948-1314."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import TINY_CFG, check, load_golden, golden_state_dict, relerr  # noqa: E402


def _peft_both(sd, with_lora=True, lora_std=None):
    """peft names (+ seeded non-zero LoRA) for BOTH T5s of the QA variants: t5_model.* (localizer) and answerer_model.*"""
    from test_model_gpu import _peft_sd
    loc = _peft_sd({k: v for k, v in sd.items() if not k.startswith("answerer_model.")}, with_lora=with_lora, lora_std=lora_std)
    ans_in = {"t5_model." + k[len("answerer_model."):]: v for k, v in sd.items() if k.startswith("answerer_model.")}
    ans = _peft_sd(ans_in, with_lora=with_lora, seed=1, lora_std=lora_std)
    out = dict(loc)
    for k, v in ans.items():
        if k.startswith("t5_model."):
            k2 = "answerer_model." + k[len("t5_model."):]
            if "lora_" in k2:      # (other seeds than the localizer's adapters: the key names differ)
                from weights import seeded_array
                v = torch.from_numpy(seeded_array(k2, tuple(v.shape), std=lora_std))
            out[k2] = v
    return out


def _setup(task, with_lora=False, lora_std=None, mean=False):
    import lavis  # noqa: F401
    from lavis.common.registry import registry
    from mrblip.engine import EngineConfig
    from mrblip.tokenizer import FixtureTokenizer

    g = load_golden("mr_tiny_qa")
    s = g["strings"]
    sd = golden_state_dict(g)
    sdl = _peft_both(sd, with_lora=with_lora, lora_std=lora_std) if with_lora else dict(sd)
    tok = FixtureTokenizer()
    cls = registry.get_model_class("blip2_mr")
    model = cls(img_size=56, num_query_token=8, engine_config=EngineConfig.tiny(), weights=dict(sdl), tokenizer=tok, interleave_data=True,
                task=task, input_time_format="seconds_integers", frame_token_aggregation="mean" if mean else None, seed=42,
                num_frames_for_answer=s["nfa"])
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                   qa_input=s["qa_input"], qa_output=s["qa_output"], question_id=s["question_id"],
                   query_prompt=["Query: " + q for q in s["qa_input"]], task_prompt=["Relevant windows: "] * 2, video_prompt_end=["<extra_id_0>"] * 2)
    return model, g, s, samples, sdl, tok


def test_forward_qa_uniform_against_the_reference_golden():
    """the reference's own forward_QA (peft stubbed: LoRA at B = 0) — loss, encoder input, labels; frame selection bit-exact"""
    model, g, s, samples, sdl, tok = _setup("qformer_freeze_lora_QA")
    model.eval()
    assert model.is_qa and not model.use_localizer and model.train_engine is model.answerer
    names = [n for n, _ in model.named_parameters()]
    assert names and all(n.startswith("answerer_model.base_model.model.") and ".lora_" in n for n in names)      # only the answerer's LoRA trains
    with torch.no_grad():
        out = model(samples)
    ans = model.answerer
    B, S = g["inputs_atts"].shape
    d = ans.cfg.d_model
    emb = ans.ws["inputs_embeds"].view(B, S, d).cpu()
    check("qa_tiny.inputs_embeds vs reference-fp32", relerr(emb, g["inputs_embs"]), 6e-3)
    assert model.last_relevant_moments == [[0, 45.0], [0, 205.0]]
    check("qa_tiny.loss (uniform sampling) vs reference-fp32 (rel)", abs(out["loss"].item() - float(g["loss_uniform"])) / float(g["loss_uniform"]), 6e-4)
    logits = ans.ws["d_logits"].view(B, -1, 32128).cpu()
    check("qa_tiny.logits vs reference-fp32", relerr(logits[..., ::64], g["logits_sub"]), 1.2e-2)
    check("qa_tiny.logits_lse vs reference-fp32", relerr(torch.logsumexp(logits, -1), g["logits_lse"]), 2e-5)
    # the frames the answerer saw: extract_frames of the model == the reference's indices
    fr = model.extract_frames(samples, [[0, 45.0], [0, 205.0]], s["nfa"])
    for b in range(2):
        assert [int((samples["video"][b] - fr[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(s["nfa"])] == g["idx_uniform"][b].tolist()
    m, fr = model.get_relevant_frames(samples, s["preds"], s["nfa"])
    assert [[float(x) for x in mm] for mm in m] == s["moments"]
    for b in range(2):
        assert [int((samples["video"][b] - fr[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(s["nfa"])] == g["idx_loc"][b].tolist()
    m2, fr2 = model.get_relevant_frames(samples, s["preds2"], s["nfa"])
    assert [[float(x) for x in mm] for mm in m2] == s["moments2"]
    for b in range(2):
        assert [int((samples["video"][b] - fr2[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(s["nfa"])] == g["idx_loc2"][b].tolist()


def test_forward_qa_with_localizer_windows_against_the_reference_golden():
    """stage 1 = the localizer's generate(): here its answer is GIVEN (the reference golden was produced the same way); stage 2 must see the
    frames get_relevant_frames picks for it"""
    model, g, s, samples, sdl, tok = _setup("qformer_freeze_lora_QA_with_localizer")
    model.eval()
    assert model.use_localizer
    real_generate = model.generate
    model.generate = lambda smp, **kw: {"prediction": s["preds"]}
    with torch.no_grad():
        out = model(samples)
    check("qa_tiny.loss (localizer windows) vs reference-fp32 (rel)", abs(out["loss"].item() - float(g["loss_loc"])) / float(g["loss_loc"]), 6e-4)
    assert [[float(x) for x in mm] for mm in model.last_relevant_moments] == s["moments"]
    # ... and with the REAL stage 1: the localizer's beam search runs on the HIP decoder and its answer goes through post_process /
    # moment_str_to_list / extract_frames (whatever the random localizer says, the loss is finite and the windows lie inside the clip)
    model.generate = real_generate
    with torch.no_grad():
        out2 = model(samples)
    assert torch.isfinite(out2["loss"]).item() and len(model.last_relevant_moments) == 2
    for (a, b), dur in zip(model.last_relevant_moments, samples["duration"].tolist()):
        assert b <= round(dur) + 1e-6           # (an end beyond the clip is clipped: blip2_mr.py:1115-1116)


def test_answerer_lora_gradients_against_the_oracle():
    """non-zero LoRA on the answerer (and on the frozen localizer): loss and every dA / dB of the answerer against the oracle's autograd; nothing
    else receives a gradient (the frame path runs without one, blip2_mr.py:325-363)"""
    from oracle import mrblip_oracle as O
    model, g, s, samples, sdl, tok = _setup("qformer_freeze_lora_QA", with_lora=True)
    model.eval()
    for k, v in sdl.items():
        v.requires_grad_(k.startswith("answerer_model.") and "lora_" in k)
    out = model(samples)
    out["loss"].backward()
    orc = O.Oracle(sdl, TINY_CFG, emu_bf16=True, lora=dict(r=8, alpha=8))
    ref = orc.forward_qa(tok, samples, s["nfa"])
    ref["loss"].backward()
    check("qa_tiny.lora!=0: loss vs emu-oracle (rel)", abs(out["loss"].item() - ref["loss"].item()) / abs(ref["loss"].item()), 3e-4)
    ans = model.answerer
    worst = 0.0
    num = den = 0.0
    for a in ans.adapters:
        base = "answerer_model.base_model.model." + a.name
        ga, gb = sdl[base + ".lora_A.default.weight"].grad, sdl[base + ".lora_B.default.weight"].grad
        ea, eb = relerr(a.dA.cpu(), ga), relerr(a.dBt.cpu().t(), gb)
        worst = max(worst, ea, eb)
        num += float((a.dA.cpu() - ga).pow(2).sum() + (a.dBt.cpu().t() - gb).pow(2).sum())
        den += float(ga.pow(2).sum() + gb.pow(2).sum())
    check("qa_tiny.lora!=0: all answerer LoRA gradients (flat) vs emu-oracle autograd", (num / den) ** 0.5, 3e-2)
    check("qa_tiny.lora!=0: worst answerer adapter dA / dB vs emu-oracle autograd", worst, 6e-2)
    # autograd side of the model: the named parameters' .grad alias the flat gradient
    named = dict(model.named_parameters())
    a0 = ans.adapters[0]
    k0 = "answerer_model.base_model.model." + a0.name + ".lora_A.default.weight"
    assert named[k0].grad is not None and torch.equal(named[k0].grad, a0.dA)
    # the localizer engine received no gradient at all, and t5_proj / ln_vision of the answerer's buffer stay untouched
    assert float(model.engine.grad.abs().sum()) == 0.0 and float(ans.grad[ans.n_lora:].abs().sum()) == 0.0
    # checkpoint surface: the answerer's LoRA under answerer_model.*, t5_proj / ln_vision, no localizer LoRA (frozen in the reference)
    sd_out = model.state_dict()
    assert k0 in sd_out and "t5_proj.weight" in sd_out and not any(k.startswith("t5_model.") for k in sd_out)
    # a moment-retrieval checkpoint (t5_model.* LoRA) loads into the LOCALIZER
    la = model.engine.adapters[0]
    mr_ck = {"t5_model.base_model.model." + la.name + ".lora_A.default.weight": torch.full_like(la.A, 0.25).cpu()}
    msg = model.load_state_dict(mr_ck)
    assert not msg.unexpected_keys and float(la.A.mean()) == 0.25


def test_video_qa_generate_against_the_oracle():
    """videoQA_generate (uniform sampling): two greedy steps of the answerer on the HIP decoder, the option argmax of step 1 — against the
    oracle's restatement of HF's greedy search with min_length = 8 (HF's generate is not available on the reference's T5 class in this image)"""
    from oracle import mrblip_oracle as O
    model, g, s, samples, sdl, tok = _setup("qformer_freeze_lora_QA", with_lora=True)
    model.eval()
    out = model.videoQA_generate(samples)
    orc = O.Oracle(sdl, TINY_CFG, emu_bf16=True, lora=dict(r=8, alpha=8))
    # (the reference embeds the question with the LOCALIZER's table, which its QA constructor has cast to bf16: blip2_mr.py:206-209, 1262)
    emb_key = next(k for k in sdl if k.startswith("t5_model.") and k.endswith("shared.weight"))
    sd2 = dict(sdl)
    sd2[emb_key] = sdl[emb_key].detach().bfloat16().float()
    orc = O.Oracle(sd2, TINY_CFG, emu_bf16=True, lora=dict(r=8, alpha=8))
    pred, opt, first = orc.qa_answer(tok, samples, s["nfa"])
    assert model.last_first_token.tolist() == first.tolist()
    check("qa_tiny.generate: option logits of the second step vs emu-oracle", relerr(model.last_option_logits, opt), 1.5e-2)
    margin = (opt.topk(2, -1).values[:, 0] - opt.topk(2, -1).values[:, 1]).min().item()
    if margin > 5e-2:
        assert out["output_text"] == pred
    assert out["answer"] == s["qa_output"] and out["qid"] == s["question_id"] and out["relevant_moments"] == [[[0, 45.0], [0, 205.0]]]
