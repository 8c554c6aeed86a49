"""Parity at REAL depth / width and the BASELINE.json configurations the tiny fixtures do not reach (VERDICT r1 "configs_untested"):

* C1 (configs[0]): full ViT-g/14 (39 blocks x 1408) + bert-base Q-Former (12 layers, 32 queries) + Flan-T5-base-sized T5 (12 + 12 layers),
  4 frames, batch 1 — the HIP step against the REFERENCE's own fp32 outputs (tests/golden/mr_c1.npz, produced by
  tests/golden/make_golden_c1.py importing /root/reference) and against the bf16-emulating oracle.  Shows how the bf16 rounding grows
  through 39 + 12 + 24 real layers.
* C3 (B = 4 per GPU, QVH, XL): one 4-clip step == the accumulation of four 1-clip steps.
* C5 (ActivityNet, T = 120, S ~ 4000): a full-size step runs, loss finite and at the random-init level, gradients finite.
* C4 (Charades, 32 -> 1 mean pool, T = 20) at full width.

Every measured error goes to gpurun_out/parity_errors.json (tests/util.py: check / record).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import check, load_golden, record, relerr  # noqa: E402

C1_CFG = dict(
    vit=dict(embed_dim=1408, depth=39, num_heads=16, img=224, patch=14),
    qf=dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=12, cross_attention_freq=2, num_query_token=32),
    t5=dict(d_model=768, d_kv=64, d_ff=2048, num_layers=12, num_decoder_layers=12, num_heads=12, vocab_size=32128, num_buckets=32,
            max_distance=128, eps=1e-6),
)


def _c1_samples(g):
    from weights import seeded_array

    s = g["strings"]
    video = torch.from_numpy(seeded_array("c1.input.video", (1, 4, 3, 224, 224), std=1.0, fast=True))
    return dict(video=video, timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                query_prompt=s["query_prompt"], task_prompt=s["task_prompt"], video_prompt_end=s["video_prompt_end"],
                relevant_windows=s["relevant_windows"])


@pytest.mark.parametrize("vit_operands", ["bf16", "fp16"])
def test_c1_real_depth_against_reference_and_oracle(vit_operands):
    """vit_operands = "fp16": the frozen ViT on IEEE fp16 operands like the reference's GPU path (EngineConfig.vit_operands; measured and not
    the default: see there) — same assertions, its errors logged under their own names."""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from oracle import mrblip_oracle as O

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12, vit_operands=vit_operands)
    dev = torch.device("cuda:0")
    eng = MrBlipEngine(cfg, StateDictSource(sd), dev)   # LoRA: peft default init (B = 0): the forward equals the reference's LoRA-free run
    eng.training = False
    assert eng.vit_dtype == (torch.float16 if vit_operands == "fp16" else torch.bfloat16)
    import util
    _check = util.check

    def check(name, value, tol):   # noqa: F811  (fp16-ViT rows get their own names in the error log)
        return _check(name.replace("c1.", "c1[vit fp16]." if vit_operands == "fp16" else "c1.", 1), value, tol)

    lay = P.build_layout(tok, samples, repl, 32, T=4)
    assert lay.S == g["inputs_atts"].shape[1]
    assert np.array_equal(lay.attention_mask.numpy(), g["inputs_atts"]) and np.array_equal(lay.labels.numpy(), g["labels"])   # integer work: bit-exact
    eng.zero_grad()
    loss = eng.forward_backward(samples["video"].to(dev), lay, backward=True)
    torch.cuda.synchronize()
    # ---- vs the reference's own fp32 CPU run (north star: "fp logits within 1e-3 relative" is for like-for-like arithmetic; the
    # HIP path feeds bf16 operands to the MFMA, as the reference's GPU autocast path does, so the gap below is bf16 rounding — the
    # split-bf16 verification test (tests/test_verify_fp32_gpu.py) shows the same kernels reach < 1e-3 when fed fp32-accurate operands)
    Tv = 257
    xv = eng.ws["vit_x"].view(4, Tv, 1408)[:, ::8, ::4].cpu()
    check("c1.vit.out (39 blocks) vs reference-fp32", relerr(xv, g["vit_sub"]), 1.3e-2)
    ln = eng.ws["img"].view(4, Tv, -1)[:, ::8, :1408:4].float().cpu()
    check("c1.ln_vision vs reference-fp32", relerr(ln, g["ln_sub"]), 1.3e-2)
    qo = eng._qf_last_f32.view(4, 32, 768)[:, :, ::2].cpu()
    check("c1.qformer.out (12 layers) vs reference-fp32", relerr(qo, g["qf_out"]), 1.2e-2)
    emb = eng.ws["inputs_embeds"].view(1, lay.S, 768)[..., ::4].cpu()
    check("c1.inputs_embeds vs reference-fp32", relerr(emb, g["inputs_embs_sub"]), 1.2e-2)
    enc = eng.ws["e_out"][:, :768].float().view(1, lay.S, 768)[..., ::4].cpu()
    check("c1.t5.enc_out (12 layers) vs reference-fp32", relerr(enc, g["enc_sub"]), 2.5e-2)
    logits = eng.ws["d_logits"].view(1, -1, 32128).cpu()
    check("c1.logits vs reference-fp32", relerr(logits[..., ::64], g["logits_sub"]), 2.4e-2)
    check("c1.logits_lse vs reference-fp32", relerr(torch.logsumexp(logits, -1), g["logits_lse"]), 2e-5)
    check("c1.loss vs reference-fp32 (rel)", abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), 1e-3)
    check("c1.grad t5_proj.weight vs reference-fp32 autograd", relerr(eng.dproj_w.cpu()[::4], g["grad__t5_proj__weight"]), 5e-2)
    check("c1.grad t5_proj.bias vs reference-fp32 autograd", relerr(eng.dproj_b.cpu(), g["grad__t5_proj__bias"]), 5e-2)
    check("c1.grad ln_vision.weight vs reference-fp32 autograd", relerr(eng.dlnv_w.cpu(), g["grad__ln_vision__weight"]), 5e-2)
    check("c1.grad ln_vision.bias vs reference-fp32 autograd", relerr(eng.dlnv_b.cpu(), g["grad__ln_vision__bias"]), 5e-2)
    # ---- vs the oracle with bf16-operand emulation (same rounding points): logic errors would show here
    for k in ("t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"):
        sd[k].requires_grad_(True)
    orc = O.Oracle(sd, C1_CFG, emu_bf16=True)
    ref = orc.forward_mr(tok, samples, repl)
    ref["loss"].backward()
    check("c1.logits vs emu-oracle", relerr(logits, ref["logits"].detach()), 1.5e-2)
    # (measured 2.8e-4 .. 3.1e-4 across the round-3 builds — the thin LoRA products changed their fp32 summation order; 2x that)
    check("c1.loss vs emu-oracle (rel)", abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()), 6e-4)
    check("c1.t5.enc_out vs emu-oracle", relerr(eng.ws["e_out"][:, :768].float().cpu().view(1, lay.S, 768), ref["enc"].detach()), 1.5e-2)
    check("c1.grad t5_proj.weight vs emu-oracle autograd", relerr(eng.dproj_w.cpu(), sd["t5_proj.weight"].grad), 3.6e-2)
    check("c1.grad ln_vision.weight vs emu-oracle autograd", relerr(eng.dlnv_w.cpu(), sd["ln_vision.weight"].grad), 3.6e-2)


def test_c1_real_depth_nonzero_lora_gradients_against_oracle_autograd():
    """VERDICT r3 missing 1 / next 3(a): the LoRA adapters are 92 % of what the optimizer updates, and C1 / C2 above run with peft's
    initial B = 0, where the branch contributes exactly zero.  Here every adapter of the real-depth C1 model gets seeded NON-ZERO A and B
    (N(0, 0.02), the bench's lora_init_nonzero scale) and every adapter's dA / dB — plus loss, logits, t5_proj / ln_vision gradients — is
    compared with the autograd of the emu-bf16 oracle (peft semantics y = W x + (alpha / r) B A x restated from its published algorithm,
    blip2_mr.py:182-200, 236; peft itself is absent: parity unpinned for the LoRA numerics) and, for the record, with the
    oracle's plain fp32 run."""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from oracle import mrblip_oracle as O
    from test_model_gpu import _peft_sd

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    sdl = _peft_sd(sd, lora_std=0.02)
    train_keys = [k for k in sdl if ("lora_" in k) or k.startswith("t5_proj") or k.startswith("ln_vision")]
    for k in train_keys:
        sdl[k].requires_grad_(True)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)
    dev = torch.device("cuda:0")
    eng = MrBlipEngine(cfg, StateDictSource(sdl), dev)
    eng.training = False
    lay = P.build_layout(tok, samples, repl, 32, T=4)
    eng.zero_grad()
    loss = eng.forward_backward(samples["video"].to(dev), lay, backward=True)
    torch.cuda.synchronize()
    logits = eng.ws["d_logits"].view(1, -1, 32128).cpu()
    assert len(eng.adapters) == 12 * 7 + 12 * 11 + 1   # q k v o wi_0 wi_1 wo per encoder block, + cross q k v o per decoder block, + lm_head

    def compare(tag, emu, tol_loss, tol_logits, tol_tail, tol_lora):
        for k in train_keys:
            sdl[k].grad = None
        orc = O.Oracle(sdl, C1_CFG, emu_bf16=emu, lora=dict(r=8, alpha=8))
        ref = orc.forward_mr(tok, samples, repl)
        ref["loss"].backward()
        check(f"c1.lora!=0: loss vs {tag} (rel)", abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()), tol_loss)
        check(f"c1.lora!=0: logits vs {tag}", relerr(logits, ref["logits"].detach()), tol_logits)
        check(f"c1.lora!=0: grad t5_proj.weight vs {tag} autograd", relerr(eng.dproj_w.cpu(), sdl["t5_proj.weight"].grad), tol_tail)
        check(f"c1.lora!=0: grad ln_vision.weight vs {tag} autograd", relerr(eng.dlnv_w.cpu(), sdl["ln_vision.weight"].grad), tol_tail)
        worst, worst_name, num, den = 0.0, "", 0.0, 0.0
        by_kind = {}
        for a in eng.adapters:
            base = "t5_model.base_model.model." + a.name
            ga, gb = sdl[base + ".lora_A.default.weight"].grad, sdl[base + ".lora_B.default.weight"].grad
            ea, eb = relerr(a.dA.cpu(), ga), relerr(a.dBt.cpu().t(), gb)
            num += float((a.dA.cpu() - ga).pow(2).sum() + (a.dBt.cpu().t() - gb).pow(2).sum())
            den += float(ga.pow(2).sum() + gb.pow(2).sum())
            parts = a.name.split(".")   # e.g. encoder.block.3.layer.0.SelfAttention.q | decoder.block.3.layer.1.EncDecAttention.k | lm_head
            kind = (parts[0][:3] + "." + ".".join(parts[-2:])) if len(parts) > 2 else a.name
            by_kind[kind] = max(by_kind.get(kind, 0.0), ea, eb)
            if max(ea, eb) > worst:
                worst, worst_name = max(ea, eb), a.name
        for kind, v in sorted(by_kind.items()):
            record(f"c1.lora!=0: worst dA/dB of {kind} adapters vs {tag}", v, tol_lora)
        check(f"c1.lora!=0: ALL LoRA gradients (flat, {len(eng.adapters)} adapters) vs {tag} autograd", math.sqrt(num / den), tol_lora / 2)
        check(f"c1.lora!=0: worst single adapter dA/dB vs {tag} autograd ({worst_name})", worst, tol_lora)

    # tolerances = 2x the measured values of round 4 (emu-oracle: loss 2.1e-4, logits 1.0e-2, t5_proj / ln_vision 2.0-2.2e-2, all 217
    # adapters' gradients together 1.9e-2, the worst single adapter 4.1e-2; fp32 oracle: 2.2e-4, 1.2e-2, 2.1e-2, 4.6e-2)
    compare("emu-oracle", True, 5e-4, 2e-2, 4.5e-2, 8e-2)
    compare("oracle-fp32", False, 5e-4, 2.4e-2, 5e-2, 9e-2)


def test_c1_real_depth_training_mode_dropout_parity():
    """VERDICT r4 next 2(b), second half: training-mode parity at REAL depth (39-block ViT-g, 12-layer Q-Former, 12 + 12 T5 layers), non-zero
    LoRA in every adapter, every dropout of the reference ON — the HIP step against the emu-bf16 oracle fed with the SAME masks, rebuilt
    on the CPU through ``dropout_site_map()`` and the oracle's restatement of the counter hashes (hidden states: one hash per element pair;
    attention probabilities: draws v3).  Until round 5 this existed at tiny dimensions only (tests/test_model_gpu.py).  A wrong call-site
    id, a backward that regenerates another mask than its forward used, or keep bits read in the wrong layout would leave the loss right
    and the gradients wrong."""
    from weights import seeded_state_dict
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, StateDictSource
    from mrblip.tokenizer import FixtureTokenizer
    from oracle import mrblip_oracle as O
    from test_model_gpu import _peft_sd

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    sdl = _peft_sd(sd, lora_std=0.02)
    train_keys = [k for k in sdl if ("lora_" in k) or k.startswith("t5_proj") or k.startswith("ln_vision")]
    for k in train_keys:
        sdl[k].requires_grad_(True)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    cfg = EngineConfig(d_model=768, d_kv=64, t5_heads=12, d_ff=2048, t5_layers=12, t5_dec_layers=12)
    dev = torch.device("cuda:0")
    eng = MrBlipEngine(cfg, StateDictSource(sdl), dev, seed=20240)
    eng.training = True
    eng.graph_mode = "0"
    lay = P.build_layout(tok, samples, repl, 32, T=4)
    eng.zero_grad()
    loss = eng.forward_backward(samples["video"].to(dev), lay, backward=True)
    torch.cuda.synchronize()
    seed = int(eng.seed.item()) & 0xFFFFFFFF
    sites = eng.dropout_site_map()
    used = set()

    def provider(name, shape):
        site, p, kind = sites[name]
        used.add(name)
        if p <= 0:
            return None
        if kind == "attn":
            return O.dropout_keep_attn(*shape, seed, site, p) / (1.0 - p)
        return O.dropout_keep(shape, seed, site, p) / (1.0 - p)

    orc = O.Oracle(sdl, C1_CFG, emu_bf16=True, lora=dict(r=8, alpha=8), dropout=provider)
    ref = orc.forward_mr(tok, samples, repl)
    assert len(used) > 200 and "t5.dec.11.cross.attn" in used and any(k.startswith("lora:") for k in used)
    tag = "c1.train-mode (real depth, LoRA != 0, all dropouts on): "
    check(tag + "loss vs emu-oracle, same masks (rel)", abs(loss.item() - ref["loss"].item()) / abs(ref["loss"].item()), 5e-4)   # (measured 2.2e-4; gradients 3.0-3.3e-2, worst adapter 5.8e-2, norm ratio 1 + 2.6e-4, cosine deficit 4.6e-4)
    eng.training = False
    l_eval = eng.forward_backward(samples["video"].to(dev), lay, backward=False).item()
    assert abs(l_eval - ref["loss"].item()) > 1e-3          # dropout really happened
    ref["loss"].backward()
    check(tag + "grad t5_proj.weight vs emu-oracle autograd", relerr(eng.dproj_w.cpu(), sdl["t5_proj.weight"].grad), 5e-2)
    check(tag + "grad ln_vision.weight vs emu-oracle autograd", relerr(eng.dlnv_w.cpu(), sdl["ln_vision.weight"].grad), 5e-2)
    num = den = 0.0
    worst, worst_name = 0.0, ""
    hs, rs = [], []
    for a in eng.adapters:
        base = "t5_model.base_model.model." + a.name
        ga, gb = sdl[base + ".lora_A.default.weight"].grad, sdl[base + ".lora_B.default.weight"].grad
        ha, hb = a.dA.cpu(), a.dBt.cpu().t()
        ea, eb = relerr(ha, ga), relerr(hb, gb)
        num += float((ha - ga).pow(2).sum() + (hb - gb).pow(2).sum())
        den += float(ga.pow(2).sum() + gb.pow(2).sum())
        hs += [ha.reshape(-1), hb.reshape(-1)]
        rs += [ga.reshape(-1), gb.reshape(-1)]
        if max(ea, eb) > worst:
            worst, worst_name = max(ea, eb), a.name
    check(tag + f"ALL LoRA gradients (flat, {len(eng.adapters)} adapters) vs emu-oracle autograd, same masks", math.sqrt(num / den), 5e-2)
    check(tag + f"worst single adapter dA/dB ({worst_name})", worst, 1e-1)
    ratio, cos = _bias_report(tag + "all adapters", torch.cat(hs), torch.cat(rs))
    assert abs(ratio - 1.0) < 1e-2 and 1.0 - cos < 1e-3, (ratio, cos)


def _c2_setup(lora_init=None):
    """the engine, layout, clip and golden of the BENCHED size (BASELINE.json configs[1]); weights regenerated from their reference keys"""
    import os

    from util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "mr_c2.npz")):
        pytest.skip("tests/golden/mr_c2.npz not generated")
    from weights import seeded_array
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine
    from mrblip.tokenizer import FixtureTokenizer

    g = load_golden("mr_c2")
    st = g["strings"]
    shapes = dict((k, tuple(s)) for k, s in g["manifest"])

    class NameKeyed:   # weights by reference key, generated on demand (never 16 GB of fp32 on the host at once)
        def get(self, key, shape=None):
            if key not in shapes:
                raise KeyError(key)
            return torch.from_numpy(seeded_array(key, shapes[key], wscale=st["wscale"], fast=True))

        def has(self, key):
            return key in shapes

    dev = torch.device("cuda:0")
    src = NameKeyed()
    eng = MrBlipEngine(EngineConfig.flan_t5_xl_qvh(), src, dev, lora_init=lora_init)   # lora_init None: peft default init (B = 0) = the reference's LoRA-free run
    eng.training = False
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    T = int(st["T"])
    video = torch.from_numpy(seeded_array("c2.input.video", (1, T, 3, 224, 224), std=1.0, fast=True))
    samples = dict(video=video, timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]), query_prompt=st["query_prompt"],
                   task_prompt=st["task_prompt"], video_prompt_end=st["video_prompt_end"], relevant_windows=st["relevant_windows"])
    lay = P.build_layout(tok, samples, repl, 32, T=T)
    assert lay.S == g["inputs_atts"].shape[1] and lay.S > 1900
    assert np.array_equal(lay.attention_mask.numpy(), g["inputs_atts"]) and np.array_equal(lay.labels.numpy(), g["labels"])   # integer work: bit-exact
    return eng, src, lay, video, g, T


def test_c2_benched_size_against_reference():
    """VERDICT r2 missing 5 / weak 2: parity numbers AT THE BENCHED SIZE (BASELINE.json configs[1]: 60 frames, ViT-g/14 + Q-Former(32) +
    Flan-T5-XL: d 2048, 24 + 24 layers, S_enc ~ 2000).  tests/golden/mr_c2.npz holds sub-sampled outputs of the REFERENCE's own
    forward_mr + backward at that size (make_golden_c2.py, CPU fp32, eval mode); the 4 G weights are regenerated from their key names.
    Integer work (mask, labels) bit-exact; every tower's output, logits, loss and the t5_proj / ln_vision gradients against the reference."""
    import time

    t0 = time.time()
    eng, _, lay, video, g, T = _c2_setup()
    dev = eng.dev
    eng.zero_grad()
    loss = eng.forward_backward(video.to(dev), lay, backward=True)
    torch.cuda.synchronize()
    print("c2: engine built + step in %.0f s; loss %.5f (reference %.5f)" % (time.time() - t0, loss.item(), float(g["loss"])))
    Tv, d, S = 257, 2048, lay.S
    check("c2.vit.out (60 frames) vs reference-fp32", relerr(eng.ws["vit_x"].view(T, Tv, 1408)[::6, ::16, ::16].cpu(), g["vit_sub"]), 1.3e-2)
    check("c2.ln_vision vs reference-fp32", relerr(eng.ws["img"].view(T, Tv, -1)[::6, ::16, :1408:16].float().cpu(), g["ln_sub"]), 1.3e-2)
    check("c2.qformer.out vs reference-fp32", relerr(eng._qf_last_f32.view(T, 32, 768)[::6, ::4, ::8].cpu(), g["qf_sub"]), 5e-3)
    check("c2.inputs_embeds vs reference-fp32", relerr(eng.ws["inputs_embeds"].view(1, S, d)[:, ::4, ::16].cpu(), g["inputs_embs_sub"]), 8e-3)
    check("c2.t5.enc_out (24 layers, XL) vs reference-fp32", relerr(eng.ws["e_out"][:, :d].float().view(1, S, d)[:, ::4, ::16].cpu(), g["enc_sub"]), 1.6e-2)   # (measured 1.02e-2)
    logits = eng.ws["d_logits"].view(1, -1, 32128).cpu()
    check("c2.logits vs reference-fp32", relerr(logits[..., ::64], g["logits_sub"]), 1.5e-2)   # (measured 9.25e-3: bf16 operand rounding; north_star's 1e-3 is met by the fp32-operand mode, 1.5e-5)
    check("c2.logits_lse vs reference-fp32", relerr(torch.logsumexp(logits, -1), g["logits_lse"]), 1e-5)
    check("c2.loss vs reference-fp32 (rel)", abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), 6e-4)   # (bf16 rounding of the towers: 2.7e-5 .. 1.7e-4 between builds; C1: 4.6e-4)
    check("c2.grad t5_proj.weight vs reference-fp32 autograd", relerr(eng.dproj_w.cpu()[::16, ::4], g["grad__t5_proj__weight"]), 3e-2)   # (measured 2.05e-2 / 1.73e-2 / 1.65e-2 / 1.69e-2)
    check("c2.grad t5_proj.bias vs reference-fp32 autograd", relerr(eng.dproj_b.cpu(), g["grad__t5_proj__bias"]), 3e-2)
    check("c2.grad ln_vision.weight vs reference-fp32 autograd", relerr(eng.dlnv_w.cpu(), g["grad__ln_vision__weight"]), 3e-2)
    check("c2.grad ln_vision.bias vs reference-fp32 autograd", relerr(eng.dlnv_b.cpu(), g["grad__ln_vision__bias"]), 3e-2)
    del eng
    torch.cuda.empty_cache()


def _bias_report(tag, got, want):
    """A rounding-limited gradient is an UNBIASED noisy copy of the true one: its norm ratio is 1 +- err and its cosine 1 - err^2 / 2.  A
    logic error (a missing term, a wrong scale or mask) shows as a norm ratio off 1 or a cosine far below that.  Recorded per adapter class."""
    got, want = torch.as_tensor(got, dtype=torch.float64).reshape(-1), torch.as_tensor(want, dtype=torch.float64).reshape(-1)
    ratio = (got.norm() / want.norm().clamp_min(1e-30)).item()
    cos = (torch.dot(got, want) / (got.norm() * want.norm()).clamp_min(1e-30)).item()
    # (a SIGNED quantity's magnitude — it can sit arbitrarily close to 0 by chance, so no regression ceiling: the bound is the claim itself,
    # the norm ratio of a rounding-limited gradient — worst class measured 4.3e-3 (lm_head), the others 0.05-0.33 %)
    record(tag + ": |g_hip| / |g_ref| - 1 (abs)", abs(ratio - 1.0), 8e-3)
    print(f"bias report {tag}: |g_hip| / |g_ref| - 1 = {ratio - 1.0:+.3e}, 1 - cosine = {1.0 - cos:.3e}")
    record(tag + ": 1 - cosine", 1.0 - cos, 1e-3)
    return ratio, cos


def test_c2_benched_size_nonzero_lora_gradients():
    """VERDICT r4 missing 2 / next 2(b): the BENCHED model's LoRA path — XL width (d 2048, d_ff 5120, the K = 10240 dX, 24 + 24 layers), S = 2012,
    where the stacked cross K / V projection, the in-GEMM thin role and the tall-input thin kernels engage — with NON-ZERO A and B in every one
    of the 433 adapters, against the fp32 ORACLE's autograd (tests/golden/mr_c2_lora.npz from make_golden_c2_lora.py: sub-sampled dA / dB of
    every adapter + their full norms; the reference cannot provide this, peft is absent: "parity unpinned" for the LoRA numerics, DESIGN.md
    section 2).  Eval mode.  Besides the relative errors, every adapter class reports norm ratio and cosine (_bias_report): rounding noise is
    unbiased, a logic error is not."""
    import os

    from util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "mr_c2_lora.npz")):
        pytest.skip("tests/golden/mr_c2_lora.npz not generated")
    from weights import seeded_array
    gl = load_golden("mr_c2_lora")
    names, stride, std = gl["strings"]["names"], int(gl["strings"]["stride"]), float(gl["strings"]["lora_std"])

    def lora_init(a, gen):
        base = "t5_model.base_model.model." + a.name
        a.A.copy_(torch.from_numpy(seeded_array(base + ".lora_A.default.weight", (8, a.in_dim), std=std)))
        a.Bt.copy_(torch.from_numpy(seeded_array(base + ".lora_B.default.weight", (a.out, 8), std=std)).t())

    eng, _, lay, video, g, T = _c2_setup(lora_init=lora_init)
    dev = eng.dev
    eng.zero_grad()
    loss = eng.forward_backward(video.to(dev), lay, backward=True)
    torch.cuda.synchronize()
    d, S = 2048, lay.S
    tag = "c2.lora!=0: "
    # tolerances = ~1.6x the values measured in round 5 (loss 1.8e-4, encoder output 1.05e-2, logits 9.2e-3, t5_proj / ln_vision 2.2e-2 /
    # 1.9e-2, all 433 adapters flat 1.7e-2, the worst single adapter 6.0e-2 (a decoder cross-attention q), norm ratios within 4.3e-3 of 1,
    # cosine deficits <= 4.4e-4 = err^2 / 2 of an unbiased 3e-2 error)
    check(tag + "loss vs oracle-fp32 (rel)", abs(loss.item() - float(gl["loss"])) / abs(float(gl["loss"])), 4e-4)
    logits = eng.ws["d_logits"].view(1, -1, 32128).cpu()
    check(tag + "t5.enc_out vs oracle-fp32", relerr(eng.ws["e_out"][:, :d].float().view(1, S, d)[:, ::4, ::16].cpu(), gl["enc_sub"]), 1.7e-2)
    check(tag + "logits vs oracle-fp32", relerr(logits[..., ::64], gl["logits_sub"]), 1.5e-2)
    check(tag + "grad t5_proj.weight vs oracle-fp32 autograd", relerr(eng.dproj_w.cpu()[::16, ::4], gl["grad__t5_proj__weight"]), 3.6e-2)
    check(tag + "grad ln_vision.weight vs oracle-fp32 autograd", relerr(eng.dlnv_w.cpu(), gl["grad__ln_vision__weight"]), 3.2e-2)
    by_name = {a.name: a for a in eng.adapters}
    assert sorted(by_name) == sorted(names) and len(names) == 24 * 7 + 24 * 11 + 1
    oa = ob = 0
    num = den = 0.0
    worst, worst_name = 0.0, ""
    kinds = {}
    for nm in names:
        a = by_name[nm]
        na, nb = 8 * len(range(0, a.in_dim, stride)), len(range(0, a.out, stride)) * 8
        ga = torch.from_numpy(gl["lora_dA_sub"][oa: oa + na]).view(8, -1)
        gb = torch.from_numpy(gl["lora_dB_sub"][ob: ob + nb]).view(-1, 8)
        oa, ob = oa + na, ob + nb
        ha, hb = a.dA.cpu()[:, ::stride], a.dBt.cpu().t()[::stride]
        ea, eb = relerr(ha, ga), relerr(hb, gb)
        num += float((ha - ga).pow(2).sum() + (hb - gb).pow(2).sum())
        den += float(ga.pow(2).sum() + gb.pow(2).sum())
        parts = nm.split(".")
        kind = (parts[0][:3] + "." + ".".join(parts[-2:])) if len(parts) > 2 else nm
        k = kinds.setdefault(kind, dict(worst=0.0, h=[], r=[]))
        k["worst"] = max(k["worst"], ea, eb)
        k["h"] += [ha.reshape(-1), hb.reshape(-1)]
        k["r"] += [ga.reshape(-1), gb.reshape(-1)]
        if max(ea, eb) > worst:
            worst, worst_name = max(ea, eb), nm
        if os.environ.get("MRB_BIAS_DEBUG") and (nm == "lm_head" or nm.endswith("block.23.layer.2.DenseReluDense.wo") or nm.endswith("block.23.layer.1.DenseReluDense.wo")):
            print(f"bias debug {nm}: dA ratio-1 {float(ha.double().norm() / ga.double().norm()) - 1:+.3e} (err {ea:.2e}), dB ratio-1 {float(hb.double().norm() / gb.double().norm()) - 1:+.3e} (err {eb:.2e})")
    assert oa == gl["lora_dA_sub"].size and ob == gl["lora_dB_sub"].size
    for kind, k in sorted(kinds.items()):
        record(tag + f"worst dA/dB (sub-sampled) of {kind} adapters vs oracle-fp32", k["worst"], 1e-1)
        ratio, cos = _bias_report(tag + f"{kind} adapters, dA and dB together", torch.cat(k["h"]), torch.cat(k["r"]))
        assert abs(ratio - 1.0) < 8e-3 and 1.0 - cos < 1e-3, (kind, ratio, cos)
    check(tag + f"ALL LoRA gradients (flat over the sub-samples of {len(names)} adapters) vs oracle-fp32 autograd", math.sqrt(num / den), 2.8e-2)
    check(tag + f"worst single adapter dA/dB vs oracle-fp32 autograd ({worst_name})", worst, 1e-1)
    del eng
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def xl():
    """the bench's engine: QVH shape, Flan-T5-XL dims, random-init weights generated on the device"""
    from mrblip import prompt as P
    from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource
    from mrblip.tokenizer import FixtureTokenizer
    import bench

    dev = torch.device("cuda:0")
    # std 0.012, not the bench's 0.02: T5 attention is unscaled, so at d_model 2048 N(0, 0.02) weights give score std ~6.5 over 2012 keys —
    # nearly one-hot softmaxes that amplify a last-bit difference 40x through the decoder (make_golden_c2.py, which re-scaled the C2
    # golden for the same reason).  0.012 gives the score std ~2.4 of the C1 / C2 fixtures: comparisons between kernel choices then test
    # the kernels, not the conditioning of a random model.
    eng = MrBlipEngine(EngineConfig.flan_t5_xl_qvh(), RandomSource(dev, seed=1234, std=0.012), dev, lora_init=bench.lora_init_nonzero, seed=42)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    return eng, tok, repl, bench, dev


def _layout(xl, B, T, duration, mean_pool=False, seed=1234):
    from mrblip import prompt as P

    eng, tok, repl, bench, dev = xl
    samples = bench.synthetic_samples(B, T, duration, dev, seed)
    return samples, P.build_layout(tok, samples, repl, 1 if mean_pool else eng.cfg.num_query, T=T)


def test_c3_batch4_step_equals_four_accumulated_single_clip_steps(xl):
    """C3 shape (B = 4 per GPU, QVH T = 60, XL), dropout off so both runs see the same function: the gradient of the 4-clip mean loss
    must equal the mean of the four single-clip gradients (the data-parallel contract: a rank's B clips == B accumulated micro-steps)."""
    eng = xl[0]
    eng.training = False
    eng.cfg.mean_pool = False
    # The contract is exact only when both runs execute the SAME kernels.  The thin LoRA products are the one place where the kernel
    # choice follows the row count with a different summation order (row kernel up to 2100 rows = one QVH clip, MFMA skinny kernel above
    # = four clips), the attention forward another (one clip, 2016 query tiles, takes the key-split 8-wave form, four clips do not), the
    # decoder's projections a third (one clip's <= 16 label rows take the fused one-launch kernel of csrc/decproj.hip).  Pin the choices
    # for the exact comparison; the product setting is compared with it — loss AND gradient — further down.
    import os
    rows_max = eng.lora_rows_max_m
    eng.lora_rows_max_m = 256
    os.environ["MRB_ATTN_KS2"] = "0"
    dec_proj = eng.dec_proj_enabled
    eng.dec_proj_enabled = False   # (the fused decoder projection serves <= 16 rows: one clip's labels, not four clips')
    # round 4: two more shape-dependent choices — the stacked cross K / V projection of all decoder layers (one clip: its backward adds the
    # layers' contributions to the encoder-output gradient chunk-wise; four ragged clips keep the per-layer launches) and the cross-block key
    # split of the decoder's cross attention (other merge order of the softmax partials)
    ckv_b, xs_ws = eng.cross_kv_batched, eng.xs_ws
    eng.cross_kv_batched, eng.xs_ws = False, None
    # round 5: one more — the encoder backward's input gradients as K-split parts of the 4-wave kernel: four splits for one clip's 2012 rows
    # (64 output tiles), none for four clips' 8048 (256 tiles): another summation order over K
    bw4 = eng.enc_bwd_w4
    eng.enc_bwd_w4 = False
    samples, lay4 = _layout(xl, 4, 60, 150.0)
    # four different clips (same prompt, hence the same layout / label length per clip)
    video = samples["video"]
    video[1] = video[1].flip(-1)
    video[2] = video[2] * 0.7
    video[3] = video[3].flip(-2) * 1.2
    eng.zero_grad()
    l4 = eng.forward_backward(video, lay4, backward=True).item()
    g4 = eng.grad.clone()
    _, lay1 = _layout(xl, 1, 60, 150.0)
    eng.zero_grad()
    ls = [eng.forward_backward(video[i:i + 1].contiguous(), lay1, backward=True).item() for i in range(4)]
    g1 = eng.grad.clone() / 4
    assert math.isfinite(l4) and all(math.isfinite(x) for x in ls)
    check("c3.loss B=4 vs mean of 4 x B=1 (rel)", abs(l4 - sum(ls) / 4) / abs(l4), 1e-6)
    check("c3.flat-grad B=4 vs 4 accumulated B=1 steps", relerr(g4, g1), 5e-6)
    check("c3.grad-norm B=4 vs accumulated (rel)", abs(g4.norm().item() - g1.norm().item()) / g1.norm().item(), 1e-5)
    assert len(set(round(x, 3) for x in ls)) > 1  # the clips really differ
    # the product setting for four clips: the decoder's 4 x L_dec rows take the one-launch projections with 3 row tiles (and their
    # head-transposed copies, rows = clip * L_dec + position).  Not bit-equal to the two-launch run above (other summation order: u and the
    # outputs differ in their last bf16 bit); on this fixture's conditioning (see the xl fixture) the difference stays a rounding-sized
    # one in the loss AND in the flat gradient (round 3 asserted 3e-2 on the loss only, on an ill-conditioned N(0, 0.02) model where the
    # same switch moved the gradient by a relative 1.4).
    eng.dec_proj_enabled = True
    eng.zero_grad()
    l4f = eng.forward_backward(video, lay4, backward=True).item()
    g4f = eng.grad.clone()
    check("c3.loss B=4, fused decoder projections vs two-launch path (rel)", abs(l4f - l4) / abs(l4), 1e-4)        # measured 2.7e-5
    check("c3.flat-grad B=4, fused decoder projections vs two-launch path", relerr(g4f, g4), 1e-2)                   # measured 5.0e-3
    # ... and the thin-product kernel choice (row kernel / MFMA thin kernel by row count), same comparison
    eng.lora_rows_max_m = rows_max
    eng.zero_grad()
    l4t = eng.forward_backward(video, lay4, backward=True).item()
    check("c3.loss B=4, product thin-LoRA kernel choice vs pinned (rel)", abs(l4t - l4f) / abs(l4f), 1e-4)          # measured 9.4e-6
    check("c3.flat-grad B=4, product thin-LoRA kernel choice vs pinned", relerr(eng.grad, g4f), 1.3e-2)              # measured 6.1e-3
    # ... and the round-5 K-split parts of the encoder backward (B = 4: one part + the LoRA part per product)
    eng.enc_bwd_w4 = bw4
    eng.zero_grad()
    l4k = eng.forward_backward(video, lay4, backward=True).item()
    assert l4k == l4t                                                                                                 # (the forward is the same launches)
    check("c3.flat-grad B=4, encoder input gradients as 4-wave-kernel parts vs the generic tile path", relerr(eng.grad, g4f), 1.3e-2)    # measured 6.3e-3
    # ... and the round-4 choices for ONE clip (stacked cross K / V, key-split cross attention): the accumulated single-clip steps again
    eng.cross_kv_batched, eng.xs_ws = ckv_b, xs_ws
    eng.zero_grad()
    ls4 = [eng.forward_backward(video[i:i + 1].contiguous(), lay1, backward=True).item() for i in range(4)]
    check("c3.loss 4 x B=1, stacked cross K/V + key-split cross attention vs pinned per-layer forms (rel)", abs(sum(ls4) / 4 - l4t) / abs(l4t), 2e-4)
    check("c3.flat-grad 4 x B=1, stacked cross K/V + key-split cross attention vs pinned per-layer forms", relerr(eng.grad / 4, g4f), 2e-2)
    eng.lora_rows_max_m = rows_max
    eng.dec_proj_enabled = dec_proj
    del os.environ["MRB_ATTN_KS2"]


@pytest.mark.parametrize("name,T,dur,mean", [("c5.anet T=120", 120, 120.0, False), ("c4.charades T=20 mean-pool", 20, 30.0, True),
                                             ("c2.qvh T=60", 60, 150.0, False)])
def test_full_size_step_runs_and_is_sane(xl, name, T, dur, mean):
    """training mode (all dropouts on), full width and depth: finite loss at the random-init level (ln 32128 = 10.38 for uniform
    logits; the N(0, 0.02) lm_head gives logits of std ~0.9 -> ~5.5-6.5), finite non-zero gradients in every trainable segment,
    deterministic eval-mode loss."""
    eng = xl[0]
    eng.training = True
    eng.cfg.mean_pool = mean
    samples, lay = _layout(xl, 1, T, dur, mean)
    eng.zero_grad()
    loss = eng.forward_backward(samples["video"], lay, backward=True)
    torch.cuda.synchronize()
    l0 = loss.item()
    record(name + ": S_enc", lay.S)
    record(name + ": loss (training mode)", l0)
    assert math.isfinite(l0) and 3.0 < l0 < 12.0, l0
    gr = eng.grad
    assert bool(torch.isfinite(gr).all())
    nd = eng.n_decay
    for seg, t in (("lora+t5_proj.weight", gr[:nd]), ("t5_proj.bias", eng.dproj_b), ("ln_vision.weight", eng.dlnv_w), ("ln_vision.bias", eng.dlnv_b)):
        record(name + ": |grad| " + seg, t.norm().item())
        assert t.norm().item() > 0
    # eval mode: deterministic (two passes give the same loss up to the atomic summation order) and different from the training-mode loss (dropout was on).
    # (No descent check here: with N(0, 0.02) random weights the T5 residual stream is ~0.02 in scale, RMSNorm amplifies by 1/rms and
    # the gradient norm is ~1e6 — any representable step leaves the linear regime.  Gradient CORRECTNESS at real width is what the C1
    # test (vs the reference's autograd) and the C3 test (batch vs accumulation) establish.)
    eng.training = False
    le0 = eng.forward_backward(samples["video"], lay, backward=False).item()
    le1 = eng.forward_backward(samples["video"], lay, backward=False).item()
    record(name + ": eval loss", le0)
    assert abs(le0 - le1) <= 1e-6 * abs(le0) and math.isfinite(le0) and le0 != l0   # (the loss reduction is an fp32 atomic sum: last-bit order effects)
    eng.cfg.mean_pool = False
