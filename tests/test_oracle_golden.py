"""Pins the CPU oracle (oracle/mrblip_oracle.py) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import mrblip_oracle as O
from mrblip.tokenizer import FixtureTokenizer
from util import TINY_CFG, load_golden, golden_state_dict, relerr


def test_vit_matches_reference():
    g = load_golden("vit_tiny")
    orc = O.Oracle(golden_state_dict(g), TINY_CFG)
    img = torch.from_numpy(g["image"])
    assert relerr(orc.vit(img, n_blocks=1), g["block0"]) < 2e-6
    assert relerr(orc.vit(img), g["out"]) < 2e-6


def test_qformer_matches_reference():
    g = load_golden("qformer_tiny")
    orc = O.Oracle(golden_state_dict(g), TINY_CFG)
    ln = orc.ln_vision(torch.from_numpy(g["vit_out"]))
    assert relerr(ln, g["ln_out"]) < 2e-6
    assert relerr(orc.qformer(ln), g["out"]) < 5e-6


def test_t5_matches_reference():
    g = load_golden("t5_tiny")
    orc = O.Oracle(golden_state_dict(g), TINY_CFG)
    x = torch.from_numpy(g["inputs_embeds"]).requires_grad_(True)
    labels = torch.from_numpy(g["labels"])
    assert torch.equal(O.shift_right(labels), torch.from_numpy(g["shift_right"]))
    loss, logits, enc = orc.t5_loss(x, torch.from_numpy(g["attention_mask"]), labels, (labels != -100).long())
    assert relerr(enc, g["enc_out"]) < 5e-6
    assert relerr(logits[..., ::64], g["logits_sub"]) < 1e-5
    assert relerr(torch.logsumexp(logits, -1), g["logits_lse"]) < 1e-6
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    assert relerr(x.grad, g["d_inputs_embeds"]) < 2e-5


def test_relative_position_buckets_exact():
    g = load_golden("t5_buckets")
    assert np.array_equal(O.relative_position_bucket(g["rel"], True), g["bidir"])
    assert np.array_equal(O.relative_position_bucket(g["rel"], False), g["unidir"])


def _mr(tag, mean):
    g = load_golden(tag)
    tok = FixtureTokenizer()
    sd = golden_state_dict(g)
    for k in list(sd):
        sd[k].requires_grad_(k.startswith("t5_proj") or k.startswith("ln_vision"))
    orc = O.Oracle(sd, TINY_CFG)
    annoying, _ = O.find_annoying_numbers(tok, 200)
    assert annoying == g["annoying"].tolist()
    repl = O.annoying_replacement_dict(annoying)
    assert sorted(repl.items()) == [tuple(r) for r in g["annoying_map"].tolist()]
    s = g["strings"]
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]),
                   duration=torch.from_numpy(g["duration"]), query_prompt=s["query_prompt"], task_prompt=s["task_prompt"],
                   video_prompt_end=s["video_prompt_end"], relevant_windows=s["relevant_windows"])
    out = orc.forward_mr(tok, samples, repl, mean_pool=mean)
    assert torch.equal(out["inputs_atts"], torch.from_numpy(g["inputs_atts"]))       # bit-exact mask
    assert torch.equal(out["labels"], torch.from_numpy(g["labels"]))                 # bit-exact label ids
    ge = torch.from_numpy(g["inputs_embs"])
    assert out["inputs_embs"].shape == ge.shape
    # rows that are pure embedding gathers / zero padding must be bit-exact (timestamp-token indexing)
    emb = sd["t5_model.shared.weight"]
    exact_rows = (ge[..., None, :] == emb[None, None, :2000]).all(-1).any(-1) | (ge == 0).all(-1)
    assert exact_rows.sum() > 10
    assert torch.equal(out["inputs_embs"][exact_rows], ge[exact_rows])
    assert relerr(out["inputs_embs"], ge) < 1e-5
    assert relerr(out["logits"][..., ::64], g["logits_sub"]) < 2e-5
    assert abs(out["loss"].item() - float(g["loss"])) < 2e-5
    out["loss"].backward()
    for n in ["t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"]:
        assert relerr(sd[n].grad, g["grad__" + n.replace(".", "__")]) < 1e-4, n
    assert s["trainable_top"] == ["ln_vision", "t5_model", "t5_proj"]


def test_forward_mr_matches_reference():
    _mr("mr_tiny", False)


def test_forward_mr_meanpool_matches_reference():
    _mr("mr_tiny_mean", True)


def test_timestamps_integer_exact():
    g = load_golden("timestamps")
    tok = FixtureTokenizer()
    repl = O.annoying_replacement_dict(O.find_annoying_numbers(tok, 200)[0])
    ts, d, prompt = O.timestamps_as_seconds_integers(torch.from_numpy(g["ts"]), torch.from_numpy(g["dur"]), repl)
    assert ts[0].tolist() == g["out"].tolist()
    assert d == g["out_dur"].tolist()
    assert prompt == g["prompt"]


def test_post_process_and_moments():
    g = load_golden("post_process")["cases"]
    for c, p, m in zip(g["cases"], g["post"], g["moments"]):
        assert O.post_process(c) == p, c
        assert O.moment_str_to_list(p) == m, c


def test_lr_schedule():
    g = load_golden("lr_sched")
    st = {}
    lrs = []
    for ep in range(5):
        for it in range(20):
            lrs.append(O.lr_at(ep, ep * 20 + it, st, max_epoch=5, min_lr=0.0, init_lr=3e-4, warmup_steps=30, warmup_start_lr=1e-8))
    assert np.allclose(lrs, g["lrs"], rtol=1e-12, atol=0)


def test_c1_real_depth_oracle_matches_reference():
    """BASELINE.json configs[0] at real depth/width (ViT-g 39 blocks x 1408, bert-base Q-Former, T5-base-sized 12 + 12 layers, 4
    frames): the fp32 oracle against tests/golden/mr_c1.npz, which make_golden_c1.py captured from the reference's own forward_mr."""
    from weights import seeded_array, seeded_state_dict
    from test_fullsize_gpu import C1_CFG, _c1_samples

    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    for k in ("t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"):
        sd[k].requires_grad_(True)
    tok = FixtureTokenizer()
    repl = O.annoying_replacement_dict(O.find_annoying_numbers(tok, 200)[0])
    samples = _c1_samples(g)
    ref = O.Oracle(sd, C1_CFG).forward_mr(tok, samples, repl)
    assert np.array_equal(ref["inputs_atts"].numpy(), g["inputs_atts"]) and np.array_equal(ref["labels"].numpy(), g["labels"])
    assert relerr(ref["inputs_embs"].detach()[..., ::4], g["inputs_embs_sub"]) < 1e-4
    assert relerr(ref["enc"].detach()[..., ::4], g["enc_sub"]) < 1e-4
    assert relerr(ref["logits"].detach()[..., ::64], g["logits_sub"]) < 1e-4
    assert abs(ref["loss"].item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    ref["loss"].backward()
    assert relerr(sd["t5_proj.weight"].grad[::4], g["grad__t5_proj__weight"]) < 1e-3
    assert relerr(sd["ln_vision.weight"].grad, g["grad__ln_vision__weight"]) < 1e-3
