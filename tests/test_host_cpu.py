"""Host-side integer logic of the product (mrblip.prompt) against the golden vectors and the pinned oracle: bit-exact
timestamp-token indexing, interleave index map, masks, labels, relative-position LUT.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import mrblip_oracle as O
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer
from util import TINY_CFG, load_golden, golden_state_dict


def test_annoying_numbers_and_timestamps_match_golden():
    tok = FixtureTokenizer()
    g = load_golden("mr_tiny")
    multi, _ = P.find_annoying_numbers(tok, 200)
    assert multi == g["annoying"].tolist()
    repl = P.annoying_replacement_dict(multi)
    assert sorted(repl.items()) == [tuple(r) for r in g["annoying_map"].tolist()]
    t = load_golden("timestamps")
    ts, d = P.seconds_integers(torch.from_numpy(t["ts"]), torch.from_numpy(t["dur"]), repl)
    assert ts[0] == t["out"].tolist() and d == t["out_dur"].tolist()


def _layout_case(tag, mean):
    g = load_golden(tag)
    tok = FixtureTokenizer()
    sd = golden_state_dict(g)
    orc = O.Oracle(sd, TINY_CFG)
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    s = g["strings"]
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                   query_prompt=s["query_prompt"], task_prompt=s["task_prompt"], video_prompt_end=s["video_prompt_end"],
                   relevant_windows=s["relevant_windows"])
    with torch.no_grad():
        out = orc.forward_mr(tok, samples, repl, mean_pool=mean)
    n = 1 if mean else 8
    lay = P.build_layout(tok, samples, repl, n, T=3)
    B = 2
    assert lay.S == g["inputs_embs"].shape[1]
    assert torch.equal(lay.attention_mask.long(), torch.from_numpy(g["inputs_atts"]))          # bit-exact mask
    assert torch.equal(lay.labels, torch.from_numpy(g["labels"]))                               # bit-exact labels
    assert torch.equal(lay.decoder_input_ids, O.shift_right(torch.from_numpy(g["labels"])))
    emb = sd["t5_model.shared.weight"]
    frames = out["frames"].reshape(-1, emb.shape[1])
    inp = torch.full((B * lay.S, emb.shape[1]), float("nan"))
    inp[lay.frame_dst.long()] = frames[lay.frame_src.long()]
    src = lay.emb_src.long()
    rows = torch.where((src >= 0)[:, None], emb[src.clamp_min(0)], torch.zeros(1, emb.shape[1]))
    inp[lay.emb_dst.long()] = rows
    assert not torch.isnan(inp).any()                                                             # every row is covered exactly
    assert len(set(lay.frame_dst.tolist()) | set(lay.emb_dst.tolist())) == B * lay.S
    assert torch.equal(inp.reshape(B, lay.S, -1), out["inputs_embs"])                           # bit-exact interleave
    assert np.allclose(inp.reshape(B, lay.S, -1).numpy(), g["inputs_embs"], rtol=0, atol=2e-5)


def test_interleave_layout_bit_exact():
    _layout_case("mr_tiny", False)


def test_interleave_layout_meanpool_bit_exact():
    _layout_case("mr_tiny_mean", True)


def test_bias_lut_matches_bucket_golden():
    g = load_golden("t5_buckets")
    for bidir, key in ((True, "bidir"), (False, "unidir")):
        got = [P.relative_position_bucket(int(r), bidir) for r in g["rel"]]
        assert got == g[key].tolist()
    table = torch.randn(32, 4)
    orc = O.Oracle({}, TINY_CFG)
    for bidir in (True, False):
        lut = P.bias_lut(table, bidir)
        ref = orc.t5_bias(table, 300, 300, bidir)[0]                       # [H, q, k]
        rel = (torch.arange(300)[None, :] - torch.arange(300)[:, None]).clamp(-128, 128) + 128
        assert torch.equal(lut[:, rel], ref)


def test_interleave_layout_edge_cases_against_oracle():
    """Ragged and extreme inputs of the prompt builder against the reference-pinned oracle restatement (bit-exact rows, masks, labels):
    one frame; batches whose clips differ in timestamp-token counts (multi-token 'annoying' integers are remapped, durations are not
    all equal -> left zero padding with mask 1); queries of very different lengths (right padding with mask 0); a query beyond
    max_txt_len = 200 tokens (truncation); empty target; many target windows."""
    torch.manual_seed(0)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    d = 16
    emb = torch.randn(tok.vocab_size if hasattr(tok, "vocab_size") else 32128, d)
    orc = O.Oracle({"t5_model.shared.weight": emb}, TINY_CFG)
    long_q = "Query: " + " ".join(["word%d" % i for i in range(300)]) + "\n"
    cases = [
        dict(T=1, n=4, dur=[7.0], q=["Query: a\n"], w=["[[0, 1]]"]),
        dict(T=3, n=2, dur=[150.0, 9.49, 1234.5], q=["Query: a dog\n", "Query: a much longer query about several different things\n", "Query: b\n"],
             w=["[[8, 16]]", "[[0, 2], [4, 6], [100, 120]]", ""]),
        dict(T=5, n=1, dur=[30.0, 30.0], q=[long_q, "Query: short\n"], w=["[[1, 2]]", "[[3, 4]]"]),
        dict(T=4, n=8, dur=[199.5, 0.4], q=["Query: x\n", "Query: y\n"], w=["[[0, 199]]", "[[0, 0]]"]),
    ]
    for c in cases:
        B, T, n = len(c["dur"]), c["T"], c["n"]
        dur = torch.tensor(c["dur"])
        ts = torch.stack([torch.tensor([round((i + 0.5) * float(x) / T, 2) for i in range(T)]) for x in c["dur"]])
        samples = dict(video=torch.zeros(B, T, 3, 2, 2), timestamps=ts, duration=dur, query_prompt=c["q"],
                       task_prompt=["Given the video and the query, find the relevant windows.\nRelevant windows: "] * B,
                       video_prompt_end=["<extra_id_0>"] * B, relevant_windows=c["w"])
        frames = torch.randn(B, T * n, d)
        embs, atts = orc.prompt_concatenation(tok, ts, dur, frames, samples["video_prompt_end"], samples["query_prompt"], samples["task_prompt"], repl, n)
        lay = P.build_layout(tok, samples, repl, n, T=T)
        assert lay.S == embs.shape[1], (c["T"], lay.S, embs.shape)
        assert torch.equal(lay.attention_mask.long(), atts)
        inp = torch.full((B * lay.S, d), float("nan"))
        inp[lay.frame_dst.long()] = frames.reshape(-1, d)[lay.frame_src.long()]
        src = lay.emb_src.long()
        inp[lay.emb_dst.long()] = torch.where((src >= 0)[:, None], emb[src.clamp_min(0)], torch.zeros(1, d))
        assert not torch.isnan(inp).any()
        assert torch.equal(inp.reshape(B, lay.S, d), embs)
        ans = tok(samples["relevant_windows"], padding="longest", truncation=True, max_length=200, return_tensors="pt")
        labels = ans.input_ids.masked_fill(ans.input_ids == tok.pad_token_id, -100)
        assert torch.equal(lay.labels, labels) and torch.equal(lay.decoder_input_ids, O.shift_right(labels))


def test_seconds_floats_layout_matches_reference_golden():
    """input_time_format="seconds_floats" (utils.py:464-485): product layout and oracle against the reference's own forward_mr
    (tests/golden/mr_tiny_floats.npz): mask and every embedding / zero-pad row of the interleaved encoder input, incl. the reference's
    float32 -> str quirk ("22.49" is tokenised as "22.489999771118164")."""
    g = load_golden("mr_tiny_floats")
    tok = FixtureTokenizer()
    sd = golden_state_dict(g)
    s = g["strings"]
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                   query_prompt=s["query_prompt"], task_prompt=s["task_prompt"], video_prompt_end=s["video_prompt_end"],
                   relevant_windows=s["relevant_windows"])
    assert P.seconds_floats(samples["timestamps"], samples["duration"])[0][0][1] == "22.489999771118164"
    lay = P.build_layout(tok, samples, {}, 8, T=3, time_format="seconds_floats")
    assert lay.S == g["inputs_embs"].shape[1]
    assert torch.equal(lay.attention_mask.long(), torch.from_numpy(g["inputs_atts"]))
    emb = sd["t5_model.shared.weight"]
    gold = torch.from_numpy(g["inputs_embs"]).reshape(-1, emb.shape[1])
    src = lay.emb_src.long()
    rows = torch.where((src >= 0)[:, None], emb[src.clamp_min(0)], torch.zeros(1, emb.shape[1]))
    assert torch.equal(gold[lay.emb_dst.long()], rows)                       # every timestamp / text / pad row: bit-exact
    assert len(set(lay.frame_dst.tolist()) | set(lay.emb_dst.tolist())) == 2 * lay.S
    orc = O.Oracle(sd, TINY_CFG)
    with torch.no_grad():
        out = orc.forward_mr(tok, samples, {}, time_format="seconds_floats")
    assert np.allclose(out["inputs_embs"].numpy(), g["inputs_embs"], rtol=0, atol=2e-5) and abs(out["loss"].item() - float(g["loss"])) < 1e-4
    with pytest.raises(ValueError):
        P.build_layout(tok, samples, {}, 8, T=3, time_format="relative_floats")


def test_attention_dropout_draws_statistics():
    """csrc/attention.hip "draws v3" (restated by the oracle): one hash per key quad — its finaliser on the 24-bit multiplier — and four
    16-bit draws, each with another byte of the hash as its top byte.  The drop rate must be p (6554 / 65536 for p = 0.1), every quad
    position must have the same marginal rate, and the draws of one quad (a draw's LOW byte is another's top byte) must stay (nearly)
    uncorrelated: P(drop j | drop i) within a few percent of P(drop) for all six pairs; keys of different quads / rows independent; the
    16 keep patterns of a quad close to the binomial expectation."""
    import torch
    from oracle import mrblip_oracle as O

    p = 0.1
    keep = O.dropout_keep_attn(2, 8, 256, 2012, seed=12345, site=77, p=p)      # 8.2 M draws
    drop = 1.0 - keep
    n = drop.numel()
    rate = drop.mean().item()
    target = int(p * 65536 + 0.5) / 65536
    assert abs(rate - target) < 4 * (target * (1 - target) / n) ** 0.5 + 1e-4, rate
    d = drop.reshape(-1, 2012)
    quad = d[:, :2012].reshape(d.shape[0], 503, 4)
    for i in range(4):
        for j in range(i + 1, 4):
            inside = (quad[:, :, i] * quad[:, :, j]).mean().item() / target       # P(drop j | drop i) inside a quad
            assert abs(inside - target) < 5e-3, (i, j, inside)                    # measured 0.099 .. 0.103 (v2's neighbours: 0.1016)
    across = (d[:, 3:2007:4] * d[:, 4:2008:4]).mean().item() / target            # keys 4i+3, 4i+4: different hashes
    rows = (d[:-1] * d[1:]).mean().item() / target
    assert abs(across - target) < 3e-3 and abs(rows - target) < 3e-3, (across, rows)
    # every draw position has the same marginal rate
    for j in range(4):
        assert abs(d[:, j::4].mean().item() - target) < 2e-3
    # the 16 keep patterns of a quad: relative deviation from the binomial expectation (the rarest, all four dropped, is 1e-4 of the quads)
    pat = (quad * torch.tensor([1.0, 2.0, 4.0, 8.0])).sum(-1).long().reshape(-1)
    cnt = torch.bincount(pat, minlength=16).double()
    for m in range(15):                                                           # (15 = all dropped: ~200 expected, too few to bound tightly)
        k = bin(m).count("1")
        exp = target ** k * (1 - target) ** (4 - k) * pat.numel()
        assert abs(cnt[m].item() - exp) < 0.06 * exp + 5 * exp ** 0.5, (m, cnt[m].item(), exp)
    # a different seed or site gives a different mask
    assert not torch.equal(keep, O.dropout_keep_attn(2, 8, 256, 2012, seed=12346, site=77, p=p))
    assert not torch.equal(keep, O.dropout_keep_attn(2, 8, 256, 2012, seed=12345, site=78, p=p))


def test_attention_dropout_quad_hashes_repeat_at_production_size_and_what_that_costs():
    """mrb_lin_fin24 feeds 24 bits into its multiplier, so a (seed, site) has at most 2^24 distinct quad hashes, and one T5-XL
    encoder layer at S = 2012 draws 32 heads x 2012 queries x 503 key quads = 32.4 M of them: quads MUST repeat.  Measured here on the
    restatement at that size: how many of a layer's quads share their hash with another quad of the layer (the balls-in-bins expectation
    for 2^24 equally likely values), that the repeats are spread evenly (no value is hit much more often than Poisson says: no lattice
    structure a training signal could lock on to) and that two quads of the SAME query row never share a hash (an attention row never
    sees a repeated 4-key pattern).  What it costs: far-apart score tiles of a layer may carry the same 4-key keep pattern — the marginal
    rate, the within-quad conditionals and the row / neighbour independence (test above) are untouched; the oracle restates the same function,
    so this is a property of the RNG, not a parity gap.  (The remedy — folding the dropped top byte back in — costs 2 VALU per hash, ~4 % of
    the VALU-bound encoder attention forward: not taken.)"""
    import numpy as np
    import torch
    from oracle import mrblip_oracle as O

    H, S = 32, 2012
    skq = (S + 3) // 4
    n = H * S * skq
    idx = torch.arange(n, dtype=torch.int64)
    h = O.dropout_hash_lin24(idx, seed=12345, site=77).numpy().astype(np.uint32)
    vals, counts = np.unique(h, return_counts=True)
    distinct = vals.size
    lam = n / 2.0 ** 24
    expect_distinct = 2.0 ** 24 * (1.0 - np.exp(-lam))
    assert distinct <= 2 ** 24
    # measured: 14.58 M distinct values of 32.4 M quads (2^24 equally likely values: 14.34 M — the Weyl walk covers slightly BETTER than random)
    assert abs(distinct - expect_distinct) < 0.03 * expect_distinct, (distinct, expect_distinct)
    shared = 1.0 - (counts == 1).sum() / n                                                            # quads whose hash another quad also has
    assert abs(shared - (1.0 - np.exp(-lam))) < 0.03, shared                                          # measured 0.873 (Poisson: 0.855)
    assert counts.max() <= 16, counts.max()                                                           # measured 10 (Poisson(1.93): P(> 16) ~ 1e-10 per value)
    # inside one query row (503 consecutive indices = a Weyl walk) every quad has its own hash
    rows = h.reshape(H * S, skq)[:: 97]
    assert all(np.unique(r).size == skq for r in rows)


@pytest.mark.parametrize("tag,fmt", [("mr_tiny_nointerleave", "seconds_integers"), ("mr_tiny_nointerleave_floats", "seconds_floats")])
def test_non_interleaved_prompt_layout_and_oracle_against_reference(tag, fmt):
    """interleave_data: False (blip2_mr.py:783-822; the reference constructor's default, no shipped config): the prompt is
    [ video_prompt text | all frame tokens | video_prompt_end | text ].  Golden from the imported reference
    (tests/golden/make_golden_nointerleave.py): the video_prompt strings, attention mask and labels bit-exact; the token rows of the
    encoder input bit-exact (embedding gathers), the whole encoder input and the loss through the oracle."""
    import json

    import numpy as np
    import torch
    from util import TINY_CFG, load_golden, golden_state_dict
    from mrblip import prompt as P
    from mrblip.tokenizer import FixtureTokenizer
    from oracle import mrblip_oracle as O

    g = load_golden(tag)
    st = g["strings"]
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                   query_prompt=st["query_prompt"], task_prompt=st["task_prompt"], video_prompt_end=st["video_prompt_end"],
                   relevant_windows=st["relevant_windows"])
    # the reference returns video_prompt + "frames" + video_prompt_end as its description of the prompt
    want = [v.split("frames")[0] for v in st["video_prompt"]]
    assert P.video_prompt_strings(samples, repl, fmt) == want
    lay = P.build_layout(tok, samples, repl, 8, T=3, time_format=fmt, interleave=False)
    assert lay.S == g["inputs_atts"].shape[1]
    assert np.array_equal(lay.attention_mask.numpy(), g["inputs_atts"]) and np.array_equal(lay.labels.numpy(), g["labels"])
    sd = golden_state_dict(g)
    emb = sd["t5_model.shared.weight"]
    got = torch.zeros(2 * lay.S, emb.shape[1])
    got[lay.emb_dst.long()] = emb[lay.emb_src.long()]
    ref = torch.from_numpy(g["inputs_embs"]).reshape(2 * lay.S, -1)
    rows = torch.zeros(2 * lay.S, dtype=torch.bool)
    rows[lay.emb_dst.long()] = True
    assert torch.equal(got[rows], ref[rows])                       # token rows: bit-exact gathers
    assert int((~rows).sum()) == lay.frame_dst.numel() == 2 * 3 * 8 and not rows[lay.frame_dst.long()].any()
    with torch.no_grad():
        out = O.Oracle(sd, TINY_CFG).forward_mr(tok, samples, repl, time_format=fmt, interleave=False)
    assert torch.equal(out["inputs_atts"].int(), lay.attention_mask)
    assert float((out["inputs_embs"] - torch.from_numpy(g["inputs_embs"])).abs().max()) < 2e-5
    assert abs(out["loss"].item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
