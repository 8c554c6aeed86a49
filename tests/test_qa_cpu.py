"""Video-QA half of SURVEY.md §8 (f4), CPU side: the oracle's restatement of forward_QA / get_relevant_frames / extract_frames
(blip2_mr.py:309-431, 1098-1164) against goldens the REFERENCE produced (tests/golden/make_golden_qa.py -> mr_tiny_qa.npz), and the host-side
answerer layout (mrblip.prompt.build_qa_layout) against the reference's own encoder input, mask and labels."""
import numpy as np
import torch

from oracle import mrblip_oracle as O
from mrblip import prompt as P
from mrblip.tokenizer import FixtureTokenizer
from util import TINY_CFG, load_golden, golden_state_dict, relerr


def _setup():
    g = load_golden("mr_tiny_qa")
    s = g["strings"]
    samples = dict(video=torch.from_numpy(g["video"]), timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]),
                   qa_input=s["qa_input"], qa_output=s["qa_output"], question_id=s["question_id"])
    return g, s, samples, golden_state_dict(g), FixtureTokenizer()


def test_forward_qa_matches_the_reference():
    g, s, samples, sd, tok = _setup()
    assert any(k.startswith("answerer_model.") for k in sd) and any(k.startswith("t5_model.") for k in sd)
    assert not torch.equal(sd["answerer_model.shared.weight"], sd["t5_model.shared.weight"])      # two T5s with their own weights
    orc = O.Oracle(sd, TINY_CFG)
    with torch.no_grad():
        r = orc.forward_qa(tok, samples, s["nfa"])
    assert r["frame_idx"] == g["idx_uniform"].tolist()
    assert np.array_equal(r["inputs_atts"].numpy(), g["inputs_atts"]) and np.array_equal(r["labels"].numpy(), g["labels"])
    assert relerr(r["inputs_embs"], g["inputs_embs"]) < 2e-6
    assert relerr(r["logits"][..., ::64], g["logits_sub"]) < 5e-6
    assert relerr(torch.logsumexp(r["logits"], -1), g["logits_lse"]) < 1e-6
    assert abs(r["loss"].item() - float(g["loss_uniform"])) < 1e-5
    assert orc.t5_prefix == "t5_model."                       # the answerer prefix does not leak out of forward_qa


def test_localizer_frame_selection_matches_the_reference():
    """get_relevant_frames / extract_frames for given localizer answers: a plain window, start >= end (-> end = duration), no window at all
    (-> the whole video), an end beyond the duration (clipped to round(duration)) with a second window that is ignored"""
    g, s, samples, sd, tok = _setup()
    orc = O.Oracle(sd, TINY_CFG)
    m = orc.relevant_moments(s["preds"], samples["duration"])
    assert [[float(x) for x in mm] for mm in m] == s["moments"]
    m2 = orc.relevant_moments(s["preds2"], samples["duration"])
    assert [[float(x) for x in mm] for mm in m2] == s["moments2"]
    nfa = s["nfa"]
    assert orc.extract_frames(samples["video"], samples["timestamps"], samples["duration"], m, nfa)[1] == g["idx_loc"].tolist()
    assert orc.extract_frames(samples["video"], samples["timestamps"], samples["duration"], m2, nfa)[1] == g["idx_loc2"].tolist()
    with torch.no_grad():
        r = orc.forward_qa(tok, samples, nfa, moments=m)
    assert abs(r["loss"].item() - float(g["loss_loc"])) < 1e-5


def test_answerer_layout_is_bit_exact():
    g, s, samples, sd, tok = _setup()
    nq = TINY_CFG["qf"]["num_query_token"]
    lay = P.build_qa_layout(tok, s["qa_input"], s["qa_output"], s["nfa"] * nq)
    B, S = g["inputs_atts"].shape
    assert lay.S == S and np.array_equal(lay.attention_mask.numpy(), g["inputs_atts"])
    assert np.array_equal(lay.labels.numpy(), g["labels"]) and np.array_equal(lay.decoder_mask.numpy(), g["dec_mask"])
    assert torch.equal(lay.decoder_input_ids, O.shift_right(lay.labels))
    # the index maps rebuild the reference's encoder input from (frame tokens, the ANSWERER's embedding table)
    emb = sd["answerer_model.shared.weight"]
    ref = torch.from_numpy(g["inputs_embs"])
    frames = ref[:, : s["nfa"] * nq].reshape(-1, ref.shape[-1])
    out = torch.zeros(B * S, ref.shape[-1])
    out[lay.frame_dst.long()] = frames[lay.frame_src.long()]
    out[lay.emb_dst.long()] = emb[lay.emb_src.long()]
    assert torch.equal(out.view(B, S, -1), ref)
    assert lay.labels.tolist() == [[71, 1], [205, 1]]          # "A" / "C" + EOS: the option ids of blip2_mr.py:1299
