"""Goldens for the video-QA half of SURVEY.md §8 (f4): the reference's two-stage path at tiny dimensions —
``forward_QA`` (blip2_mr.py:309-431; uniform sampling over the whole video, and the frame selection of the with_localizer variant:
``get_relevant_frames`` / ``extract_frames``, :1098-1164, fed with given localizer predictions), ``get_frame_embeddings_and_attentions``
(:948-988) and ``videoQA_generate`` / ``videoQA_answer`` (:990-1096, 1237-1314).  Writes mr_tiny_qa.npz: inputs, the (key, shape) manifest
(the answerer T5 has its OWN weights: keys ``answerer_model.*``), the selected frame indices, the answerer's encoder input / mask / labels,
loss, sub-sampled logits, the answer-option logits of the second generated step and the predicted option.

peft is absent from the image: get_peft_model is the identity in ref_shim, i.e. the LoRA branch is at its initial state B = 0.
Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_qa.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "mr-blip_amd"))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import make_golden as MG  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402

NFA = 3      # num_frames_for_answer


def main():
    R = ref_shim.install(FixtureTokenizer, MG.TINY)
    mr = R["mr"]
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(77)
    model = mr.BLIP2_MR(img_size=MG.IMG, vit_precision="fp32", num_query_token=MG.NQ, t5_model="google/flan-t5-xl",
                        input_time_format="seconds_integers", interleave_data=True, frame_token_aggregation=False,
                        task="qformer_freeze_lora_QA", num_frames_for_answer=NFA)
    model.eval()
    # (the QA constructor casts the frozen LOCALIZER T5 to bf16, blip2_mr.py:206-209: make_golden's manifest_of lists fp32 tensors only —
    # take every floating tensor, the load rounds the seeded fp32 values into the bf16 ones)
    man = [(k, list(v.shape)) for k, v in model.state_dict().items() if v.is_floating_point()]
    missing = model.load_state_dict({k: v for k, v in MG.seeded_state_dict(man).items()}, strict=False)
    assert not missing.unexpected_keys
    assert model.t5_model.shared.weight.dtype == torch.bfloat16 and model.answerer_model.shared.weight.dtype == torch.float32
    Bv, T = 2, 6
    video = torch.randn(Bv, T, 3, MG.IMG, MG.IMG, generator=g)
    timestamps = torch.tensor([[2.5, 10.0, 17.5, 25.0, 32.5, 40.0], [7.0, 45.2, 83.4, 121.6, 159.8, 198.0]], dtype=torch.float32)
    duration = torch.tensor([45.0, 205.0])
    qa_input = ["Question: what does the man open? Options: A: a door B: a box C: a can D: a book E: a bag. Answer: ",
                "Question: where does the dog run? Options: A: home B: away C: park D: road E: yard. Answer: "]
    qa_output = ["A", "C"]
    samples = dict(video=video, timestamps=timestamps, duration=duration, qa_input=qa_input, qa_output=qa_output,
                   question_id=["q0", "q1"], iters=1)
    cap = {}
    orig_fwd = model.answerer_model.forward

    def afwd(*a, **k):
        r = orig_fwd(*a, **k)
        if k.get("labels") is not None:
            cap["embs"], cap["atts"] = k["inputs_embeds"].detach().clone(), k["attention_mask"].clone()
            cap["labels"], cap["dec_mask"] = k["labels"].clone(), k["decoder_attention_mask"].clone()
            cap["logits"] = r.logits.detach().clone()
        return r

    model.answerer_model.forward = afwd
    # ---- forward_QA, uniform sampling over the whole video (use_localizer False)
    out = model(dict(samples))
    loss_uniform = out["loss"].detach().clone()
    uni = {k: v.clone() for k, v in cap.items()}
    # which frames were taken: extract_frames over [0, duration]
    rel_uniform = model.extract_frames(samples, [[0, d.item()] for d in duration], NFA)
    idx_uniform = [[int((video[b] - rel_uniform[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(NFA)] for b in range(Bv)]
    # ---- the with_localizer variant's frame selection for GIVEN localizer outputs (generate() itself is pinned elsewhere)
    preds = ["[[8, 16]]", "[[150, 120]]"]          # a plain window; start >= end -> end = duration (blip2_mr.py:1133-1134)
    moments, rel_loc = model.get_relevant_frames(samples, preds, NFA)
    idx_loc = [[int((video[b] - rel_loc[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(NFA)] for b in range(Bv)]
    preds2 = ["[[-1, -1]]", "[[30, 400], [1, 2]]"]  # no window -> whole video; end beyond the duration is clipped, first window taken
    moments2, rel_loc2 = model.get_relevant_frames(samples, preds2, NFA)
    idx_loc2 = [[int((video[b] - rel_loc2[b, j]).flatten(1).abs().sum(1).argmin()) for j in range(NFA)] for b in range(Bv)]
    # forward of the answerer on the localizer-selected frames (what forward_QA does after stage 1)
    model.use_localizer = True
    model.generate = lambda s, **k: {"prediction": preds}
    out_loc = model(dict(samples))
    loss_loc = out_loc["loss"].detach().clone()
    model.use_localizer = False
    # ---- videoQA_generate (uniform): HF generate on the answerer, second step's option logits
    ans = {}
    try:
        orig_gen = model.answerer_model.generate

        def gen(*a, **k):
            r = orig_gen(*a, **k)
            ans["scores1"] = r.scores[1].detach().clone()
            ans["seq"] = r.sequences.clone()
            return r

        model.answerer_model.generate = gen
        o = model.videoQA_generate(dict(samples))
        ans["pred"] = list(o["output_text"])
        print("videoQA_generate:", o["output_text"], o["answer"], o["qid"], o["relevant_moments"])
    except Exception as e:  # noqa
        print("videoQA_generate through HF generate failed in this container:", type(e).__name__, e)
    extra = {}
    if "scores1" in ans:
        extra = dict(gen_scores1_options=ans["scores1"][:, [71, 272, 205, 309, 262]], gen_sequences=ans["seq"], gen_pred=np.array(ans["pred"]))
    MG.save("mr_tiny_qa", man, video=video, timestamps=timestamps, duration=duration,
            loss_uniform=loss_uniform, inputs_embs=uni["embs"], inputs_atts=uni["atts"], labels=uni["labels"], dec_mask=uni["dec_mask"],
            logits_sub=uni["logits"][..., ::64], logits_lse=torch.logsumexp(uni["logits"], -1),
            idx_uniform=np.array(idx_uniform), idx_loc=np.array(idx_loc), idx_loc2=np.array(idx_loc2), loss_loc=loss_loc,
            strings_json=np.frombuffer(json.dumps(dict(qa_input=qa_input, qa_output=qa_output, question_id=samples["question_id"], preds=preds, preds2=preds2,
                                                       moments=[[float(x) for x in m] for m in moments], moments2=[[float(x) for x in m] for m in moments2],
                                                       nfa=NFA)).encode(), dtype=np.uint8), **extra)
    print("loss uniform", float(loss_uniform), "loss with given localizer windows", float(loss_loc), "idx", idx_uniform, idx_loc, idx_loc2, moments, moments2)


if __name__ == "__main__":
    main()
