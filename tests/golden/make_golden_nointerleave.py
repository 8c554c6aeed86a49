"""Golden for interleave_data=False (blip2_mr.py:783-822, the constructor's default at :82; the shipped configs set True): the
reference's forward_mr at tiny dimensions with the prompt [ video_prompt tokens | all frame tokens | video_prompt_end | text ] ->
mr_tiny_nointerleave.npz / mr_tiny_nointerleave_floats.npz (encoder input, mask, labels, loss, the video_prompt strings).
Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nointerleave.py"""
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


def main():
    R = ref_shim.install(FixtureTokenizer, MG.TINY)
    mr = R["mr"]
    for name, fmt, ts, dur, wins in (
            ("mr_tiny_nointerleave", "seconds_integers", [[2.5, 22.49, 39.0], [7.0, 105.2, 187.6]], [45.0, 250.0], ["[[8, 16]]", "[[0, 4], [22, 150]]"]),
            ("mr_tiny_nointerleave_floats", "seconds_floats", [[2.5, 22.49, 39.0], [7.0, 105.2, 187.65]], [45.0, 250.5], ["[[8.5, 16.25]]", "[[0.0, 4.0], [22.0, 150.0]]"])):
        torch.manual_seed(0)
        g = torch.Generator().manual_seed(99)
        model = mr.BLIP2_MR(img_size=MG.IMG, vit_precision="fp32", num_query_token=MG.NQ, t5_model="google/flan-t5-xl",
                            input_time_format=fmt, interleave_data=False, frame_token_aggregation=False, task="qformer_freeze_lora")
        model.eval()
        man = MG.load_seeded(model)
        video = torch.randn(2, 3, 3, MG.IMG, MG.IMG, generator=g)
        samples = dict(video=video, timestamps=torch.tensor(ts, dtype=torch.float32), duration=torch.tensor(dur),
                       query_prompt=["Query: a man opens the red door\n", "Query: the dog runs\n"],
                       task_prompt=["Given the video and the query, find the relevant windows.\nRelevant windows: "] * 2,
                       video_prompt_end=["<extra_id_0>"] * 2, relevant_windows=wins)
        cap = {}
        orig = model.prompt_concatenation

        def pc(*a, **k):
            r = orig(*a, **k)
            cap["embs"], cap["atts"], cap["prompt"] = r[0].detach().clone(), r[1].clone(), r[2]
            return r

        model.prompt_concatenation = pc
        orig_t5 = model.t5_model.forward

        def t5f(*a, **k):
            cap["labels"] = k["labels"].clone()
            return orig_t5(*a, **k)

        model.t5_model.forward = t5f
        out = model(samples)
        MG.save(name, man, video=video, timestamps=samples["timestamps"], duration=samples["duration"], loss=out["loss"], inputs_embs=cap["embs"],
                inputs_atts=cap["atts"], labels=cap["labels"],
                strings_json=np.frombuffer(json.dumps(dict(query_prompt=samples["query_prompt"], task_prompt=samples["task_prompt"],
                                                           video_prompt_end=samples["video_prompt_end"], relevant_windows=samples["relevant_windows"],
                                                           video_prompt=cap["prompt"], time_format=fmt)).encode(), dtype=np.uint8))
        print(name, "loss", float(out["loss"]), "S", cap["embs"].shape[1], cap["prompt"])


if __name__ == "__main__":
    main()
