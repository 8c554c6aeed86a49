"""One-off (build container, ~40 GB RAM, minutes): the fp32 ORACLE at the benched size C2 against the reference-generated golden
tests/golden/mr_c2.npz (make_golden_c2.py) — the pin of the oracle at Flan-T5-XL width / 24 + 24 layers / S ~ 2000.  Too slow for the
CPU suite (which pins the oracle at real ViT depth through mr_c1.npz); the numbers it prints are recorded in DESIGN.md §2.
    python tests/golden/check_oracle_c2.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), HERE):
    sys.path.insert(0, p)

from util import load_golden, relerr  # noqa: E402
from weights import seeded_array  # noqa: E402
from mrblip import prompt as P  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402
from oracle import mrblip_oracle as O  # noqa: E402

CFG = dict(
    vit=dict(embed_dim=1408, depth=39, num_heads=16, img=224, patch=14),
    qf=dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=12, cross_attention_freq=2, num_query_token=32),
    t5=dict(d_model=2048, d_kv=64, d_ff=5120, num_layers=24, num_decoder_layers=24, num_heads=32, vocab_size=32128, num_buckets=32,
            max_distance=128, eps=1e-6),
)


def main():
    g = load_golden("mr_c2")
    st = g["strings"]
    t0 = time.time()
    sd = {k: torch.from_numpy(seeded_array(k, s, wscale=st["wscale"], fast=True)) for k, s in g["manifest"]}
    for k in ("t5_proj.weight", "t5_proj.bias", "ln_vision.weight", "ln_vision.bias"):
        sd[k].requires_grad_(True)
    T = int(st["T"])
    samples = dict(video=torch.from_numpy(seeded_array("c2.input.video", (1, T, 3, 224, 224), std=1.0, fast=True)),
                   timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]), query_prompt=st["query_prompt"],
                   task_prompt=st["task_prompt"], video_prompt_end=st["video_prompt_end"], relevant_windows=st["relevant_windows"])
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    orc = O.Oracle(sd, CFG)
    t1 = time.time()
    out = orc.forward_mr(tok, samples, repl)
    t2 = time.time()
    out["loss"].backward()
    t3 = time.time()
    print("oracle C2: weights %.0f s, forward %.1f s, backward %.1f s" % (t1 - t0, t2 - t1, t3 - t2))
    assert np.array_equal(out["inputs_atts"].numpy(), g["inputs_atts"]) and np.array_equal(out["labels"].numpy(), g["labels"])
    rows = [("loss (rel)", abs(out["loss"].item() - float(g["loss"])) / abs(float(g["loss"]))),
            ("inputs_embeds", relerr(out["inputs_embs"].detach()[:, ::4, ::16], g["inputs_embs_sub"])),
            ("enc_out", relerr(out["enc"].detach()[:, ::4, ::16], g["enc_sub"])),
            ("logits", relerr(out["logits"].detach()[..., ::64], g["logits_sub"])),
            ("logits_lse", relerr(torch.logsumexp(out["logits"].detach(), -1), g["logits_lse"])),
            ("grad t5_proj.weight", relerr(sd["t5_proj.weight"].grad[::16, ::4], g["grad__t5_proj__weight"])),
            ("grad t5_proj.bias", relerr(sd["t5_proj.bias"].grad, g["grad__t5_proj__bias"])),
            ("grad ln_vision.weight", relerr(sd["ln_vision.weight"].grad, g["grad__ln_vision__weight"])),
            ("grad ln_vision.bias", relerr(sd["ln_vision.bias"].grad, g["grad__ln_vision__bias"]))]
    for n, v in rows:
        print("oracle-fp32 vs reference golden at C2: %-24s %.3e" % (n, v))
    assert all(v < 2e-4 for _, v in rows), rows


if __name__ == "__main__":
    main()
