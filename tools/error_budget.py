"""Where does the product path's ~1e-2 logits error come from?    CPU experiment on BASELINE configs[0]
("C1": real 39-block ViT-g, bert-base Q-Former, T5-base dims, 4 frames): the oracle's bf16-operand emulation — operands of every GEMM
and attention product rounded to bf16 where the HIP path stores bf16, fp32 accumulate — is switched on for ONE tower at a time, for all
of them, and for all but the decoder's last layer + lm_head (what split-bf16 "hi + lo" operands on those last small GEMMs would buy),
and logits / loss are compared with the oracle's own fp32 run.  Runs in the build container (no GPU):
    python tools/error_budget.py > profiles/r03_error_budget_c1.txt
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mr-blip_amd"), ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

from util import load_golden, relerr  # noqa: E402
from weights import seeded_state_dict, seeded_array  # noqa: E402
from mrblip import prompt as P  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402
from oracle import mrblip_oracle as O  # noqa: E402
from test_fullsize_gpu import C1_CFG  # noqa: E402


def main():
    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    s = g["strings"]
    samples = dict(video=torch.from_numpy(seeded_array("c1.input.video", (1, 4, 3, 224, 224), std=1.0, fast=True)),
                   timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]), query_prompt=s["query_prompt"],
                   task_prompt=s["task_prompt"], video_prompt_end=s["video_prompt_end"], relevant_windows=s["relevant_windows"])
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    nd = C1_CFG["t5"]["num_decoder_layers"]

    def run(towers):
        orc = O.Oracle(sd, C1_CFG, emu_bf16=towers is not False)
        orc.emu_towers = None if towers in (None, False) else set(towers)
        with torch.no_grad():
            return orc.forward_mr(tok, samples, repl)

    t0 = time.time()
    ref = run(False)
    print("# logits / loss error of the oracle with bf16-operand emulation in the named towers, against its own fp32 run (C1, %d decoder layers)" % nd)
    print("# reference golden check: oracle-fp32 logits vs reference-fp32 %.2e" % relerr(ref["logits"][..., ::64], g["logits_sub"]))
    print("%-58s %12s %12s %12s" % ("bf16 operands in", "logits relL2", "enc relL2", "loss rel"))
    everything = ["vit", "qf", "proj", "enc", "dec", "head"]
    cases = [("ViT only (39 blocks)", ["vit"]), ("Q-Former only", ["qf"]), ("t5_proj only", ["proj"]), ("T5 encoder only", ["enc"]),
             ("T5 decoder only", ["dec"]), ("lm_head only", ["head"]),
             ("decoder last layer + lm_head only", ["dec:%d" % (nd - 1), "head"]),
             ("everything (= the product path's rounding points)", None),
             ("everything but lm_head (hi+lo on lm_head)", ["vit", "qf", "proj", "enc", "dec"]),
             ("everything but decoder last layer + lm_head (hi+lo there)", ["vit", "qf", "proj", "enc"] + ["dec:%d" % i for i in range(nd - 1)]),
             ("everything but the T5 decoder + lm_head", ["vit", "qf", "proj", "enc"]),
             ("everything but the ViT", ["qf", "proj", "enc", "dec", "head"])]
    for name, towers in cases:
        out = run(towers)
        print("%-58s %12.3e %12.3e %12.3e" % (name, relerr(out["logits"], ref["logits"]), relerr(out["enc"], ref["enc"]),
                                              abs(out["loss"].item() - ref["loss"].item()) / abs(ref["loss"].item())), flush=True)
    print("# %.0f s on %d threads" % (time.time() - t0, torch.get_num_threads()))


if __name__ == "__main__":
    main()
