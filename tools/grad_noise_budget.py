"""Is the 2-5e-2 gap between the HIP step's gradients and the reference's fp32 autograd ROUNDING, or a logic error hiding under a loose
tolerance?  (VERDICT r4 missing 1 / weak 2.)  CPU experiment on BASELINE configs[0] ("C1": real 39-block ViT-g, bert-base Q-Former,
T5-base dims, 4 frames) with NON-ZERO LoRA in every adapter, the setting of tests/test_fullsize_gpu.py::
test_c1_real_depth_nonzero_lora_gradients_against_oracle_autograd.

The HIP path rounds to bf16 at fixed points: every GEMM / attention operand in the forward AND the dy copies that feed the backward's dX /
dW GEMMs.  Which way a value rounds depends on fp32 bits that differ between any two correct implementations (summation order), so the
gradient of one implementation is ONE DRAW from a distribution around the exact gradient.  This script measures the width of that
distribution with the oracle alone: the same rounding points, but rounding STOCHASTICALLY to one of the two bf16 neighbours (probability
by distance), two independent seeds.  If
    rel(draw_1, fp32)  ~  rel(draw_2, fp32)  ~  rel(draw_1, draw_2) / sqrt(2)  ~  what tests measure for rel(HIP, fp32)
the HIP gradients sit where a correct bf16-operand implementation must sit; a logic error would put HIP outside that band or bias it
(norm ratio, cosine: tests/test_fullsize_gpu.py::_bias_report).
    python tools/grad_noise_budget.py > profiles/r05_grad_noise_budget_c1.txt
"""
import math
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
from test_model_gpu import _peft_sd  # noqa: E402


def stochastic_bf16(gen):
    def fn(x):
        e = torch.frexp(x.detach())[1].float()          # x = m * 2^e, m in [0.5, 1): a bf16 ulp (8 significant bits) is 2^(e - 8)
        u = torch.rand(x.shape, generator=gen) - 0.5
        return torch.where(x == 0, x, (x + u * torch.exp2(e - 8.0)).bfloat16().float())     # (an exact zero has no neighbours to choose from)
    return fn


def main():
    g = load_golden("mr_c1")
    sd = seeded_state_dict(g["manifest"], wscale=g["strings"]["wscale"], fast=True)
    sdl = _peft_sd(sd, lora_std=0.02)
    train = [k for k in sdl if ("lora_" in k) or k.startswith("t5_proj") or k.startswith("ln_vision")]
    for k in train:
        sdl[k].requires_grad_(True)
    s = g["strings"]
    samples = dict(video=torch.from_numpy(seeded_array("c1.input.video", (1, 4, 3, 224, 224), std=1.0, fast=True)),
                   timestamps=torch.from_numpy(g["timestamps"]), duration=torch.from_numpy(g["duration"]), query_prompt=s["query_prompt"],
                   task_prompt=s["task_prompt"], video_prompt_end=s["video_prompt_end"], relevant_windows=s["relevant_windows"])
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    lora_keys = sorted(k for k in train if "lora_" in k)

    def run(emu, grad_round=False, seed=None):
        for k in train:
            sdl[k].grad = None
        orc = O.Oracle(sdl, C1_CFG, emu_bf16=emu, lora=dict(r=8, alpha=8))
        orc.emu_grad = grad_round
        if seed is not None:
            orc.round_fn = stochastic_bf16(torch.Generator().manual_seed(seed))
        out = orc.forward_mr(tok, samples, repl)
        out["loss"].backward()
        return dict(loss=out["loss"].item(), logits=out["logits"].detach().clone(), proj=sdl["t5_proj.weight"].grad.clone(),
                    lnv=sdl["ln_vision.weight"].grad.clone(), lora={k: sdl[k].grad.clone() for k in lora_keys})

    def row(name, a, b):
        num = sum(float((a["lora"][k] - b["lora"][k]).pow(2).sum()) for k in lora_keys)
        den = sum(float(b["lora"][k].pow(2).sum()) for k in lora_keys)
        worst = max(relerr(a["lora"][k], b["lora"][k]) for k in lora_keys)
        print("%-64s %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e" % (name, abs(a["loss"] - b["loss"]) / abs(b["loss"]), relerr(a["logits"], b["logits"]),
                                                                   relerr(a["proj"], b["proj"]), relerr(a["lnv"], b["lnv"]), math.sqrt(num / den), worst), flush=True)

    def norms(name, a, b):
        """signed norm ratios - 1 (a bias shows here; unbiased noise e moves a norm by + e^2 / 2)"""
        def nr(keys):
            return math.sqrt(sum(float(a["lora"][k].double().pow(2).sum()) for k in keys) / sum(float(b["lora"][k].double().pow(2).sum()) for k in keys)) - 1
        ka, kb = [k for k in lora_keys if "lora_A" in k], [k for k in lora_keys if "lora_B" in k]
        print("%-64s |g|/|g_ref| - 1:  t5_proj %+.2e  ln_vision %+.2e  LoRA all %+.2e  lora_A %+.2e  lora_B %+.2e  lm_head A %+.2e B %+.2e" % (
            name, float(a["proj"].double().norm() / b["proj"].double().norm()) - 1, float(a["lnv"].double().norm() / b["lnv"].double().norm()) - 1,
            nr(lora_keys), nr(ka), nr(kb), nr([k for k in ka if "lm_head" in k]), nr([k for k in kb if "lm_head" in k])), flush=True)

    t0 = time.time()
    f32 = run(False)
    rne = run(True)
    rne_g = run(True, grad_round=True)
    d1 = run(True, grad_round=True, seed=101)
    d2 = run(True, grad_round=True, seed=202)
    print("# C1, non-zero LoRA (N(0, 0.02)) in all %d adapters; relative L2 errors; %0.f s on %d threads" % (len(lora_keys) // 2, time.time() - t0, torch.get_num_threads()))
    print("%-64s %10s %10s %10s %10s %10s %10s" % ("comparison", "loss", "logits", "dW t5_proj", "dW ln_vis", "LoRA flat", "LoRA worst"))
    row("bf16 operands, round-to-nearest (tests' emu-oracle) vs fp32", rne, f32)
    row("  + gradients rounded at every linear output   vs fp32", rne_g, f32)
    row("stochastic rounding, seed 101                    vs fp32", d1, f32)
    row("stochastic rounding, seed 202                    vs fp32", d2, f32)
    row("stochastic seed 101 vs seed 202 (two correct draws)", d1, d2)
    row("stochastic seed 101 vs round-to-nearest + grad rounding", d1, rne_g)
    print("# signed norm ratios (round 5: the HIP gradient's norm sits 0.04-0.4 % BELOW the fp32 one in every class — does the bf16-operand oracle's?)")
    norms("bf16 operands, round-to-nearest vs fp32", rne, f32)
    norms("  + gradients rounded at every linear output vs fp32", rne_g, f32)
    norms("stochastic rounding, seed 101 vs fp32", d1, f32)
    norms("stochastic rounding, seed 202 vs fp32", d2, f32)
    print("# HIP step vs fp32 oracle, same model (tests/test_fullsize_gpu.py, profiles/r05_parity_errors.json): loss 2.2e-4, logits 1.2e-2,")
    print("# t5_proj / ln_vision gradients 2.1e-2, all LoRA gradients flat 2.5e-2 (emu-oracle: 1.9e-2), worst single adapter 4.6e-2")


if __name__ == "__main__":
    main()
