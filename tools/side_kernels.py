"""Launches the HBM-bound side kernels of the train step on the bench's QVH shapes (bench.hbm_kernel_report) plus a CALIBRATION launch of
known traffic, so that a `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass (tools/pmc_side.sh) can attribute HBM-side bytes to
each of them.  Also runs the four ViT GEMM shapes once per block (39 x) for the MFMA-busy pass."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrblip import ops, prompt as P  # noqa: E402
from mrblip.engine import EngineConfig, MrBlipEngine, RandomSource  # noqa: E402
from mrblip.tokenizer import FixtureTokenizer  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "side"
if mode == "side":
    cfg = EngineConfig(vit_depth=1, t5_layers=1, t5_dec_layers=1, qf_layers=2)   # shapes only: one layer of each tower keeps the set-up short
    eng = MrBlipEngine(cfg, RandomSource(dev, seed=1), dev)
    tok = FixtureTokenizer()
    repl = P.annoying_replacement_dict(P.find_annoying_numbers(tok, 200)[0])
    samples = bench.synthetic_samples(1, 60, 150.0, dev, 1234)
    layout = P.build_layout(tok, samples, repl, 32, T=60)
    # calibration: a plain fp32 -> fp32 copy of 86.8 MB (cast_drop_kernel without dropout): 86.8 MB read, 86.8 MB written
    x = torch.randn(15420, 1408, device=dev)
    y = torch.empty_like(x)
    for _ in range(10):
        ops.cast_dropout(x, out_f32=y)
    torch.cuda.synchronize()
    rows = bench.hbm_kernel_report(eng, samples["video"], layout, iters=10)
    print(json.dumps(dict(calibration_bytes=dict(read=x.numel() * 4, write=x.numel() * 4, grid_hint="cast_drop_kernel, first 11 launches"), rows=rows)))
else:  # "gemm": the four frozen-ViT GEMMs as the engine launches them (39 blocks), all CUs
    M, D, F = 15420, 1408, 6144
    h = torch.randn(M, D, device=dev).bfloat16()
    f = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
    qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    x = torch.randn(M, D, device=dev)
    w1, w2 = (torch.randn(F, D, device=dev) * 0.03).bfloat16(), (torch.randn(D, F, device=dev) * 0.02).bfloat16()
    wq, wp = (torch.randn(3 * D, D, device=dev) * 0.03).bfloat16(), (torch.randn(D, D, device=dev) * 0.03).bfloat16()
    b1, b2, bq = torch.randn(F, device=dev), torch.randn(D, device=dev), torch.randn(3 * D, device=dev)
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 39):
        ops.gemm(h, wq, qkv, bias=bq)
        ops.gemm(h, wp, x, bias=b2, residual=x)
        ops.gemm(h, w1, f, bias=b1, act=1)
        ops.gemm(f, w2, x, bias=b2, residual=x)
    torch.cuda.synchronize()
    print("ok")
