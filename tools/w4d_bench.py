"""cfg 13 (epilogue after each tile) vs cfg 15 (deferred epilogue in the next tile's MFMA shadow) at the ViT qkv / fc1 shapes"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev = torch.device("cuda:0")
for name, M, N, K, act in [("vit_fc1 bias+GELU", 15420, 6144, 1408, 1), ("vit_qkv bias", 15420, 4224, 1408, 0), ("sq8192", 8192, 8192, 8192, 0)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = dict(shape=name)
    for rep in range(2):   # interleaved A/B
        for cfg in (13, 15):
            t = timeit(lambda: ops.gemm(a, w, out, bias=bias, act=act, tile_cfg=cfg))
            row.setdefault(f"cfg{cfg}_us", []).append(round(t, 1))
    fl = 2.0 * M * N * K
    row["cfg13_TF"] = round(fl / min(row["cfg13_us"]) / 1e6, 1)
    row["cfg15_TF"] = round(fl / min(row["cfg15_us"]) / 1e6, 1)
    row["frac15"] = round(row["cfg15_TF"] / 2500.0, 3)
    print(json.dumps(row), flush=True)
