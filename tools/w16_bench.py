"""One tile config (default 16 = the 16x16x32-MFMA form of the 4-wave tile) at the ViT shapes + 8192^3: time, TFLOP/s, check vs cfg 2.
  python tools/w16_bench.py [cfg]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402

CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tag = "cfg%d" % CFG


def timeit(fn, iters=30, warm=8):
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
torch.manual_seed(0)
row = {"tag": tag}
for name, M, N, K, act, res in [("fc1", 15420, 6144, 1408, 1, False), ("qkv", 15420, 4224, 1408, 0, False), ("fc2", 15420, 1408, 6144, 0, True),
                                ("sq8192", 8192, 8192, 8192, 0, False)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, dtype=torch.float32 if res else torch.bfloat16, device=dev)
    ref = torch.empty_like(out)
    ops.gemm(a, w, ref, bias=bias, act=act, residual=r, tile_cfg=2)
    ops.gemm(a, w, out, bias=bias, act=act, residual=r, tile_cfg=CFG)
    err = ((out.float() - ref.float()).norm() / ref.float().norm()).item()
    ts = [timeit(lambda: ops.gemm(a, w, out, bias=bias, act=act, residual=r, tile_cfg=CFG)) for _ in range(3)]
    row[name] = {"us": [round(t, 1) for t in ts], "TF": round(2.0 * M * N * K / min(ts) / 1e6, 1), "err_vs_cfg2": float("%.2e" % err)}
print(json.dumps(row), flush=True)
