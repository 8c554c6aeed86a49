"""Fit t(K) = a + b*K for one (M, N, cfg): a = per-launch prologue/epilogue cost, b = main-loop rate.
python tools/gemm_ksweep.py M N cfg [act]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
M, N, cfg = (int(x) for x in sys.argv[1:4])
act = sys.argv[4] if len(sys.argv) > 4 else "none"
dev = torch.device("cuda:0")
res = []
for K in (704, 1408, 2816, 5632):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(tile_cfg=cfg)
    if act == "gelu":
        kw.update(bias=bias, act=1)
    for _ in range(5):
        ops.gemm(a, w, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.gemm(a, w, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    res.append((K, t))
    print(f"K={K} {t:.1f} us  {2*M*N*K/t/1e6:.0f} TF")
(k0, t0), (k1, t1) = res[1], res[3]
b = (t1 - t0) / (k1 - k0)
print(f"cfg{cfg} act={act}: per-64-K-step {b*64:.2f} us/launch-wide, intercept {t0 - b*k0:.1f} us")
