"""t(K) = a + b K for the T5 residual GEMM shape (M = 2012, N = 2048, fp32 out + residual [+ dropout]):  python tools/gemm_ksweep_t5.py cfg [drop]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
cfg = int(sys.argv[1]); drop_on = len(sys.argv) > 2
dev = torch.device("cuda:0")
M, N = 2012, 2048
seed = torch.tensor([7], dtype=torch.int32, device=dev)
res = []
for K in (1024, 2048, 4096, 8192):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    x = torch.randn(M, N, device=dev)
    kw = dict(residual=x, tile_cfg=cfg)
    if drop_on: kw["drop"] = ops.Dropout(seed, 5, 0.1)
    for _ in range(5): ops.gemm(a, w, x, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): ops.gemm(a, w, x, **kw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 40 * 1e3
    res.append((K, t)); print(f"K={K} {t:.1f} us {2*M*N*K/t/1e6:.0f} TF")
(k0, t0), (k1, t1) = res[1], res[3]
b = (t1 - t0) / (k1 - k0)
print(f"cfg{cfg} drop={drop_on}: {b*64:.3f} us per K-tile, intercept {t0 - b*k0:.1f} us")
