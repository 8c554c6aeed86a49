"""The four ViT-g GEMMs of one block with their real epilogues (60 frames: M = 15420 rows), stand-alone, HIP events:
qkv (bias, bf16 out), proj (bias + fp32 residual in place), fc1 (bias + GELU, bf16 out), fc2 (bias + fp32 residual in place).  [--reserve N]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M, D, F = int(os.environ.get("M", "15420")), 1408, 6144
torch.manual_seed(0)
h = torch.randn(M, D, device=dev).bfloat16(); f = torch.randn(M, F, device=dev).bfloat16()
x = torch.randn(M, D, device=dev)
cases = [("qkv", h, 3 * D, D, None, 0, torch.bfloat16), ("proj", h, D, D, x, 0, torch.float32), ("fc1", h, F, D, None, 1, torch.bfloat16), ("fc2", f, D, F, x, 0, torch.float32)]
for name, a, N, K, res, act, dt in cases:
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16(); b = torch.randn(N, device=dev)
    out = res if res is not None else torch.empty(M, N, dtype=dt, device=dev)
    fn = lambda: ops.gemm(a, w, out, bias=b, residual=res, act=act)
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e3
    print(f"{name:5s} M={M} N={N:5d} K={K:5d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s")
