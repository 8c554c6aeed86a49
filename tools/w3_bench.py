"""cfg 13 (two stages) vs cfg 17 (third W stage) of the 4-wave 256x256 GEMM: the four ViT GEMMs at 60 frames and 8192^3, stand-alone."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M, D, F = 15420, 1408, 6144
torch.manual_seed(0)
h = torch.randn(M, D, device=dev).bfloat16(); f = torch.randn(M, F, device=dev).bfloat16()
x = torch.randn(M, D, device=dev)
big = torch.randn(8192, 8192, device=dev).bfloat16()
cases = [("qkv", h, 3 * D, D, None, 0, torch.bfloat16), ("proj", h, D, D, x, 0, torch.float32), ("fc1", h, F, D, None, 1, torch.bfloat16), ("fc2", f, D, F, x, 0, torch.float32),
         ("8192^3", big, 8192, 8192, None, 0, torch.bfloat16)]
for reserve in (0, 64):
    for name, a, N, K, res, act, dt in cases:
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16(); b = torch.randn(N, device=dev)
        out = res if res is not None else torch.empty(a.shape[0], N, dtype=dt, device=dev)
        line = f"reserve {reserve:2d} {name:7s} M={a.shape[0]} N={N:5d} K={K:5d}:"
        for rep in range(2):
            for cfg in (13, 17):
                fn = lambda: ops.gemm(a, w, out, bias=b, residual=res, act=act, tile_cfg=cfg, cu_reserve=reserve)
                for _ in range(3): fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20): fn()
                e.record(); torch.cuda.synchronize()
                t = s.elapsed_time(e) / 20 * 1e3
                line += f"  cfg{cfg} {t:7.1f} us {2.0 * a.shape[0] * N * K / t / 1e6:5.0f} TF"
        print(line, flush=True)
