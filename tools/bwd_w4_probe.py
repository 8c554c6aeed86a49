"""What a K-SPLIT of the 4-wave kernel would buy for the encoder backward's [2012 x 2048] input-gradient GEMMs (64 tiles of 256x256: a
quarter of the chip): emulated as ONE product with 4 x the rows and K / 4 (the same 256 units of 256x256xK/4, fp32 out), cold weights,
against the generic tile (cfg 4) on the real shape; plus the wo backward [2012 x 5120 x 2048] (cfg 14: 216 tiles, one round) and the
rank-8 add-on / reduction launches the split would need."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
bf16, f32 = torch.bfloat16, torch.float32


def timeit(fn, n=48):
    for i in range(4):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K, ks, odt in (("wi bwd", 2048, 10240, 4, f32), ("qkv bwd", 2048, 6144, 4, f32), ("qkv bwd", 2048, 6144, 3, f32), ("o bwd", 2048, 2048, 4, f32),
                            ("wo fwd", 2048, 5120, 4, f32), ("o fwd", 2048, 2048, 2, f32)):
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(24)]
    out = torch.empty(M, N, dtype=odt, device=dev)
    line = f"{name:8s} N={N:5d} K={K:5d}:"
    for c in (4, 2, 8):
        t = timeit(lambda i: ops.gemm(a, ws[i % 24], out, tile_cfg=c))
        line += f"  cfg{c} {t:6.1f}us"
    a4 = torch.randn(M * ks, K // ks, device=dev).bfloat16()
    w4 = [(torch.randn(N, K // ks, device=dev) * 0.05).bfloat16() for _ in range(24 * ks)]
    out4 = torch.empty(M * ks, N, dtype=odt, device=dev)
    for c in (13, 14):
        t = timeit(lambda i: ops.gemm(a4, w4[i % len(w4)], out4, tile_cfg=c))
        line += f"  split{ks} emulated cfg{c} {t:6.1f}us"
    print(line, flush=True)
for name, N, K in (("wo bwd", 5120, 2048), ("wi fwd", 10240, 2112)):
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(24)]
    out = torch.empty(M, N, dtype=bf16, device=dev)
    line = f"{name:8s} N={N:5d} K={K:5d}:"
    for c in (2, 8, 13, 14):
        t = timeit(lambda i: ops.gemm(a, ws[i % 24], out, tile_cfg=c))
        line += f"  cfg{c} {t:6.1f}us"
    print(line, flush=True)
# the add-ons
seed = torch.zeros(4, dtype=torch.int32, device=dev)
dx = torch.randn(M, 2048, device=dev)
for R in (8, 24):
    G = torch.randn(M, 64, device=dev).bfloat16()
    A = torch.randn(R, 2048, device=dev).bfloat16()
    t = timeit(lambda i: ops.lora_dx_add(dx, G[:, :R], A, drop=None))
    print(f"lora_dx_add fp32 [2012 x 2048] R={R}: {t:.1f} us")
dxb = torch.randn(M, 5120, device=dev).bfloat16()
G = torch.randn(M, 64, device=dev).bfloat16()
A = torch.randn(8, 5120, device=dev).bfloat16()
t = timeit(lambda i: ops.lora_dx_add(dxb, G[:, :8], A, drop=None))
print(f"lora_dx_add bf16 [2012 x 5120] R=8: {t:.1f} us")
parts = torch.randn(4, M, 2048, device=dev)
o = torch.empty(M, 2048, device=dev)
t = timeit(lambda i: torch.sum(parts, 0, out=o))
print(f"torch.sum of 4 fp32 parts [2012 x 2048]: {t:.1f} us")
