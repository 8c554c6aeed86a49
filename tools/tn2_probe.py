"""The 4-wave kernel with 256x128 tiles (tile_cfg 22) for the [2012 x 2048] outputs: 128 tiles, so TWO K ranges fill the chip and the
consumer adds two partial products instead of four.  Emulated as one plain product with 2 x the rows and K / 2 (256 units), cold weights,
beside the 256x256 tile with 4 x the rows and K / 4, and checked against fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
f32 = torch.float32


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


# correctness of the new tile first
a = torch.randn(517, 384, device=dev).bfloat16()
w = (torch.randn(264, 384, device=dev) * 0.1).bfloat16()
parts = torch.empty(3, 517, 264, device=dev)
ops.gemm_ksplit(a, w, parts, 384, 3, tile_cfg=22)
ref = sum(a[:, i * 128:(i + 1) * 128].float() @ w[:, i * 128:(i + 1) * 128].float().t() for i in range(3))
print("256x128 tile, 3 splits: rel err", ((parts.sum(0) - ref).norm() / ref.norm()).item())
for name, N, K in (("wi bwd", 2048, 10240), ("qkv bwd", 2048, 6144), ("wo fwd", 2048, 5120), ("o", 2048, 2048)):
    line = f"{name:8s} N={N} K={K}:"
    for cfg, ks in ((13, 4), (22, 2), (22, 4)):
        a4 = torch.randn(M * ks, K // ks, device=dev).bfloat16()
        w4 = [(torch.randn(N, K // ks, device=dev) * 0.05).bfloat16() for _ in range(24 * ks)]
        out4 = torch.empty(1, M * ks, N, dtype=f32, device=dev)
        t = timeit(lambda i: ops.gemm_ksplit(a4, w4[i % len(w4)], out4, K // ks, 1, tile_cfg=cfg))
        line += f"  cfg{cfg} x{ks} splits {t:6.1f}us"
    print(line, flush=True)
