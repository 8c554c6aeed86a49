"""Energy ledger of the dominant kernel: the ViT fc1 GEMM [15420 x 6144 x 1408] in a loop, one line per run:
average launch time; run it under tools/pwr_probe.sh to read board power and sclk beside it -> joules per launch.
   python tools/w4_energy.py <act: 1 = bias + GELU (the product form), 0 = bias only> [seconds=7] [MRBLIP_LIB selects an ablation build]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
act = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
dev = torch.device("cuda:0")
M, N, K = 15420, 6144, 1408
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
b = torch.randn(N, device=dev) * 0.1
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(20):
    ops.gemm(a, w, out, bias=b, act=act, tile_cfg=13)
torch.cuda.synchronize()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < secs:
    for _ in range(200):
        ops.gemm(a, w, out, bias=b, act=act, tile_cfg=13)
    torch.cuda.synchronize()
    n += 200
dt = time.perf_counter() - t0
print("fc1 %s lib=%s: %.1f us per launch, %.0f TFLOP/s" % ("bias+GELU" if act else "bias only", os.environ.get("MRBLIP_LIB", "default"), dt / n * 1e6, 2.0 * M * N * K * n / dt / 1e12))
