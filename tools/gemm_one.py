"""One GEMM shape / tile config in a loop (for rocprofv3 --pmc):  python tools/gemm_one.py M N K cfg [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
M, N, K, cfg = (int(x) for x in sys.argv[1:5])
it = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(it):
    ops.gemm(a, w, out, tile_cfg=cfg)
torch.cuda.synchronize()
