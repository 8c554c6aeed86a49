"""The encoder's 4-wave-kernel products in a loop (for rocprofv3 --pmc):  python tools/ksplit_one.py <wi_bwd|qkv_bwd|wo_bwd|qkv_fwd|generic_wi_bwd> [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
which = sys.argv[1]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
M = 2012
shapes = dict(wi_bwd=(2048, 10240, 4, 13, torch.float32), qkv_bwd=(2048, 6144, 4, 13, torch.float32), wo_bwd=(5120, 2048, 1, 14, torch.bfloat16))
if which in shapes:
    N, K, ks, cfg, dt = shapes[which]
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(8)]
    g = torch.randn(M, 64, device=dev).bfloat16()
    at = (torch.randn(N, 64, device=dev) * 0.05).bfloat16()
    parts = torch.empty(ks + 1, M, N, dtype=dt, device=dev)
    for i in range(it):
        ops.gemm_ksplit(a, ws[i % 8], parts, K, ks, ext=(g, at), tile_cfg=cfg)
elif which == "qkv_fwd":
    a = torch.randn(M, 2112, device=dev).bfloat16()
    ws = [(torch.randn(6144, 2112, device=dev) * 0.05).bfloat16() for _ in range(8)]
    out = torch.empty(M, 6144, dtype=torch.bfloat16, device=dev)
    for i in range(it):
        ops.gemm(a, ws[i % 8], out, tile_cfg=14)
else:
    a = torch.randn(M, 10240, device=dev).bfloat16()
    ws = [(torch.randn(2048, 10240, device=dev) * 0.05).bfloat16() for _ in range(8)]
    out = torch.empty(M, 2048, device=dev)
    for i in range(it):
        ops.gemm(a, ws[i % 8], out, tile_cfg=4)
torch.cuda.synchronize()
