"""What a ONE-ROUND tiling of the T5 qkv projection would buy: [2012 x 6144 x 2048], bf16 out, plain epilogue, across tile configs
(8 = 16-wave 256x256: 192 tiles on 256 CUs, the product's choice; 13 = 4-wave 256x256; 14 = 4-wave 256x192: 256 tiles = one full round;
1 = 8-wave 256x256; 2 = 128x128; 4 = 64x128), weights rotated through COLD sets larger than the Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
for N, K in ((6144, 2048), (10240, 2048), (2048, 2048), (2048, 5120)):
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(24)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    line = f"N={N:5d} K={K:5d}:"
    for c in (8, 13, 14, 1, 2, 4):
        try:
            for i in range(4):
                ops.gemm(a, ws[i], out, tile_cfg=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(48):
                ops.gemm(a, ws[i % 24], out, tile_cfg=c)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 48 * 1e3
            line += f"  cfg{c} {t:6.1f}us {2*M*N*K/t/1e6:5.0f}TF"
        except Exception as e:
            line += f"  cfg{c} ERR {str(e)[:30]}"
    print(line, flush=True)
