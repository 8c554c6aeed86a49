"""The Q-Former's GEMM shapes (M = frames x 32 rows) per tile config, cold weights (a ring of 16 weight copies), chip to itself.
   python tools/qf_gemm_bench.py [frames]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
M = F * 32
dev = torch.device("cuda:0")
shapes = [("o / cross q,o  f32+res", 768, 768, True), ("qkv", 2304, 768, False), ("ffn in +gelu", 3072, 768, False), ("ffn out f32+res", 768, 3072, True),
          ("dX ffn-out^T (bf16)", 3072, 768, False), ("dX qkv^T f32+res", 768, 2304, True)]
for name, N, K, f32res in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(16)]
    out = torch.empty(M, N, dtype=torch.float32 if f32res else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if f32res else None
    line = f"{name:26s} [{M} x {N} x {K}]"
    for cfg in (0, 5, 22, 23, 4, 18, 19, 2, 20):
        try:
            for i in range(4):
                ops.gemm(a, ws[i], out, residual=res, tile_cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(48):
                ops.gemm(a, ws[i % 16], out, residual=res, tile_cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            line += f"  cfg{cfg}: {e0.elapsed_time(e1) / 48 * 1e3:6.1f}"
        except Exception as ex:  # noqa: BLE001
            line += f"  cfg{cfg}: n/a"
    print(line, flush=True)
