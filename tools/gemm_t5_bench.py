"""T5-encoder residual GEMMs (fp32 out + residual, M = 2012 rows) across tile configs:  python tools/gemm_t5_bench.py [cfg ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
cfgs = [int(x) for x in sys.argv[1:]] or [4, 12, 13, 14, 2, 5]
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "2012"))
shapes = [(2048, 2048), (2048, 5120), (2048, 6144), (2048, 10240), (5120, 2048), (6144, 2048)]
for N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ref = None
    line = f"N={N:5d} K={K:5d}:"
    for c in cfgs:
        try:
            for _ in range(3):
                ops.gemm(a, w, out, residual=res, tile_cfg=c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, out, residual=res, tile_cfg=c)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e3
            if ref is None:
                ref = out.clone()
            err = (out - ref).abs().max().item()
            line += f"  cfg{c} {t:6.1f}us {2*M*N*K/t/1e6:5.0f}TF" + ("" if err == 0 else f" (maxdiff {err:.2e})")
        except Exception as e:
            line += f"  cfg{c} ERR {str(e)[:40]}"
    print(line)
