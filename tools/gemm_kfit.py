"""time(K) at fixed M,N for both tile configs: slope = per-K-tile cost, intercept = fixed per-launch (prologue + epilogue) cost."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mrblip import ops
from gemm_bench import timeit
dev = torch.device("cuda:0")
for (M, N) in [(15420, 6144), (15420, 1408), (2012, 6144)]:
    for out_dt in (torch.bfloat16, torch.float32):
        for cfg in (1, 2):
            ts = {}
            for K in (64, 704, 1408, 2816, 5632):
                a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
                out = torch.empty(M, N, dtype=out_dt, device=dev)
                ts[K] = timeit(lambda: ops.gemm(a, w, out, tile_cfg=cfg)) * 1e6
            slope = (ts[5632] - ts[1408]) / ((5632 - 1408) / 64)
            print(json.dumps(dict(M=M, N=N, out=str(out_dt)[6:], cfg=cfg, us={k: round(v, 1) for k, v in ts.items()}, us_per_ktile=round(slope, 2),
                                  fixed_us=round(ts[1408] - slope * 22, 1), TF_slope=round(2.0 * M * N * 64 / slope / 1e6, 1))), flush=True)
