"""Micro-benchmark of mrblip_gemm_bf16 at the hot-path shapes (HIP events), next to torch.matmul (hipBLASLt) as a yardstick."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402

SHAPES = [  # (name, M, N, K)
    ("vit_qkv", 15420, 4224, 1408), ("vit_proj", 15420, 1408, 1408), ("vit_fc1", 15420, 6144, 1408), ("vit_fc2", 15420, 1408, 6144),
    ("t5_qkv", 2023, 6144, 2048), ("t5_o", 2023, 2048, 2048), ("t5_wi", 2023, 10240, 2048), ("t5_wo", 2023, 2048, 5120),
    ("t5_dx_wi", 2023, 2048, 10240), ("t5_dx_qkv", 2023, 2048, 6144),
    ("qf_kv", 15420, 1536, 1408), ("qf_q", 1920, 768, 768), ("qf_qkv", 1920, 2304, 768), ("qf_fc1", 1920, 3072, 768), ("qf_fc2", 1920, 768, 3072), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
    ("dec_q", 12, 2048, 2048), ("dec_wi", 12, 10240, 2048), ("lm_head", 12, 32128, 2048), ("dec_wo", 12, 2048, 5120),
    ("dec_dx_wi", 12, 2048, 10240), ("dec_down", 12, 24, 2048), ("dec_g_wi", 12, 8, 10240), ("enc_down", 2012, 24, 2048), ("enc_g_wi", 2012, 8, 10240),
]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    res = []
    for name, M, N, K in SHAPES:
        if os.environ.get("ONLY") and not name.startswith(os.environ["ONLY"]):
            continue
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        row = dict(name=name, M=M, N=N, K=K)
        fl = 2.0 * M * N * K
        cfgs = [3] if (M <= 64 or N <= 64) else [int(c) for c in os.environ.get("CFGS", "2,4,8").split(",")]
        for cfg in cfgs:
            t = timeit(lambda: ops.gemm(a, w, out, tile_cfg=cfg))
            row[f"cfg{cfg}_us"] = round(t * 1e6, 1)
            row[f"cfg{cfg}_TF"] = round(fl / t / 1e12, 1)
        t = timeit(lambda: torch.matmul(a, w.t(), out=out))
        row["torch_us"] = round(t * 1e6, 1)
        row["torch_TF"] = round(fl / t / 1e12, 1)
        if M <= 64 or N <= 64:
            row["cfg3_GBs"] = round((N + M) * K * 2 / (row["cfg3_us"] * 1e-6) / 1e9, 1)
        print(json.dumps(row), flush=True)
        res.append(row)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
