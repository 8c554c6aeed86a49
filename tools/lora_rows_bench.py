"""The LoRA thin products: row kernel (csrc/lora.hip) vs the MFMA skinny kernel it replaces, and the fused RMSNorm + down launch."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    seed = torch.tensor([5], dtype=torch.int32, device=dev)
    rows = []
    for name, M, K, R, p in [("enc down qkv", 2012, 2048, 24, 0.05), ("enc down o", 2012, 2048, 8, 0.05), ("enc down wi", 2012, 2048, 16, 0.05),
                             ("enc down wo", 2012, 5120, 8, 0.05), ("enc g qkv", 2012, 6144, 24, 0.0), ("enc g wi", 2012, 10240, 16, 0.0),
                             ("enc g o", 2012, 2048, 8, 0.0), ("dec down qkv", 8, 2048, 24, 0.05), ("dec down wo", 8, 5120, 8, 0.05),
                             ("dec g wi", 8, 10240, 16, 0.0), ("dec g lm_head", 8, 32128, 8, 0.0), ("b4 down qkv", 8048, 2048, 24, 0.05)]:
        x = torch.randn(M, K, device=dev).bfloat16()
        a = (torch.randn(R, K, device=dev) * 0.05).bfloat16()
        u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev)
        d = ops.Dropout(seed, 3, p) if p > 0 else None
        seg = None
        if name.startswith(("enc g", "dec g")) and R > 8:
            w = K // (R // 8)
            seg = [v for j in range(R // 8) for v in (j * w, (j + 1) * w)]
        t_new = timeit(lambda: ops.lora_rows(x, a, u, K, drop=d, seg=seg))
        t_old = timeit(lambda: ops.lora_down(x, a, u, K, drop=d))
        row = dict(name=name, M=M, K=K, R=R, rows_us=round(t_new, 1), skinny_us=round(t_old, 1))
        if K == 2048 and p > 0:
            xf = torch.randn(M, K, device=dev)
            w = torch.ones(K, device=dev)
            xn = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
            row["rmsnorm_us"] = round(timeit(lambda: ops.rmsnorm_fwd(xf, w, 1e-6, out_bf16=xn)), 1)
            row["fused_norm_down_us"] = round(timeit(lambda: ops.rmsnorm_lora_fwd(xf, w, 1e-6, xn, a, u, drop=d)), 1)
        print(json.dumps(row), flush=True)
        rows.append(row)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "lora_rows_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
