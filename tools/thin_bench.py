"""The four GEMMs of a T5-XL encoder layer's forward at M = 2012 on cold weights (a rotation of weight sets larger than the Infinity Cache):
LoRA "down" product as a launch of its own + GEMM, against the GEMM with the thin role (and with prefetch workgroups for the next set).
python tools/thin_bench.py      (MRB_GEMM_ROLES_LAST=0: role workgroups always in front of the tiles)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
bf = lambda t: t.bfloat16()
seed = torch.tensor([3], dtype=torch.int32, device=dev)


def case(name, N, K, nad, gated=False, f32=False, tout=False, cfg=0, pfb=32):
    rows = 2 * N if gated else N
    R = 8 * nad
    nset = -(-640 * 2**20 // (rows * K * 2))
    a = bf(torch.randn(M, K, device=dev)); w0 = bf(torch.randn(rows, K, device=dev) * 0.03)
    acat = bf(torch.randn(R, K, device=dev) * 0.05)
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev); wext = bf(torch.randn(rows, 64, device=dev) * 0.05)
    ws = [w0] + [w0.clone() for _ in range(nset - 1)]
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if f32 else None
    h = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev) if gated else None
    drop = ops.Dropout(seed, 7, 0.1) if (f32 or gated) else None
    idrop = ops.Dropout(seed, 9, 0.05)
    touts = [torch.zeros(1, 32, 64, ops.rup32(M), dtype=torch.bfloat16, device=dev) for _ in range(3)] if tout else None

    def loop(mode, n=3 * 24):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            if rep == 1: s.record()
            for i in range(n):
                w = ws[i % nset]
                if "pf" in mode: ops.gemm_prefetch(ws[(i + 1) % nset], pfb)
                if "thin" not in mode: ops.lora_rows(a, acat, u, K, drop=idrop)
                ops.gemm(a, w, out, aext=u, wext=wext, residual=res, out2=h, gated=gated, drop=drop, tile_cfg=cfg, tout=touts, t_rows=M,
                         thin=(acat, K, idrop) if "thin" in mode else None)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    print(f"{name:22s} N={N:5d} K={K:5d} R={R:2d}:  own launch {loop('own'):6.1f}   thin role {loop('thin'):6.1f}   own + prefetch {loop('own pf'):6.1f}   thin + prefetch {loop('thin pf'):6.1f} us", flush=True)


case("qkv (+ q/k/v^T, cfg 8)", 6144, 2048, 3, tout=True)
case("qkv (+ q/k/v^T, cfg 2)", 6144, 2048, 3, tout=True, cfg=2)
case("qkv (+ q/k/v^T, cfg 4)", 6144, 2048, 3, tout=True, cfg=4)
case("o (fp32 residual)", 2048, 2048, 1, f32=True, pfb=128)
case("wi (gated)", 5120, 2048, 2, gated=True)
case("wo (fp32 residual)", 2048, 5120, 1, f32=True)
