"""Would an in-block K split pay for the [2012 x 2048] outputs (o, wo)?  Emulation without writing the kernel: the 128x128 tile on a problem
with TWICE the rows and HALF the K (512 tiles of K / 2: two blocks per CU, the same flops and operand bytes per CU as a two-group split of
256 tiles) against today's 64x128 tile on the real shape.  Cold weights (rotation of sets).   python tools/splitk_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
bf = lambda t: t.bfloat16()
seed = torch.tensor([3], dtype=torch.int32, device=dev)


def run(M, N, K, cfg, n=96):
    nset = -(-640 * 2**20 // (N * K * 2))
    a = bf(torch.randn(M, K, device=dev)); w0 = bf(torch.randn(N, K, device=dev) * 0.03)
    u = bf(torch.randn(M, 64, device=dev)); wext = bf(torch.randn(N, 64, device=dev) * 0.05)
    ws = [w0] + [w0.clone() for _ in range(nset - 1)]
    out = torch.empty(M, N, dtype=torch.float32, device=dev); res = torch.randn(M, N, device=dev)
    drop = ops.Dropout(seed, 7, 0.1)
    res_t = {}
    for mode in ("warm", "cold"):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            if rep == 1: s.record()
            for i in range(n):
                ops.gemm(a, ws[0] if mode == "warm" else ws[i % nset], out, aext=u, wext=wext, residual=res, drop=drop, tile_cfg=cfg)
        e.record(); torch.cuda.synchronize()
        res_t[mode] = s.elapsed_time(e) / n * 1e3
    return res_t


for name, K in (("o", 2048), ("wo", 5120)):
    a = run(2012, 2048, K, 4); b = run(2012, 2048, K, 2); c = run(4024, 2048, K // 2, 2)
    print(f"{name:3s} K={K}: 64x128 (today) warm {a['warm']:6.1f} cold {a['cold']:6.1f} | 128x128, 256 tiles warm {b['warm']:6.1f} cold {b['cold']:6.1f} | "
          f"128x128, 512 tiles of K/2 warm {c['warm']:6.1f} cold {c['cold']:6.1f} us", flush=True)
