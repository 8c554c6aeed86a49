"""Time + cross-check tile configs on given shapes:  python tools/gemm_cmp.py "M,N,K[,mode]" ... -- cfg cfg ...   (mode: plain | gelu | res)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
args = sys.argv[1:]
i = args.index("--")
shapes, cfgs = args[:i], [int(c) for c in args[i + 1:]]
dev = torch.device("cuda:0")
for sh in shapes:
    f = sh.split(",")
    M, N, K = int(f[0]), int(f[1]), int(f[2])
    mode = f[3] if len(f) > 3 else "plain"
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias)
    if mode == "gelu":
        kw["act"] = 1
    if mode == "res":
        res = torch.randn(M, N, device=dev)
        kw["residual"] = res
        out = torch.empty(M, N, device=dev)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ref = None
    line = f"{M}x{N}x{K} {mode}:"
    for c in cfgs:
        out.zero_()
        for _ in range(3):
            ops.gemm(a, w, out, tile_cfg=c, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, out, tile_cfg=c, **kw)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        if ref is None:
            ref = out.float().clone()
        err = (out.float() - ref).abs().max().item()
        line += f"  cfg{c} {t:7.1f}us {2*M*N*K/t/1e6:5.0f}TF" + ("" if err == 0 else f" (maxdiff {err:.2e})")
    if mode == "plain":
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            torch.matmul(a, w.t(), out=out)
        t0.record()
        for _ in range(20):
            torch.matmul(a, w.t(), out=out)
        t1.record(); torch.cuda.synchronize()
        t = t0.elapsed_time(t1) / 20 * 1e3
        line += f"  blaslt {t:7.1f}us {2*M*N*K/t/1e6:5.0f}TF"
    print(line)
