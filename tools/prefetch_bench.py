"""Does a weight prefetch into the memory-side cache give the T5 GEMMs their stand-alone (re-used weights) speed back?  The four GEMMs of an
encoder layer's forward at M = 2012 over a rotation of weight sets larger than the Infinity Cache: (a) as in the step, (b) with the NEXT
set read by a few blocks on a side stream while the current GEMM runs, (c) warm (one set).   python tools/prefetch_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
bf = lambda t: t.bfloat16()
seed = torch.tensor([3], dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
NB = [int(x) for x in os.environ.get("NB", "16,32,64").split(",")]


def case(name, N, K, gated=False, f32=False):
    rows = 2 * N if gated else N
    nset = -(-640 * 2**20 // (rows * K * 2))
    a = bf(torch.randn(M, K, device=dev)); w0 = bf(torch.randn(rows, K, device=dev) * 0.03)
    u = bf(torch.randn(M, 64, device=dev)); wext = bf(torch.randn(rows, 64, device=dev) * 0.05)
    ws = [w0] + [w0.clone() for _ in range(nset - 1)]
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if f32 else None
    h = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev) if gated else None
    drop = ops.Dropout(seed, 7, 0.1) if (f32 or gated) else None

    def g(w): ops.gemm(a, w, out, aext=u, wext=wext, residual=res, out2=h, gated=gated, drop=drop)

    def loop(mode, nb=0, n=3 * 24):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            if rep == 1: s.record()
            for i in range(n):
                w = ws[0] if mode == "warm" else ws[i % nset]
                if mode == "grid": ops.gemm_prefetch(ws[(i + 1) % nset], nb)
                if mode == "side":
                    ev = torch.cuda.Event(); ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        ops.prefetch(ws[(i + 1) % nset], nb)
                g(w)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    def percall(pre, n=48):
        tot = 0.0
        for rep in range(2):
            evs = []
            for i in range(n):
                w = ws[i % nset]
                if pre == "same": ops.prefetch(w, 256)
                elif pre == "sum": w.view(torch.int32).sum()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); g(w); e.record(); evs.append((s, e))
            torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in evs) / n * 1e3
    print(f"{name:20s} per-call events: cold {percall(None):6.1f}  after a serial prefetch of the same set {percall('same'):6.1f}  after torch sum {percall('sum'):6.1f} us", flush=True)
    line = f"{name:20s} N={N:5d} K={K:5d} ({rows * K * 2 / 2**20:5.1f} MB x {nset}):  warm {loop('warm'):6.1f}  cold {loop('cold'):6.1f}"
    for nb in NB: line += f"  side/{nb} {loop('side', nb):6.1f}  in-grid/{nb} {loop('grid', nb):6.1f}"
    print(line + " us", flush=True)


def qcase(name, N, K, f32=False, act=0):
    """Q-Former shapes (M = 60 x 32 query rows, d = 768): bias epilogues, no LoRA"""
    Mq = 1920
    nset = -(-640 * 2**20 // (N * K * 2))
    a = bf(torch.randn(Mq, K, device=dev)); w0 = bf(torch.randn(N, K, device=dev) * 0.03)
    ws = [w0] + [w0.clone() for _ in range(nset - 1)]
    bias = torch.randn(N, device=dev)
    out = torch.empty(Mq, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    res = torch.randn(Mq, N, device=dev) if f32 else None
    pre = torch.empty(Mq, N, dtype=torch.bfloat16, device=dev) if act else None
    drop = ops.Dropout(seed, 7, 0.1) if f32 else None

    def loop(mode, nb=0, n=240):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            if rep == 1: s.record()
            for i in range(n):
                w = ws[0] if mode == "warm" else ws[i % nset]
                if mode == "grid": ops.gemm_prefetch(ws[(i + 1) % nset], nb)
                ops.gemm(a, w, out, bias=bias, residual=res, drop=drop, act=act, out2=pre)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    line = f"{name:20s} N={N:5d} K={K:5d} ({N * K * 2 / 2**20:5.1f} MB x {nset}):  warm {loop('warm'):6.1f}  cold {loop('cold'):6.1f}"
    for nb in (8, 16, 32): line += f"  in-grid/{nb} {loop('grid', nb):6.1f}"
    print(line + " us", flush=True)


if os.environ.get("QF"):
    qcase("qf qkv", 2304, 768)
    qcase("qf o (res)", 768, 768, f32=True)
    qcase("qf fc1 (gelu)", 3072, 768, act=1)
    qcase("qf fc2 (res)", 768, 3072, f32=True)
    sys.exit(0)
ONLY = os.environ.get("ONLY", "")
if not ONLY or ONLY == "qkv": case("qkv", 6144, 2048)
if not ONLY or ONLY == "o": case("o (fp32 residual)", 2048, 2048, f32=True)
if not ONLY or ONLY == "wi": case("wi (gated)", 5120, 2048, gated=True)
if not ONLY or ONLY == "wo": case("wo (fp32 residual)", 2048, 5120, f32=True)
