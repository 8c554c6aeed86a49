"""The four GEMMs of a T5-XL encoder layer's FORWARD at the QVH shape (M = 2012; LoRA K extension, the layer's real epilogues) per tile
config, stand-alone — the encoder forward is the one phase of the step that runs with the chip to itself (the look-ahead ViT starts
behind it), so these times count 1:1 in the step.   python tools/enc_fwd_gemm_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M = int(os.environ.get("M", "2012"))
bf = lambda t: t.bfloat16()
seed = torch.tensor([3], dtype=torch.int32, device=dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


CF = [int(c) for c in os.environ.get("CFGS", "").split(",") if c]
COLD = int(os.environ.get("COLD", "0"))     # > 0: rotate through COLD weight sets (> the 256 MB Infinity Cache), as the step's 24 layers do


def case(name, N, K, cfgs, gated=False, f32=False, tout=False):
    rows = 2 * N if gated else N
    a = bf(torch.randn(M, K, device=dev)); w = bf(torch.randn(rows, K, device=dev) * 0.03)
    u = bf(torch.randn(M, 64, device=dev)); wext = bf(torch.randn(rows, 64, device=dev) * 0.05)
    if COLD:
        nset = max(COLD, -(-640 * 2**20 // (rows * K * 2)))      # more than the Infinity Cache holds
        ws = [w] + [w.clone() for _ in range(nset - 1)]; wes = [wext] + [wext.clone() for _ in range(nset - 1)]
        state = [0]
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if f32 else None
    h = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev) if gated else None
    drop = ops.Dropout(seed, 7, 0.1) if (f32 or gated) else None
    touts = [torch.zeros(1, 32, 64, ops.rup32(M), dtype=torch.bfloat16, device=dev) for _ in range(3)] if tout else None
    line = f"{name:24s} N={N:5d} K={K:5d}:"
    for c in cfgs:
        try:
            def run():
                if COLD:
                    state[0] = (state[0] + 1) % nset
                    ww, we = ws[state[0]], wes[state[0]]
                else:
                    ww, we = w, wext
                ops.gemm(a, ww, out, aext=u, wext=we, residual=res, out2=h, gated=gated, drop=drop, tile_cfg=c, tout=touts, t_rows=M)
            t = timeit(run, n=48 if COLD else 20)
            line += f"  cfg{c} {t:6.1f}us {2 * M * rows * K / t / 1e6:5.0f}TF"
        except Exception as e:
            line += f"  cfg{c} ERR({str(e)[:30]})"
    print(line, flush=True)


if not os.environ.get("ONLY_WO"):
  case("qkv (bf16 + q/k/v^T)", 6144, 2048, CF or [0, 8, 2, 4, 1], tout=True)
  case("qkv (bf16, no copies)", 6144, 2048, CF or [0, 8, 12, 2, 4, 1])
  case("o (fp32 residual)", 2048, 2048, CF or [0, 4, 5, 2], f32=True)
  case("wi (gated)", 5120, 2048, CF or [0, 2, 4, 8, 1, 9], gated=True)
case("wo (fp32 residual)", 2048, 5120, CF or [0, 4, 5, 2], f32=True)
