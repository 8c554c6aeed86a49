"""The adapted projections of 8 decoder rows: the fused one-launch kernel (csrc/decproj.hip) against the two launches it replaces
(LoRA row kernel or fused RMSNorm + row kernel, then the skinny / gated tile GEMM), stand-alone, HIP events."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
R = int(os.environ.get("R", "8"))
bf = lambda t: t.bfloat16()
seed = torch.tensor([3], dtype=torch.int32, device=dev)


def timeit(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def case(name, N, K, Rk, norm, gated, f32out, bwd=False):
    rows = 2 * N if gated else N
    w = bf(torch.randn(rows, K, device=dev) * 0.03); a = bf(torch.randn(Rk, K, device=dev) * 0.05)
    wext = torch.zeros(rows, 64, dtype=torch.bfloat16, device=dev); wext[:, :Rk] = bf(torch.randn(rows, Rk, device=dev) * 0.05)
    x32 = torch.randn(R, K, device=dev); gamma = torch.ones(K, device=dev)
    x = bf(torch.randn(R, K, device=dev)); xn = torch.zeros(R, K, dtype=torch.bfloat16, device=dev)
    u = torch.zeros(R, 64, dtype=torch.bfloat16, device=dev)
    out = torch.empty(R, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
    res = torch.randn(R, N, device=dev) if f32out else None
    h = torch.zeros(R, 2 * N, dtype=torch.bfloat16, device=dev) if gated else None
    ld, od = ops.Dropout(seed, 4, 0.05), (ops.Dropout(seed, 8, 0.1) if (f32out or gated) and not bwd else None)
    if bwd:
        two = lambda: (ops.lora_rows(x, a, u, K), ops.lora_dx(x, w, u, wext, out, K, residual=res, drop=ld))
        one = lambda: ops.dec_proj(x, w, a, wext, u, out, K, residual=res, ext_drop=ld)
    elif norm:
        two = lambda: (ops.rmsnorm_lora_fwd(x32, gamma, 1e-6, xn, a, u, drop=ld), ops.gemm(xn, w, out, aext=u, wext=wext, out2=h, gated=gated, drop=od, tile_cfg=2 if gated else 0))
        one = lambda: ops.dec_proj(xn, w, a, wext, u, out, K, x32=x32, gamma=gamma, eps=1e-6, out2=h, gated=gated, in_drop=ld, out_drop=od)
    else:
        two = lambda: (ops.lora_rows(x, a, u, K, drop=ld), ops.gemm(x, w, out, aext=u, wext=wext, residual=res, drop=od))
        one = lambda: ops.dec_proj(x, w, a, wext, u, out, K, residual=res, in_drop=ld, out_drop=od)
    t2, t1 = timeit(two), timeit(one)
    mb = rows * K * 2 / 1e6
    print(f"{name:28s} R={R} N={N:5d} K={K:5d} Rk={Rk:2d}: two launches {t2:6.1f} us, fused {t1:6.1f} us  ({mb:5.1f} MB of weights -> {mb / t1:5.2f} TB/s)")


case("qkv (norm)", 6144, 2048, 24, True, False, False)
case("o / co (residual)", 2048, 2048, 8, False, False, True)
case("cross q (norm)", 2048, 2048, 8, True, False, False)
case("wi (norm, gated)", 5120, 2048, 16, True, True, False)
case("wo (residual)", 2048, 5120, 8, False, False, True)
case("lm_head", 32128, 2048, 8, False, False, True)
case("bwd dX qkv", 2048, 6144, 24, False, False, True, bwd=True)
case("bwd dX wi", 2048, 10240, 16, False, False, True, bwd=True)
case("bwd dX wo", 5120, 2048, 8, False, False, False, bwd=True)
case("bwd dX o", 2048, 2048, 8, False, False, False, bwd=True)
