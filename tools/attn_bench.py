"""Micro-benchmark of the attention kernels at the hot-path shapes."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

dev = torch.device("cuda:0")
seed = torch.tensor([1], dtype=torch.int32, device=dev)
for name, B, H, Sq, Sk, D, lut_on, drop_on, causal in [
    ("t5enc", 1, 32, 2012, 2012, 64, True, True, False), ("t5enc_masked", 1, 32, 2012, 2012, 64, True, True, False),
    ("t5enc_nodrop", 1, 32, 2012, 2012, 64, True, False, False),
    ("t5enc_plain", 1, 32, 2012, 2012, 64, False, False, False), ("vit", 60, 16, 257, 257, 88, False, False, False),
    ("qf_cross", 60, 12, 32, 257, 64, False, True, False), ("dec_cross", 1, 32, 8, 2012, 64, False, True, False)]:
    if os.environ.get("ATTN_ONLY") and name not in os.environ["ATTN_ONLY"].split(","):
        continue
    kmask = None
    if name in ("t5enc_masked", "dec_cross"):
        kmask = torch.zeros(B, ops.rup32(Sk), dtype=torch.int32, device=dev); kmask[:, :Sk] = 1
    q = torch.randn(B, Sq, H, D, device=dev).bfloat16(); k = torch.randn(B, Sk, H, D, device=dev).bfloat16(); v = torch.randn(B, Sk, H, D, device=dev).bfloat16()
    do = torch.randn(B, Sq, H, D, device=dev).bfloat16()
    o = torch.empty_like(q); lse = torch.zeros(B, H, ops.rup32(Sq), device=dev); delta = torch.zeros_like(lse)
    lut = torch.randn(H, 257, device=dev) if lut_on else None
    drop = ops.Dropout(seed, 3, 0.1) if drop_on else None
    vt = ops.head_transpose(v)
    scale = 1.0 if D == 64 else D ** -0.5
    dbits = torch.empty(ops.drop_bits_shape(B, H, Sq, Sk), dtype=torch.int32, device=dev) if (drop_on and os.environ.get("NO_DBITS") is None) else None
    t = timeit(lambda: ops.attention_fwd(q, k, vt, o, lse, scale=scale, bias_lut=lut, kmask=kmask, causal=causal, drop=drop, drop_bits=dbits))
    fl = 4.0 * B * H * Sq * Sk * D
    row = dict(name=name, fwd_us=round(t * 1e6, 1), fwd_TF=round(fl / t / 1e12, 1))
    if D > 64:
        row["fwd_rowv_us"] = round(timeit(lambda: ops.attention_fwd_rowv(q, k, v, o, lse, scale=scale)) * 1e6, 1)
    if D <= 64:
        kt, qt, dot = ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        t = timeit(lambda: ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=scale, bias_lut=lut, kmask=kmask, causal=causal, drop=drop, drop_bits=dbits))
        row.update(bwd_us=round(t * 1e6, 1), bwd_TF=round(2.5 * fl / t / 1e12, 1))
    print(json.dumps(row), flush=True)
