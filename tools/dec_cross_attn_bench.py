"""The decoder's cross attention stand-alone: 8 label rows against the 2012 encoder keys, 32 heads (forward, dQ + dK/dV)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
B, H, D, Sq = 1, 32, 64, int(os.environ.get("SQ", "8"))
for Sk in (2012, 692, 72):
    q = torch.randn(B, Sq, H, D, device=dev).bfloat16() * 0.3; k = torch.randn(B, Sk, H, D, device=dev).bfloat16() * 0.3
    v = torch.randn(B, Sk, H, D, device=dev).bfloat16(); do = torch.randn(B, Sq, H, D, device=dev).bfloat16()
    seed = torch.tensor([7], dtype=torch.int32, device=dev); drop = ops.Dropout(seed, 5, 0.1)
    vt, kt, qt, dot = ops.head_transpose(v), ops.head_transpose(k), ops.head_transpose(q), ops.head_transpose(do)
    o = torch.empty_like(q); lse = torch.zeros(B, H, ops.rup32(Sq), device=dev); delta = torch.zeros_like(lse)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    def t(fn, n=50):
        for _ in range(5): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    tf = t(lambda: ops.attention_fwd(q, k, vt, o, lse, scale=1.0, drop=drop))
    tb = t(lambda: ops.attention_bwd(q, k, v, o, do, kt, qt, dot, lse, delta, dq, dk, dv, scale=1.0, drop=drop))
    print(f"Sq={Sq} Sk={Sk}: forward {tf:.1f} us, backward (dQ + dK/dV) {tb:.1f} us")
