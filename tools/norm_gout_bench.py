"""RMSNorm backward with four K-split parts + the LoRA part and the fused g product (norm_bwd<RMS, GOUT>) at the T5-XL encoder shape:
   MRBLIP_LIB=exp_libs/libold_norm.so python tools/norm_gout_bench.py   (A/B of library builds)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M, D = 2012, 2048
torch.manual_seed(1)
parts = torch.randn(5, M, D, device=dev)
x = torch.randn(M, D, device=dev); w = torch.rand(D, device=dev) + 0.5; add = torch.randn(M, D, device=dev)
seed = torch.tensor([5], dtype=torch.int32, device=dev)
drop, edrop = ops.Dropout(seed, 3, 0.1), ops.Dropout(seed, 4, 0.05)
gb = (torch.randn(8, D, device=dev) * 0.1).bfloat16(); gout = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev)
dx = torch.empty(M, D, device=dev); ob = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
for name, kw in (("parts + g product", dict(g_prod=(gb, gout))), ("parts only", dict())):
    f = lambda: ops.rmsnorm_bwd(parts, x, w, 1e-6, dx, dx_add=add, out_bf16=ob, out_drop=drop, ext_drop=edrop, ext_part=True, **kw)
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): f()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 50 * 1e3:.1f} us   (checksum {float(dx.double().sum()):.6e} {float(gout.float().sum()):.6e})")
