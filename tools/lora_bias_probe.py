"""Is there a systematic shrink in the LoRA weight-gradient pair of ONE adapter?  dA = g^T x with g = bf16(dy (sB)), dB^T = u^T dy with
u = bf16(x (sA)^T): the launches the engine uses (lora_rows for g / u, lora_grads for the pair), against fp64 torch on the SAME bf16
operands (kernel error only) and on the un-rounded fp32 operands (operand rounding included); signed norm ratios.  Shapes: lm_head
(12 rows, N = 32128), decoder wo (12 rows), encoder wo (2012 rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
import torch
from mrblip import ops
dev = torch.device("cuda:0")
bf = lambda t: t.bfloat16()
torch.manual_seed(0)
for name, M, N, K, dy_kind in (("lm_head", 12, 32128, 2048, "ce"), ("dec wo", 12, 2048, 5120, "randn"), ("enc wo", 2012, 2048, 5120, "randn"), ("enc wi", 2012, 10240, 2048, "randn")):
    x32 = torch.randn(M, K, device=dev)
    A32 = torch.randn(8, K, device=dev) * 0.02
    B32 = torch.randn(N, 8, device=dev) * 0.02
    if dy_kind == "ce":
        logits = torch.randn(M, N, device=dev) * 0.5
        p = torch.softmax(logits, -1)
        p[torch.arange(M), torch.randint(0, N, (M,), device=dev)] -= 1.0
        dy32 = p / M
    else:
        dy32 = torch.randn(M, N, device=dev) * 1e-3
    x, dy = bf(x32), bf(dy32)
    acat = torch.zeros(8, K, dtype=torch.bfloat16, device=dev); acat.copy_(bf(A32))
    bblk = torch.zeros(8, N, dtype=torch.bfloat16, device=dev); bblk.copy_(bf(B32.t()))
    u = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev)
    g = torch.zeros(M, 64, dtype=torch.bfloat16, device=dev)
    if M <= 511:
        ops.lora_rows(x, acat, u, K)
        ops.lora_rows(dy, bblk, g, N)
    else:
        ops.gemm(x, acat, u, tile_cfg=3, K=K)
        ops.gemm(dy, bblk, g, tile_cfg=3, K=N)
    dBt = torch.zeros(8, N, device=dev)
    dA = torch.zeros(8, K, device=dev)
    ops.lora_grads(dy, u, x, g, [dBt], [0], [N], [dA], K)
    torch.cuda.synchronize()
    # same bf16 operands, fp64 arithmetic, the intermediate u / g NOT rounded
    xd, dyd, Ad, Bd = x.double(), dy.double(), acat.double(), bblk.double()
    dA_k, dB_k = (dyd @ Bd.t()).t() @ xd, (xd @ Ad.t()).t() @ dyd
    # fp32 operands (what an fp32 run computes)
    dA_f, dB_f = (dy32.double() @ B32.double()).t() @ x32.double(), (x32.double() @ A32.double().t()).t() @ dy32.double()
    r = lambda a, b: float(a.double().norm() / b.norm()) - 1
    e = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(f"{name:8s} M={M:5d} N={N:6d} K={K:5d}: dA ratio-1 vs same-operand fp64 {r(dA, dA_k):+.2e} (err {e(dA, dA_k):.1e}), vs fp32-operand {r(dA, dA_f):+.2e} (err {e(dA, dA_f):.1e});"
          f"  dB ratio-1 {r(dBt, dB_k):+.2e} (err {e(dBt, dB_k):.1e}), vs fp32-operand {r(dBt, dB_f):+.2e} (err {e(dBt, dB_f):.1e});"
          f"  g ratio-1 {r(g[:, :8], dyd @ Bd.t()):+.2e}, u ratio-1 {r(u[:, :8], xd @ Ad.t()):+.2e}", flush=True)
