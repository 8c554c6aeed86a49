"""ViT self-attention forward (60 frames x 16 heads, head_dim 88) at 257 / 256 tokens: time and check against torch SDPA (fp32)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
B, H, D = 60, 16, 88
for S in (257, 256, 300):
    torch.manual_seed(0)
    qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    vt = ops.head_transpose(v)
    o = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=dev)
    for _ in range(3): ops.attention_fwd(q, k, vt, o, None, scale=D ** -0.5)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.attention_fwd(q, k, vt, o, None, scale=D ** -0.5)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e3
    o2 = torch.full_like(o, float("nan"))
    for _ in range(3): ops.attention_fwd_rowv(q, k, v, o2, None, scale=D ** -0.5)
    s.record()
    for _ in range(20): ops.attention_fwd_rowv(q, k, v, o2, None, scale=D ** -0.5)
    e.record(); torch.cuda.synchronize()
    t2 = s.elapsed_time(e) / 20 * 1e3
    s.record()
    for _ in range(20): ops.head_transpose(v, out=vt)
    e.record(); torch.cuda.synchronize()
    t3 = s.elapsed_time(e) / 20 * 1e3
    print(f"S={S}: row-major V {t2:.1f} us (equal to the V^T path: {torch.equal(o, o2)}); head_transpose(V) alone {t3:.1f} us")
    ref = torch.nn.functional.scaled_dot_product_attention(q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3), v.float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    err = ((o.float() - ref).norm() / ref.norm()).item()
    print(f"S={S}: {t:.1f} us  {4.0*B*H*S*S*D/t/1e6:.0f} TF  rel err {err:.2e}")
