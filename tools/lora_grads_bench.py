"""lora_grads (both LoRA weight gradients of a fused group in one launch) at the T5-XL encoder shapes, M = 2012 rows:
time vs the bytes it must read (dy [M,N] + x [M,K] bf16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mr-blip_amd"))
from mrblip import ops
dev = torch.device("cuda:0")
M = 2012
seed = torch.tensor([3], dtype=torch.int32, device=dev)
for name, outs, K in [("qkv", [2048, 2048, 2048], 2048), ("o", [2048], 2048), ("wi", [5120, 5120], 2048), ("wo", [2048], 5120)]:
    n, N = len(outs), sum(outs)
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    u = torch.randn(M, 64, device=dev).bfloat16(); g = torch.randn(M, 64, device=dev).bfloat16()
    dB = [torch.zeros(8, o, device=dev) for o in outs]; dA = [torch.zeros(8, K, device=dev) for _ in outs]
    col0 = [sum(outs[:i]) for i in range(n)]
    drop = ops.Dropout(seed, 9, 0.05) if os.environ.get("NO_DROP") is None else None
    f = lambda: ops.lora_grads(dy, u, x, g, dB, col0, outs, dA, K, drop=drop)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 30 * 1e3
    mb = M * (N + K) * 2 / 1e6
    print(f"{name}: {t:.1f} us for {mb:.1f} MB -> {mb / t:.2f} TB/s")

# the encoder layer's four groups in ONE launch (lora_grads_batched: what the step's side stream runs per layer, 128 MB of operands)
jobs, keep = [], []
for name, outs, K in [("qkv", [2048, 2048, 2048], 2048), ("o", [2048], 2048), ("wi", [5120, 5120], 2048), ("wo", [2048], 5120)]:
    n, N = len(outs), sum(outs)
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    u = torch.randn(M, 64, device=dev).bfloat16(); g = torch.randn(M, 64, device=dev).bfloat16()
    dB = [torch.zeros(8, o, device=dev) for o in outs]; dA = [torch.zeros(8, K, device=dev) for _ in outs]
    col0 = [sum(outs[:i]) for i in range(n)]
    drop = ops.Dropout(seed, 9, 0.05) if os.environ.get("NO_DROP") is None else None
    keep.append((dy, x, u, g, dB, dA))
    jobs.append(ops.lora_grads_job(dy, u, x, g, dB, col0, outs, dA, K, drop=drop))
for _ in range(3): ops.lora_grads_batched(jobs)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): ops.lora_grads_batched(jobs)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 30 * 1e3
mb = sum(k[0].numel() + k[1].numel() for k in keep) * 2 / 1e6
print(f"layer (4 groups, one launch): {t:.1f} us for {mb:.1f} MB -> {mb / t:.2f} TB/s")
