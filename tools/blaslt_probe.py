"""Which hipBLASLt kernels torch.matmul picks at the ViT shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(15420, 6144, 1408), (15420, 1408, 6144), (15420, 4224, 1408), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(10):
        torch.matmul(a, w.t(), out=out)
torch.cuda.synchronize()
