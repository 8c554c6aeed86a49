import torch
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev).bfloat16(); w = torch.randn(8192, 8192, device=dev).bfloat16(); out = torch.empty(8192, 8192, dtype=torch.bfloat16, device=dev)
for _ in range(7000): torch.matmul(a, w.t(), out=out)
torch.cuda.synchronize()
