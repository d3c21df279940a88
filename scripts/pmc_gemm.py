"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4608, 11008, 4096
x = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, N, device=dev).bfloat16()
dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
dW = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_raw(x, W, y, M, N, K, K, K, N)
    ops.linear_dx(dy, W, out=dx)
    ops.linear_dw(dy, x, out=dW)
torch.cuda.synchronize()
