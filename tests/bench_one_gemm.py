import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4096, 1024, 4096
A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
Cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(10):
    ops.gemm(A, B, M, N, K, lda=K, ldb=K, C_bf16=Cb, ldcb=N)
torch.cuda.synchronize()
