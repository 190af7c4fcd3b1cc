import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
def t(M,N,K,bf16out=False, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev); Cb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    run = (lambda: ops.gemm(A, B, M, N, K, lda=K, ldb=K, C_bf16=Cb, ldcb=N)) if bf16out else (lambda: ops.gemm(A, B, M, N, K, lda=K, ldb=K, C_f32=C, ldc=N))
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(iters): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for K in (64, 1024, 4096):
    print(f"dbg={os.environ.get('BMHRL_GEMM_DBG','0')} M4096 N1024 K={K:5d}: f32out {t(4096,1024,K):7.1f} us   bf16out {t(4096,1024,K,True):7.1f} us")
