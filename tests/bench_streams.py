"""Do two independent GEMM chains captured as parallel graph branches run faster than back to back?  (tuning aid)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
def mk(M, N, K):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    return lambda: ops.gemm(A, B, M, N, K, lda=K, ldb=K, C_bf16=C, ldcb=N)
for shape in ((4096, 1024, 1024), (480, 304, 1024), (12800, 2048, 128), (4096, 3072, 1024)):
    ga = [mk(*shape) for _ in range(4)]; gb = [mk(*shape) for _ in range(4)]
    for f in ga + gb: f()
    torch.cuda.synchronize()
    def timed(body):
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g): body()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    def serial():
        for f in ga + gb: f()
    side = torch.cuda.Stream()
    def parallel():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for f in gb: f()
        for f in ga: f()
        main.wait_stream(side)
    print(shape, f"serial {timed(serial):.1f} us   two branches {timed(parallel):.1f} us")
