"""GEMM micro-benchmark over the hot path's shapes (tuning aid, not a test).  Launches are captured into a HIP graph so
the timing is GPU-bound (an eager ctypes launch costs ~17 us of host time).  usage: python tests/bench_gemm.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
shapes = [  # (M, N, K, a_trans, b_trans, out, tag)
    (4096, 3072, 1024, 0, 0, "bf16", "V qkv fwd"), (4096, 1024, 1024, 0, 0, "f32", "V out/ffn fwd"), (12800, 3072, 128, 0, 0, "bf16", "A qkv fwd"),
    (12800, 128, 1024, 0, 0, "f32", "A out fwd"), (12800, 2048, 128, 0, 0, "bf16", "KV(A) fwd"), (4096, 2048, 1024, 0, 0, "bf16", "KV(V) fwd"),
    (480, 10172, 364, 0, 0, "f32", "vocab fwd"), (4096, 1024, 3072, 0, 1, "f32", "V qkv dx"), (4096, 1024, 1024, 0, 1, "bf16", "V out dx"),
    (12800, 128, 3072, 0, 1, "f32", "A qkv dx"), (1024, 1024, 4096, 1, 1, "f32", "V dW"), (3072, 1024, 4096, 1, 1, "f32", "V qkv dW"),
    (3072, 128, 12800, 1, 1, "f32", "A qkv dW"), (128, 1024, 12800, 1, 1, "f32", "A out dW"), (10172, 364, 480, 1, 1, "f32", "vocab dW"),
    (800, 800, 256, 0, 0, "bf16x64", "attn S (A self)"), (800, 256, 800, 1, 1, "bf16x64", "attn dV (A self)"),
    # caption-side GEMMs (B*L = 480 rows, d_model_caps 300): few blocks, latency bound
    (480, 1024, 300, 0, 0, "bf16", "C q proj fwd"), (480, 300, 1024, 0, 0, "f32", "C out proj fwd"), (480, 1024, 1024, 0, 0, "bf16", "C 1024 fwd"),
    (480, 300, 1024, 0, 1, "f32", "C q proj dx"), (480, 1024, 300, 0, 1, "bf16", "C out proj dx"), (4096, 2048, 300, 0, 0, "bf16", "KV(V) 300 fwd"),
]
N_IT = 20
for M, N, K, at, bt, out, tag in shapes:
    nb = 64 if out.endswith("x64") else 1
    A = torch.randn((nb, K, M) if at else (nb, M, K), device=dev).to(torch.bfloat16)
    A = torch.nn.functional.pad(A, (0, (-A.shape[-1]) % 8)).contiguous()
    B = torch.randn((nb, K, N) if bt else (nb, N, K), device=dev).to(torch.bfloat16)
    B = torch.nn.functional.pad(B, (0, (-B.shape[-1]) % 8)).contiguous()
    C = torch.zeros(nb, M, N, device=dev) if out == "f32" else None
    Cb = torch.zeros(nb, M, N, device=dev, dtype=torch.bfloat16) if out != "f32" else None
    def run():
        ops.gemm(A, B, M, N, K, lda=A.shape[-1], ldb=B.shape[-1], a_trans=bool(at), b_trans=bool(bt), C_f32=C, ldc=N, C_bf16=Cb, ldcb=N,
                 batch=(nb, 1), a_strides=(A.shape[1] * A.shape[2], 0), b_strides=(B.shape[1] * B.shape[2], 0),
                 c_strides=(M * N, 0), cb_strides=(M * N, 0), allow_split_k=bool(at and bt and out == "f32"))
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(N_IT): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N_IT
    print(f"{tag:18s} M={M:6d} N={N:6d} K={K:6d} at={at} bt={bt} {out:8s} {us:8.1f} us  {2*M*N*K*nb/us/1e6:7.1f} TF")
