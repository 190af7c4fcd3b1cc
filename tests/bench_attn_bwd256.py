"""Tuning aid: the softmax part of the head-dimension-256 attention backward -- one launch (csrc/attention_bwd256.hip) against
the three it replaces (row term, PROB GEMM, DSCORE GEMM) -- and the whole core backward either way, at the step's two shapes."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import functional as F, ops
dev = torch.device("cuda:0")


def timed(run, n=10):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(n): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for name, B, H, Sq, Sk in (("video self", 16, 4, 256, 256), ("A<-V cross", 16, 4, 800, 256)):
    dk, D = 256, 1024
    scale = 1 / math.sqrt(dk)
    Q = torch.randn(B * Sq, D, device=dev).to(torch.bfloat16)
    K = torch.randn(B * Sk, D, device=dev).to(torch.bfloat16)
    V = torch.randn(B * Sk, D, device=dev).to(torch.bfloat16)
    dO = torch.randn(B * Sq, D, device=dev).to(torch.bfloat16)
    mask = torch.ones(B, 1, Sk, dtype=torch.uint8, device=dev); mask[0, 0, 200:] = 0
    O = torch.empty(B * Sq, D, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev); rsum = torch.empty(B, H, Sq, device=dev)
    ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, Sk, 0, B, H, Sq, Sk, dk, scale, D, D, D, D)
    P = torch.empty(B, H, Sq, Sk, dtype=torch.bfloat16, device=dev); dS = torch.empty_like(P)
    t = timed(lambda: ops.attention_bwd_scores256(Q, D, K, D, V, D, dO, D, rmax, rsum, mask, Sk, P, dS, Sk, B, H, Sq, Sk, scale))
    flops = 2 * 2.0 * B * H * Sq * Sk * dk
    byts = 2.0 * (B * Sq * D * 2 + B * Sk * D * 2 * ((Sq + 127) // 128) + 2 * B * H * Sq * Sk)
    print(f"{name}: scores256 {t:7.1f} us  ({flops / t / 1e6:.1f} TFLOP/s, {byts / t / 1e3:.0f} GB/s incl. K/V re-reads)")
    dQ = torch.empty(B * Sq, D, dtype=torch.bfloat16, device=dev); dK = torch.empty(B * Sk, D, dtype=torch.bfloat16, device=dev)
    dV = torch.empty_like(dK)
    for fused in (True, False):
        F.FUSED_SCORES_BWD = fused
        t = timed(lambda: F._attn_core_bwd(dO, O, ("flash", rmax, rsum), Q, 0, D, K, 0, D, V, 0, D, dQ, 0, D, dK, 0, D, dV, 0, D, mask,
                                           Sk, 0, B, H, Sq, Sk, dk, 0.0))
        print(f"{name}: core backward, fused scores {fused}: {t:7.1f} us")
