"""Tuning aid: launch time of the head-dim-128 attention kernel over batch (workgroups per CU) and key count (tiles)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
H, Sq = 4, 256


def t(B, Sk, Sq=Sq):
    Qp = torch.randn(B, Sq, H, 128, device=dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, 128, device=dev).to(torch.bfloat16)
    mask = torch.ones(B, Sk, dtype=torch.bool, device=dev)
    ctx = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev); rsum = torch.empty(B, H, Sq, device=dev)
    run = lambda: ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, mask, Sk, B, H, Sq, Sk, 1 / 16, H * 128, 128, H * 128)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    return best


import os
for B in [int(x) for x in os.environ.get('SCAN_B', '8,16,32,64').split(',')]:
    print(f"B={B:3d} (blocks/CU {B * H * Sq / 64 / 256:.1f}): " + "  ".join(f"Sk={Sk}: {t(B, Sk):6.1f} us" for Sk in (64, 192, 448, 800, 1024)), flush=True)
