import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
B, H = 16, 4
SHAPES = {"va": ((256, 800),), "aa": ((800, 800),)}.get(os.environ.get("ATTN_SHAPE", ""), ((256, 800), (800, 800)))
for Sq, Sk in SHAPES:
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
            for _ in range(10): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"shared-128 attn Sq={Sq} Sk={Sk}: {us:.1f} us  executed {4*B*H*Sq*Sk*128/us/1e6:.1f} TF  (d_k=256 form of the same attention: {4*B*H*Sq*Sk*256/1e9:.1f} GF)")
