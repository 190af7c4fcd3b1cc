import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
B, H, Sq, Sk, dk = 16, 4, int(os.environ.get("SQ", 256)), int(os.environ.get("SK", 800)), 256
D = H * dk
Q = torch.randn(B, Sq, D, device=dev).to(torch.bfloat16); K = torch.randn(B, Sk, D, device=dev).to(torch.bfloat16)
V = torch.randn(B, Sk, D, device=dev).to(torch.bfloat16)
mask = torch.ones(B, 1, Sk, dtype=torch.bool, device=dev)
O = torch.empty(B, Sq, D, dtype=torch.bfloat16, device=dev)
rmax = torch.empty(B, H, Sq, device=dev); rsum = torch.empty(B, H, Sq, device=dev)
run = lambda: ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, Sk, 0, B, H, Sq, Sk, dk, dk ** -0.5, D, D, D, D)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g):
        for _ in range(10): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"attn fwd Sq={Sq} Sk={Sk}: {us:.1f} us  {4*B*Sq*Sk*D/us/1e6:.1f} TF")
