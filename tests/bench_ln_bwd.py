import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
for rows, D in ((4096, 1024), (12800, 128), (480, 300)):
    dy = torch.randn(rows, D, device=dev); x = torch.randn(rows, D, device=dev); g = torch.randn(D, device=dev)
    mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx = torch.empty_like(x); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    run = lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dy, dg, db, rows, D)
    run(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for _ in range(20): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"ln_bwd rows={rows} D={D}: {us:.1f} us  {rows*D*16/us/1e6:.2f} TB/s")
