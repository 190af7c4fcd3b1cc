"""Tuning aid: LayerNorm forward at the step's shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
for rows, D in ((4096, 1024), (12800, 128), (480, 300)):
    x = torch.randn(rows, D, device=dev); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    yb = torch.zeros(rows, ops.pad8(D), dtype=torch.bfloat16, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    run = lambda: ops.layernorm_fwd(x, g, b, yb, yb.shape[1], None, mean, rstd, rows, D)
    run(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for _ in range(20): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"ln_fwd rows={rows} D={D}: {us:.1f} us  {rows*D*6/us/1e6:.2f} TB/s")
