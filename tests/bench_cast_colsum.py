"""Tuning aid: cast + column-sum kernel at the step's shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
for rows, cols in ((4096, 1024), (12800, 128), (480, 300), (4096, 4096)):
    x = torch.randn(rows, cols, device=dev); y = torch.zeros(rows, ops.pad8(cols), dtype=torch.bfloat16, device=dev)
    cs = torch.zeros(cols, device=dev)
    run = lambda: ops.cast_colsum_bf16(x, cols, y, y.shape[1], rows, cols, cs)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(20): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"cast_colsum rows={rows} cols={cols}: {us:.1f} us  {rows*cols*6/us/1e6:.2f} TB/s")
