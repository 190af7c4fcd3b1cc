"""Tuning aid: time per launch of a trivial kernel inside a 20-node HIP graph (the floor under the small-kernel timings)."""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
x.add_(1); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g):
        for _ in range(20): x.add_(1)
g.replay(); torch.cuda.synchronize()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"trivial kernel in a graph: {e0.elapsed_time(e1) * 50:.2f} us per launch")
