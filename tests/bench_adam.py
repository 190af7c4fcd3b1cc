"""Tuning aid: the optimizer pass of the trainer alone (Adam over the 221 MB bucket + shadow writes), in a HIP graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
dev = torch.device("cuda:0")
tr = CaptionTrainer(syn.default_cfg(dout_p=0.1), 10172, dev, lr=1e-4)
tr.agent.train(); tr.agent.set_inference_mode(True)
b = syn.synthetic_batch(16, 256, 800, 30, 10172, seed=0)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}; cap = b["captions"].to(dev)
for _ in range(2): tr.step(fs, cap)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g):
        for _ in range(5): tr.opt.step(1.0)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
n = tr.opt.n
print(f"adam pass: {us:.1f} us for {n / 1e6:.1f} M parameters = {n * 30 / us / 1e6:.2f} TB/s (16 B read + 12 B written + 2 B shadow per parameter)")
