"""Tuning aid: time of the captured step with pieces left out (forward + loss only; + backward; + gather; + Adam)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
from bmhrl_amd.functional import SCRATCH, SEEDS, SHADOWS
dev = torch.device("cuda:0")
cfg = syn.default_cfg(dout_p=0.1, rl_att_layers=2)
tr = CaptionTrainer(cfg, 10172, dev, lr=1e-4)
tr.agent.train(); tr.agent.set_inference_mode(True)
b = syn.synthetic_batch(16, 256, 800, 30, 10172, seed=0)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}; cap = b["captions"].to(dev)
trg_in, trg_y = cap[:, :-1].contiguous(), cap[:, 1:].contiguous()


def body(level):
    tr.opt.zero_grad(); SCRATCH.begin_step(dev); SEEDS.dev.add_(1); SHADOWS.invalidate(); SHADOWS.refresh()
    if level == 0:
        with torch.no_grad():
            loss, _ = tr._forward_loss(fs, trg_in, trg_y)
    else:
        loss, _ = tr._forward_loss(fs, trg_in, trg_y)
        loss.backward()
        if level >= 2: tr.opt.gather_grads()
    SCRATCH.end_step()
    if level >= 3: tr.opt.step(1.0)


for level, name in ((0, "forward + loss (no grad)"), (1, "+ backward"), (2, "+ gradient gather"), (3, "+ Adam (whole step)")):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): body(level)
        SHADOWS.refresh()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(level)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 30:.3f} ms")
