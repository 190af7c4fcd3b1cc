"""Tuning aid: how many bf16 operand casts of a training step are served from the step's memo (a producer's offer --
StepScratch.offer_bf16 -- or an earlier cast of the same tensor) instead of a cast launch."""
import torch
from bmhrl_amd import synthetic as syn
from bmhrl_amd.functional import SCRATCH
from bmhrl_amd.train import CaptionTrainer

dev = torch.device("cuda:0")
t = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, lr=1e-3)
t.agent.train()
b = syn.synthetic_batch(2, 128, 200, 12, 300, seed=2)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
for i in range(3):
    h0 = SCRATCH.memo_hits
    t.step(fs, cap)
    print("step", i, "memo hits", SCRATCH.memo_hits - h0)
