import torch
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
from bmhrl_amd.functional import SCRATCH
dev = torch.device("cuda:0")
t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.1), 80, dev, lr=1e-3)
t.agent.train()
b = syn.synthetic_batch(4, 64, 200, 12, 80, seed=3)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
for i in range(3):
    h0 = SCRATCH.memo_hits
    t.step(fs, cap)
    print("step", i, "memo hits", SCRATCH.memo_hits - h0)
