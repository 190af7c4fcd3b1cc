"""Tuning aid: which torch (aten) operators -- not bmhrl kernels -- a training step launches, and from where."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import _lib, synthetic as syn
from bmhrl_amd.train import CaptionTrainer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg = syn.default_cfg(dout_p=0.1, rl_att_layers=2)
tr = CaptionTrainer(cfg, 10172, dev, lr=1e-4)
tr.agent.train(); tr.agent.set_inference_mode(True)
b = syn.synthetic_batch(16, 256, 800, 30, 10172, seed=0)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}; cap = b["captions"].to(dev)
for _ in range(3): tr.step(fs, cap)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.step(fs, cap)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=12)
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    st = [s for s in e.stack if "bmhrl_amd" in s or "train.py" in s][:3]
    print(f"{e.key:22s} n={e.count:3d} dev={e.device_time_total:7.1f}us  " + " | ".join(s.split('/')[-1][:55] for s in st))

print("\n-- copies by call site (count, whether or not device time was attributed to the operator)")
cp = [e for e in ka if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy")]
cp.sort(key=lambda e: -e.count)
for e in cp[:30]:
    st = [s for s in e.stack if "bmhrl_amd" in s or "train.py" in s or "autograd" in s][:3]
    print(f"{e.key:18s} n={e.count:3d}  " + " | ".join(s.split('/')[-1][:60] for s in st))
