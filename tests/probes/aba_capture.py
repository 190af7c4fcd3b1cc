"""Probe behind tests/test_split_backward_gpu.py::test_a_captured_trainer_survives_a_larger_second_trainer with progress prints and
variants: argv[1] = output file (unused), argv[2] = alone | with_b | with_b_nostep | with_b_sidestep | with_b_noa (DESIGN.md section 10 "r04")."""
import faulthandler; faulthandler.enable()
import sys, torch
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
dev = torch.device("cuda:0")
def batch(B, Tv, Ta, L, seed):
    b = syn.synthetic_batch(B, Tv, Ta, L, 300, seed=seed)
    return {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}, b["captions"].to(dev)
fa, ca = batch(2, 64, 100, 10, 2)
if "noa" in sys.argv[2]:
    A = None; losses = []
else:
    A = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3)
    A.agent.train()
    print("A built", flush=True)
    A.capture(fa, ca, warmup=1)
    print("A captured", flush=True)
    losses = [float(A.replay()) for _ in range(2)]
if sys.argv[2].startswith("with_b"):
    # a second, LARGER trainer: its own scratch arena / operand pools / graph; eager steps and a capture of its own
    fb, cb = batch(4, 128, 200, 14, 3)
    Bt = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3, seed=5)
    Bt.agent.train()
    print("B built", flush=True)
    if "nostep" not in sys.argv[2]:
        if "sidestep" in sys.argv[2]:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                Bt.step(fb, cb)
            torch.cuda.current_stream().wait_stream(s_)
        else:
            Bt.step(fb, cb)
    torch.cuda.synchronize(); print("B stepped", flush=True)
    Bt.capture(fb, cb, warmup=1)
    print("B captured", flush=True)
    lb = [float(Bt.replay()) for _ in range(2)]
    print("B replayed", flush=True)
    assert all(x == x for x in lb)
if A is not None:
    print("A replay again", flush=True)
    losses += [float(A.replay()) for _ in range(2)]
print("done", losses, flush=True)
torch.cuda.synchronize()
