"""Probe (round 4): an eager forward + backward on the LEGACY DEFAULT stream whose autograd graph is still referenced when the
same trainer's step is captured makes hipStreamEndCapture fault (SIGSEGV inside libamdhip64, frames +0x2d345d / +0x2d34a8 --
the function of the r03 nested-fork recursion, here without the recursion).  Found by the A -> B -> A test
(tests/test_split_backward_gpu.py::test_a_captured_trainer_survives_a_larger_second_trainer).  Variants (argv[1]), MI355X,
torch 2.10.0+rocm7.0, before the trainer kept its eager steps off the default stream and dropped the layer outputs it held:

  step                 trainer.step() on the default stream, then capture()            SIGSEGV in capture_end
                       (the graph was held by the encoder-layer forward hooks of the phased backward)
  step_clean           + hooks' outputs dropped, gc, empty_cache before capture()       ok
  noopt                forward + backward by hand on the default stream, the log-probs
                       (and with them the graph) kept in a variable                     SIGSEGV
  noopt_keeploss       the same, the loss kept as well                                  SIGSEGV
  noopt_side           the same on a side stream, graph kept                            ok
  step_noside          step with every side stream of the model switched off            SIGSEGV (the forks are not the cause)
  (the A -> B -> A script with B's eager step on a side stream: ok; without the eager step: ok)

What the package does about it: CaptionTrainer.step() moves itself to the process's warm-up stream when called on the default
stream, and no layer output (hence no autograd graph) outlives its step."""
import faulthandler, gc, sys, torch
faulthandler.enable()
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
from bmhrl_amd.functional import SCRATCH
from bmhrl_amd.model.bm_hrl_agent import BMEncoderLayer, BMFusionLayer, BMHrlAgent
v = sys.argv[1]
dev = torch.device("cuda:0")
b = syn.synthetic_batch(4, 128, 200, 14, 300, seed=3)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
if "noside" in v:
    BMEncoderLayer.modality_side_stream = False; BMFusionLayer.branch_side_stream = False; BMHrlAgent.critic_side_stream = False
t = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3, seed=5)
t.agent.train()
if "fwdonly" in v:
    with torch.no_grad():
        trg_in, trg_y, masks = t._head(fs, cap)
        t.agent(((fs["rgb"], fs["flow"]), fs["audio"]), trg_in, masks)
elif "noopt" in v:      # forward + backward on the default stream, no optimizer, inside begin/end step
    import contextlib
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with (torch.cuda.stream(side) if "side" in v else contextlib.nullcontext()):
        t.opt.zero_grad(); SCRATCH.begin_step(dev, t.scratch)
        trg_in, trg_y, masks = t._head(fs, cap)
        loss, kept = t._forward_loss(fs, trg_in, trg_y, None, masks)      # kept: the log-probs, attached to the graph
        t._backward(loss); SCRATCH.end_step(); t.opt.zero_grad()
    torch.cuda.current_stream().wait_stream(side)
    if 'keeploss' not in v:
        del loss
else:
    t.step(fs, cap)
torch.cuda.synchronize(); print("eager part done", flush=True)
if "clean" in v:
    t._layer_out.clear(); t.opt.zero_grad(); gc.collect(); torch.cuda.empty_cache()
if "noside" in v and "sideon" in v:
    BMEncoderLayer.modality_side_stream = True; BMFusionLayer.branch_side_stream = True; BMHrlAgent.critic_side_stream = True
t.capture(fs, cap, warmup=1)
print("captured", flush=True)
print([float(t.replay()) for _ in range(2)], flush=True)
