"""Tuning aid: what the gap between two graph launches costs -- the bench step captured once per graph (as shipped) against
two steps captured into one graph (a diagnostic patch of the trainer's capture bodies, not a product mode)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bmhrl_amd import synthetic as syn  # noqa: E402
from bmhrl_amd.train import CaptionTrainer  # noqa: E402

dev = torch.device("cuda:0")
b = syn.synthetic_batch(16, 256, 800, 30, 10172, seed=0)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)


def run(double):
    t = CaptionTrainer(syn.default_cfg(dout_p=0.1), 10172, dev, lr=1e-4)
    t.agent.train()
    t.agent.set_inference_mode(True)
    if double:
        body_a, body_b = t._graph_body_a, t._graph_body_b
        state = {"capturing": False}

        def b2(scale):
            body_b(scale)
            if torch.cuda.is_current_stream_capturing():
                body_a()
                body_b(scale)
        t._graph_body_b = b2
    t.capture(fs, cap, warmup=2)
    for _ in range(5):
        t.replay()
    torch.cuda.synchronize()
    n = 20 if double else 40
    t0 = time.perf_counter()
    for _ in range(n):
        t.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'two steps' if double else 'one step'} per graph: {dt / 40 * 1e3:.3f} ms per step", flush=True)


for d in (False, True, False, True):
    run(d)
