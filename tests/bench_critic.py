"""Tuning aid: the segment critic alone (B=16, L=30, d=300, H=600), wavefront vs layer-by-layer, in a HIP graph."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import synthetic as syn
from bmhrl_amd.model.bm_hrl_agent import SegmentCritic
dev = torch.device("cuda:0")
c = SegmentCritic(syn.default_cfg()); c.load_state_dict(syn.synthetic_critic_state(300, seed=1)); c = c.to(dev)
emb = (torch.randn(16, 30, 300) * 17.3).to(dev)
for wf, ch in ((True, 1), (True, 2), (True, 3), (True, 5), (False, 1), (True, 3)):
    c.wavefront = wf
    c.wave_chunk = ch
    c.score_and_labels(emb, 0.25); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            c.score_and_labels(emb, 0.25)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"critic wavefront={wf} chunk={ch}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
