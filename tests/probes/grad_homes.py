"""Diagnostic: which leaf gradients FlatAdam.adopt_homes could not place in the flat bucket, and why."""
import os, sys, bisect, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BMHRL_GRAD_HOMES"] = "1"; os.environ["BMHRL_DIRECT_GRADS"] = "0"
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer, FlatAdam
dev = torch.device("cuda:0")
b = syn.synthetic_batch(2, 128, 200, 12, 300, seed=2)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
t = CaptionTrainer(syn.default_cfg(dout_p=0.0), 300, dev, exploration=False, lr=1e-3)
t.agent.train()
names = {id(p): n for n, p in t.agent.named_parameters()}
orig = FlatAdam.adopt_homes
def spy(self, state):
    log = sorted(state.log, key=lambda r: r[3]); starts = [r[3] for r in log]
    rows = []
    for p in self.params:
        g = p.grad
        if g is None: rows.append((p.numel(), names.get(id(p)), "no grad")); continue
        j = bisect.bisect_right(starts, g.data_ptr()) - 1
        inside = j >= 0 and g.data_ptr() + 4 * g.numel() <= log[j][3] + 4 * log[j][2]
        rows.append((p.numel(), names.get(id(p)), ("in alloc of %d at +%d" % (log[j][2], (g.data_ptr() - log[j][3]) // 4)) if inside else
                     "outside every allocation (contig=%s)" % g.is_contiguous()))
    placed = orig(self, state)
    homes = {v.data_ptr(): v.numel() for v in state.homes.values()}
    print("placed", placed, "of", self.n)
    for n, name, why in sorted(rows, reverse=True)[:400]:
        p = [q for q in self.params if names.get(id(q)) == name][0]
        k = [i for i, q in enumerate(self.params) if q is p][0]
        at_home = any(h <= self.grad_views[k].data_ptr() < h + 4 * m for h, m in homes.items())
        if not at_home and n >= 1024:
            print("%9d %-70s %s" % (n, name, why))
    return placed
FlatAdam.adopt_homes = spy
t.capture(fs, cap, warmup=1)
print("in place", t.grad_elems_in_place)
