import sys, torch
sys.path.insert(0, '/root/repo')
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
dev = torch.device("cuda:0")
b = syn.synthetic_batch(2, 128, 200, 12, 300, seed=2)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
runs = []
for split in (False, False, True):
    t = CaptionTrainer(syn.default_cfg(dout_p=0.0), 300, dev, lr=1e-3)
    t.agent.train(); t.split_backward = split
    t.capture(fs, cap, warmup=2)
    losses = [float(t.replay()) for _ in range(3)]
    names = {id(p): n for n, p in t.agent.named_parameters()}
    sl = {names[id(p)]: (o, s) for p, o, s in zip(t.opt.params, t.opt.offsets, t.opt.sizes)}
    runs.append((losses, t.opt.flat.clone(), sl))
(l0, p0, sl), (l0b, p0b, _), (l1, p1, sl1) = runs
def d(a, b, sl_a, sl_b, pred):
    num = den = 0.0
    for n, (o, s) in sl_a.items():
        if not pred(n): continue
        o2, _ = sl_b[n]
        num += float((a[o:o+s] - b[o2:o2+s]).pow(2).sum()); den += float(a[o:o+s].pow(2).sum())
    return (num / den) ** 0.5
for tag, pred in (("all", lambda n: True), ("no K2d.bias", lambda n: not n.endswith("linear_K2d.bias")), ("only K2d.bias", lambda n: n.endswith("linear_K2d.bias")),
                  ("weights only", lambda n: n.endswith(".weight"))):
    print(tag, "single vs single", d(p0, p0b, sl, sl, pred), "single vs split", d(p0, p1, sl, sl1, pred))
print(l0, l0b, l1)
