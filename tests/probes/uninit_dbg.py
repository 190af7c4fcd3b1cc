"""Probe: does any kernel of the (unarmed) forward + backward read memory nobody wrote?  The caching allocator's free blocks are
poisoned with NaN bit patterns before a run; a gradient that turns NaN (or differs from the unpoisoned run) names the reader."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_streams_gpu import _run


def poison():
    torch.cuda.synchronize()
    blocks = []
    try:
        for _ in range(24):
            blocks.append(torch.full((1 << 26,), float("nan"), device="cuda:0"))       # 256 MB each
    except RuntimeError:
        pass
    del blocks
    small = []
    for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18):      # the small-block pool too (1 KB .. 1 MB blocks)
        small += [torch.full((n,), float("nan"), device="cuda:0") for _ in range(512)]
    del small
    torch.cuda.synchronize()


p0, s0, l0, g0 = _run(False)
p1, s1, l1, g1 = _run(False)
poison()
p2, s2, l2, g2 = _run(False)
bad = [k for k in g2 if not bool(torch.isfinite(g2[k]).all())]
print("non-finite after poison:", bad[:10], "loss", l2, "pred finite", bool(torch.isfinite(p2).all()))
w = sorted(((float((g1[k] - g2[k]).norm() / (g1[k].norm() + 1e-6 * g1[k].numel() ** 0.5)), k) for k in g1 if k not in bad), reverse=True)[:5]
print("run1 vs poisoned run:", w)
