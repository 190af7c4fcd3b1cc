"""Probe: serial run, parallel-branches run, serial run again in one process (fresh agent each): which gradients differ beyond
the order-of-atomics noise (~1e-7), and between which runs?"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_streams_gpu import _run


def worst(a, b):
    w = sorted(((float((a[k] - b[k]).norm() / (a[k].norm() + 1e-6 * a[k].numel() ** 0.5)), k) for k in a), reverse=True)[:2]
    return [(f"{x:.1e}", k.replace("bm_enc.encoder.layers.", "enc.")) for x, k in w]


p0, s0, l0, g0 = _run(False)
p1, s1, l1, g1 = _run(True)
p2, s2, l2, g2 = _run(False)
print("serial0 vs parallel:", worst(g0, g1))
print("serial0 vs serial2 :", worst(g0, g2))
print("parallel vs serial2:", worst(g1, g2))
