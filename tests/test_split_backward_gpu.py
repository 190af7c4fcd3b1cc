"""Phased backward of the trainer (data-parallel runs: head + fusion stacks, then one phase per encoder layer; the all-reduce
of a phase's gradient bucket overlaps the next phase's backward).  On one GPU the split must reproduce the single-phase step: same losses and parameters."""
import pytest
import torch

from bmhrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def test_split_backward_equals_single_phase():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd.train import CaptionTrainer
    dev = torch.device("cuda:0")
    b = syn.synthetic_batch(2, 128, 200, 12, 300, seed=2)
    fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
    cap = b["captions"].to(dev)
    runs = []
    for split in (False, False, True):
        t = CaptionTrainer(syn.default_cfg(dout_p=0.0), 300, dev, exploration=False, lr=1e-3)
        t.agent.train()
        t.split_backward = split
        init = t.opt.flat.clone()
        t.capture(fs, cap, warmup=1)
        losses = [float(t.replay()) for _ in range(3)]
        assert hasattr(t, "graph_a2") == split
        if split:
            assert len(t.graph_a2) == 2 and len(t.opt.bucket_bounds) == 4          # N = 2 encoder layers -> 3 buckets
            assert all(b > a for a, b in zip(t.opt.bucket_bounds, t.opt.bucket_bounds[1:]))
        runs.append((losses, t.opt.flat.clone(), t.opt.split_off, t.opt.n))
        bounds1 = [t.opt._elem_off(i) for i in t.opt.bucket_bounds]           # element ranges of the buckets (same order in every run)
    (l0, p0, _, _), (l0b, p0b, _, _), (l1, p1, off, n) = runs
    assert 0 < off < n                                   # both buckets are non-empty
    # Separately built trainers differ by the arrival order of fp32 atomics (split-K weight gradients, column sums), and
    # Adam turns the sign noise of near-zero gradients (key biases) into +-lr moves: the split run must sit inside the
    # spread of two single-phase runs.
    # Measured on MI355X over several boxes (tests/probes/diag_split.py): single vs single 0.6e-3 .. 1.6e-3 of |p|, single vs
    # split 1.4e-3 .. 1.8e-3 -- one population (five Adam steps of lr 1e-3 move a parameter whose gradient sign is noise by up
    # to 5e-3 against weights of ~3e-2); a phase that lost its gradients or its update would show as >= 1e-1 on its bucket.
    spread_l = max(abs(a - c) / abs(a) for a, c in zip(l0, l0b))
    spread_p = float((p0 - p0b).norm() / p0.norm())
    assert all(abs(a - c) / abs(a) < max(1e-3, 3 * spread_l) for a, c in zip(l0, l1)), (l0, l0b, l1)
    d_split = float((p0 - p1).norm() / p0.norm())
    assert d_split < max(3e-3, 3 * spread_p), (spread_p, d_split)
    # every bucket of the split run moved as far from the initial weights as the single-phase run did (no phase was dropped)
    for lo, hi in zip(bounds1, bounds1[1:]):
        moved0 = float((p0[lo:hi] - init[lo:hi]).norm())
        moved1 = float((p1[lo:hi] - init[lo:hi]).norm())
        assert moved0 > 0 and abs(moved1 - moved0) <= 5e-2 * moved0, (lo, hi, moved0, moved1)


_DET_SCRIPT = r"""
import sys, torch
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
dev = torch.device("cuda:0")
b = syn.synthetic_batch(2, 128, 200, 12, 300, seed=2)
fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
cap = b["captions"].to(dev)
t = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3)
t.agent.train()
t.split_backward = sys.argv[2] == "split"
t.capture(fs, cap, warmup=1)
losses = [float(t.replay()) for _ in range(3)]
torch.save({"losses": losses, "flat": t.opt.in_param_order(t.opt.flat).cpu(), "in_place": getattr(t, "grad_elems_in_place", 0), "n": t.opt.n,
            "early": sorted(t._early_done)}, sys.argv[1])
"""


def test_deterministic_mode_runs_agree_bit_for_bit(tmp_path):
    """BMHRL_DETERMINISTIC=1 (no K split, ordered column sums / LayerNorm parameter gradients / embedding and scatter
    gradients): two separately started runs of the same captured step -- dropout on -- end with IDENTICAL losses and weights,
    and the phased backward equals the single-phase one exactly (the default mode only agrees up to the order of its atomics)."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMHRL_DETERMINISTIC="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = []
    for i, mode in enumerate(("single", "single", "split")):
        f = tmp_path / f"run{i}.pt"
        r = subprocess.run([sys.executable, "-c", _DET_SCRIPT, str(f), mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    assert outs[0]["losses"] == outs[1]["losses"] and torch.equal(outs[0]["flat"], outs[1]["flat"])
    assert outs[0]["losses"] == outs[2]["losses"] and torch.equal(outs[0]["flat"], outs[2]["flat"])
    assert all(x == x for x in outs[0]["losses"]) and outs[0]["losses"][2] < outs[0]["losses"][0]


def test_gradients_produced_in_the_flat_bucket_change_nothing(tmp_path):
    """Data-parallel runs write the leaf gradients straight into their slices of the optimiser's flat bucket
    (FlatAdam.adopt_homes, on by itself with more than one rank); rehearsed on one rank with BMHRL_GRAD_HOMES=1 and the gather
    path forced (BMHRL_DIRECT_GRADS=0, what a multi-rank step runs): in deterministic mode losses and weights are IDENTICAL to
    the run that gathers every gradient, for the single-phase and the phased backward, and nearly every element skips the copy."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, BMHRL_DETERMINISTIC="1", BMHRL_DIRECT_GRADS="0",
                PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = {}
    for homes in ("0", "1"):
        for mode in ("single", "split"):
            f = tmp_path / f"run_{homes}_{mode}.pt"
            r = subprocess.run([sys.executable, "-c", _DET_SCRIPT, str(f), mode], env=dict(base, BMHRL_GRAD_HOMES=homes),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[homes, mode] = torch.load(f)
    ref = outs["0", "single"]
    for key, o in outs.items():
        assert o["losses"] == ref["losses"] and torch.equal(o["flat"], ref["flat"]), key
    assert outs["0", "single"]["in_place"] == 0
    for mode in ("single", "split"):
        assert outs["1", mode]["in_place"] > 0.95 * outs["1", mode]["n"], (mode, outs["1", mode]["in_place"], outs["1", mode]["n"])


_ABA_SCRIPT = r"""
import sys, torch
from bmhrl_amd import synthetic as syn
from bmhrl_amd.train import CaptionTrainer
dev = torch.device("cuda:0")
def batch(B, Tv, Ta, L, seed):
    b = syn.synthetic_batch(B, Tv, Ta, L, 300, seed=seed)
    return {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}, b["captions"].to(dev)
fa, ca = batch(2, 64, 100, 10, 2)
A = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3)
A.agent.train()
A.capture(fa, ca, warmup=1)
losses = [float(A.replay()) for _ in range(2)]
if sys.argv[2] == "with_b":
    # a second, LARGER trainer: its own scratch arena / operand pools / graph; eager steps and a capture of its own
    fb, cb = batch(4, 128, 200, 14, 3)
    Bt = CaptionTrainer(syn.default_cfg(dout_p=0.1), 300, dev, exploration=False, lr=1e-3, seed=5)
    Bt.agent.train()
    Bt.step(fb, cb)
    Bt.capture(fb, cb, warmup=1)
    lb = [float(Bt.replay()) for _ in range(2)]
    assert all(x == x for x in lb)
losses += [float(A.replay()) for _ in range(2)]
torch.cuda.synchronize()
torch.save({"losses": losses, "flat": A.opt.in_param_order(A.opt.flat).cpu()}, sys.argv[1])
"""


def test_a_captured_trainer_survives_a_larger_second_trainer(tmp_path):
    """Capture trainer A, replay; build, step and capture a LARGER trainer B (its own ScratchState: arena, operand pools,
    sync words); replay A again: A's graph must still read and write memory that is its own.  Deterministic mode, so the
    check is equality with a run in which B never existed."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMHRL_DETERMINISTIC="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = {}
    for mode in ("alone", "with_b"):
        f = tmp_path / f"aba_{mode}.pt"
        r = subprocess.run([sys.executable, "-c", _ABA_SCRIPT, str(f), mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    assert all(x == x for x in outs["alone"]["losses"])
    assert outs["alone"]["losses"] == outs["with_b"]["losses"]
    assert torch.equal(outs["alone"]["flat"], outs["with_b"]["flat"])


def test_early_adam_passes_change_nothing(tmp_path):
    """One rank: the optimizer pass of a part starts on a side stream as soon as the backward has gone past it (tensor hooks on
    the encoder-layer outputs, CaptionTrainer._early_fire).  Deterministic mode: losses and weights after three replays are
    IDENTICAL to the run with the whole update behind the backward -- a part updated before its gradients were complete, or
    a weight rewritten while a later backward kernel still reads it, would show."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, BMHRL_DETERMINISTIC="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = {}
    for early in ("0", "1"):
        f = tmp_path / f"early_{early}.pt"
        r = subprocess.run([sys.executable, "-c", _DET_SCRIPT, str(f), "single"], env=dict(base, BMHRL_EARLY_ADAM=early),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[early] = torch.load(f)
    assert outs["0"]["early"] == [] and outs["1"]["early"] == [0, 2]      # head + fusion stacks, then encoder layer 1
    assert outs["0"]["losses"] == outs["1"]["losses"] and torch.equal(outs["0"]["flat"], outs["1"]["flat"])
