"""The multi-rank step of CaptionTrainer (phased backward, one all-reduce per bucket while the next phase runs, global
token normalisation, fused Adam) rehearsed with TWO ranks that share the one GPU of the test box over gloo (RCCL needs
one GPU per rank; the code path is the same, only the backend string differs: bench.py, BMHRL_BENCH_BACKEND).

Checked: the two replicas stay bit-identical, and their step equals ONE process that sees the concatenated batch -- the
reference's DataParallel semantics (loss normalised by the token count of the whole batch)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

V, TV, TA, L, B_RANK = 60, 160, 200, 8, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cfg():
    from bmhrl_amd import synthetic as syn
    cfg = syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0)
    return cfg


def _batch(seeds, dev):
    from bmhrl_amd import synthetic as syn
    cfg = _cfg()
    parts = [syn.synthetic_batch(B_RANK, TV, TA, L, V, seed=s, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=3) for s in seeds]
    return {k: torch.cat([p[k] for p in parts]).to(dev) for k in ("rgb", "flow", "audio", "captions")}


def _reward(sampled, captions):
    """deterministic stand-in for the scorer (BASELINE configs[2] stubs the rewards): a function of the drawn tokens only"""
    return (sampled % 17).float() / 17.0


def _run(tr, b):
    fs = {k: b[k] for k in ("rgb", "flow", "audio")}
    tr.capture(fs, b["captions"], warmup=1)          # one real (eager-bodied) step, then the captured one
    loss = tr.replay()
    torch.cuda.synchronize()
    return float(loss)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmhrl_amd.train import CaptionTrainer
    dev = torch.device("cuda:0")
    tr = CaptionTrainer(_cfg(), V, dev, exploration=False, seed=0)
    assert tr._split()                               # more than one rank: phased backward + per-bucket all-reduce
    b = _batch([40 + rank], dev)
    loss = _run(tr, b)
    n_tok = int((b["captions"][:, 1:] != 1).sum())
    out[rank] = (tr.opt.in_param_order(tr.opt.flat).cpu(), (tr.opt.in_param_order(tr.opt.grad) / world).cpu(), loss, n_tok,
                 float(tr.loss_weight), getattr(tr, "grad_elems_in_place", 0) / tr.opt.n)
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    flat0, grad0, loss0, n0, w0, in_place = out[0]
    flat1, grad1, loss1, n1, w1, _ = out[1]
    assert in_place > 0.95           # the leaf gradients are produced in the flat bucket (FlatAdam.adopt_homes): no 221 MB gather
    assert torch.equal(flat0, flat1) and torch.equal(grad0, grad1)          # replicas: bit-identical
    assert n0 != n1                                                          # the ranks really see different token counts
    assert abs(w0 - 2 * n0 / (n0 + n1)) < 1e-6 and abs(w1 - 2 * n1 / (n0 + n1)) < 1e-6

    from bmhrl_amd.train import CaptionTrainer
    dev = torch.device("cuda:0")
    tr = CaptionTrainer(_cfg(), V, dev, exploration=False, seed=0)
    tr.split_backward = False
    tr.opt.direct_grads = False          # this check reads the gathered bucket (one process normally leaves the gradients in place)
    b = _batch([40, 41], dev)
    loss = _run(tr, b)
    ref_grad = tr.opt.in_param_order(tr.opt.grad).cpu()
    # loss of the whole batch = token-weighted mean of the ranks' losses (their reported loss carries the weight already)
    assert abs(loss - 0.5 * (loss0 + loss1)) <= 2e-3 * abs(loss)
    err = float((grad0 - ref_grad).norm() / ref_grad.norm())
    assert err <= 3e-2, err                                                   # bf16 operands, different tile paths for B=2 / B=4
    upd = float((flat0 - tr.opt.in_param_order(tr.opt.flat).cpu()).abs().max())
    assert upd <= 2.5e-4, upd                                                 # two Adam steps of lr 1e-4: at most 2e-4 apart


def _rl_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmhrl_amd.train import CaptionTrainer
    dev = torch.device("cuda:0")
    tr = CaptionTrainer(_cfg(), V, dev, exploration=False, seed=0, phase="worker", reward_fn=_reward)
    assert tr._split()                               # the RL phases take the phased backward too
    b = _batch([40 + rank], dev)
    loss = _run(tr, b)
    out[rank] = (tr.opt.in_param_order(tr.opt.flat).cpu(), (tr.opt.in_param_order(tr.opt.grad) / world).cpu(),
                 (tr.vopt.in_param_order(tr.vopt.grad) / world).cpu(), tr.vopt.in_param_order(tr.vopt.flat).cpu(), loss,
                 float(tr.last_value_loss))
    dist.destroy_process_group()


def test_two_ranks_worker_rl_phase_equal_one_process():
    """phase="worker" (train_bimodal_bl: sampled tokens, biased KL / (n_tokens 4/20) + the value head's masked MSE): two ranks ==
    one process on the concatenated batch, for BOTH optimisers.  The captioning term carries the token weight, the value loss
    (a plain mean over B x L) does not; rank r draws the uniforms of the global rows r B L ... (bmhrl_sample_tokens row_offset)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_rl_worker, args=(world, port, out), nprocs=world, join=True)
    flat0, grad0, vgrad0, vflat0, loss0, vl0 = out[0]
    flat1, grad1, vgrad1, vflat1, loss1, vl1 = out[1]
    assert torch.equal(flat0, flat1) and torch.equal(grad0, grad1) and torch.equal(vgrad0, vgrad1) and torch.equal(vflat0, vflat1)

    from bmhrl_amd.train import CaptionTrainer
    dev = torch.device("cuda:0")
    tr = CaptionTrainer(_cfg(), V, dev, exploration=False, seed=0, phase="worker", reward_fn=_reward)
    tr.split_backward = False
    tr.opt.direct_grads = tr.vopt.direct_grads = False
    b = _batch([40, 41], dev)
    _run(tr, b)
    ref_grad, ref_vgrad = tr.opt.in_param_order(tr.opt.grad).cpu(), tr.vopt.in_param_order(tr.vopt.grad).cpu()
    assert abs(float(tr.last_value_loss) - 0.5 * (vl0 + vl1)) <= 2e-2 * abs(float(tr.last_value_loss))
    err = float((grad0 - ref_grad).norm() / ref_grad.norm())
    verr = float((vgrad0 - ref_vgrad).norm() / ref_vgrad.norm())
    assert err <= 3e-2 and verr <= 5e-2, (err, verr)
    assert float((flat0 - tr.opt.in_param_order(tr.opt.flat).cpu()).abs().max()) <= 2.5e-4
    assert float((vflat0 - tr.vopt.in_param_order(tr.vopt.flat).cpu()).abs().max()) <= 2.5e-4
