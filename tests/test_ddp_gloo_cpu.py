"""Data-parallel plumbing on CPU: two gloo ranks, flat gradient bucket all-reduce + Adam, must equal one process that
sees both ranks' gradients averaged (the RCCL path on the GPUs runs the same code with backend nccl)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmhrl_amd.train import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(6, 4)), torch.nn.Parameter(torch.randn(9))]
    opt = FlatAdam(ps, lr=1e-2)
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g)
        opt.gather_grads()
        scale = opt.all_reduce()
        assert scale == 1.0 / world
        opt.step(scale)
    out[rank] = opt.flat.clone()
    dist.destroy_process_group()


def test_two_rank_allreduce_adam_equals_averaged_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])            # replicas stay bit-identical
    from bmhrl_amd.train import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(6, 4)), torch.nn.Parameter(torch.randn(9))]
    opt = FlatAdam(ps, lr=1e-2)
    for step in range(3):
        gs = [torch.Generator().manual_seed(100 * step + r) for r in range(world)]
        for p in ps:
            p.grad = sum(torch.randn(p.shape, generator=g) for g in gs) / world
        opt.gather_grads(); opt.step()
    assert torch.allclose(opt.flat, out[0], atol=1e-6)


def _tw_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmhrl_amd.train import FlatAdam, token_weight
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(5, 3))
    opt = FlatAdam([w], lr=1e-2)
    opt.set_buckets([1])
    g = torch.Generator().manual_seed(7 + rank)
    n_rows = 4 + 5 * rank                                   # ranks see different token counts
    x = torch.randn(n_rows, 5, generator=g)
    rows = (x @ w).pow(2).sum(-1)                           # per-token losses of this rank
    loss = rows.sum() / n_rows * token_weight(torch.tensor(n_rows))
    loss.backward()
    opt.gather_grads(0)
    h = opt.all_reduce_part(0)                              # the phased path of the multi-rank step
    if h is not None:
        h.wait()
    opt.step(1.0 / world)
    out[rank] = (opt.flat.clone(), x)
    dist.destroy_process_group()


def test_token_weighted_ranks_equal_global_normalisation():
    """two ranks with different token counts: per-rank loss / n_r times token_weight, averaged gradients == the gradient of
    sum(all rows) / sum(all tokens) (the reference's DataParallel normalisation), through all_reduce_part + Adam"""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_tw_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0])
    from bmhrl_amd.train import FlatAdam
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(5, 3))
    opt = FlatAdam([w], lr=1e-2)
    x = torch.cat([out[0][1], out[1][1]])
    ((x @ w).pow(2).sum(-1).sum() / x.shape[0]).backward()
    opt.gather_grads(); opt.step()
    assert torch.allclose(opt.flat, out[0][0], atol=1e-6)
