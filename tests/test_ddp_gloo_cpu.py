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


def _homes_params():
    torch.manual_seed(0)
    shapes = [(8, 4), (4,), (8, 4), (4,), (12, 4), (16,)]          # w_a, b_a, w_b, b_b (stacked pair) | w_c, lone
    return [torch.nn.Parameter(torch.randn(*s)) for s in shapes]


def _homes_grads(step, rank):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(16, 4, generator=g), torch.randn(8, generator=g), torch.randn(12, 4, generator=g), torch.randn(16, generator=g)


def _homes_worker(rank, world, port, out):
    """two ranks; after FlatAdam.adopt_homes the "kernels" (here: plain copies) write the gradients straight into the bucket runs
    the allocator hands out, the buckets are all-reduced one by one and Adam runs on the bucket -- no gather copy of those"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmhrl_amd.functional import ScratchState
    from bmhrl_amd.train import FlatAdam
    ps = _homes_params()
    opt = FlatAdam(ps, lr=1e-2)
    opt.set_buckets([4, 2])
    st = ScratchState()
    # the recording pass: one stacked allocation for [w_a; w_b], one for [b_a; b_b], one for w_c; `lone` comes from elsewhere
    sw, sb, wc, lone = _homes_grads(0, rank)
    st.log = [(0, 0, 64, sw.data_ptr(), sw), (0, 64, 8, sb.data_ptr(), sb), (1, 0, 48, wc.data_ptr(), wc)]
    for p, g in zip(ps, (sw[:8], sb[:4], sw[8:], sb[4:], wc, lone)):
        p.grad = g
    assert opt.adopt_homes(st) == 64 + 8 + 48
    copied = 0
    for step in range(3):
        sw, sb, wc, lone = _homes_grads(step, rank)
        opt.grad.zero_()                                     # (StepScratch.begin_step does this for the home buckets)
        st.homes[(0, 0)].copy_(sw.reshape(-1)); st.homes[(0, 64)].copy_(sb); st.homes[(1, 0)].copy_(wc.reshape(-1))
        for p in ps[:5]:
            k = [i for i, q in enumerate(opt.params) if q is p][0]
            p.grad = opt.grad_views[k]                       # what the step's allocator handed to the producing kernel
        ps[5].grad = lone
        before = opt.grad.clone()
        works = []
        for part in (0, 1):
            opt.gather_grads(part)
            works.append(opt.all_reduce_part(part))
        copied += int((opt.grad != before).sum())            # only `lone` (16 elements) may have been copied in
        for w in works:
            if w is not None:
                w.wait()
        opt.step(1.0 / world)
    out[rank] = (opt.in_param_order(opt.flat).clone(), copied)
    dist.destroy_process_group()


def test_two_ranks_with_gradients_produced_in_the_bucket_equal_one_process():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_homes_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] <= 3 * 16
    from bmhrl_amd.train import FlatAdam
    ps = _homes_params()
    opt = FlatAdam(ps, lr=1e-2)
    for step in range(3):
        per_rank = [_homes_grads(step, r) for r in range(world)]
        sw, sb, wc, lone = (sum(t) / world for t in zip(*per_rank))
        for p, g in zip(ps, (sw[:8], sb[:4], sw[8:], sb[4:], wc, lone)):
            p.grad = g.clone()
        opt.gather_grads(); opt.step()
    assert torch.allclose(opt.in_param_order(opt.flat), out[0][0], atol=1e-6)
