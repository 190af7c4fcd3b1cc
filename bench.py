#!/usr/bin/env python3
"""Headline benchmark: caption-train steps/s of the BMHRL hot path on MI355X (BASELINE.json metric).

A step = one warmstart training step of the bimodal transformer (forward, label-smoothing KL loss, backward,
gradient all-reduce when N > 1, Adam) on one synthetic batch B=16 per GPU, Tv=256, Ta=800, L=30, V=10172, N=2 layers,
d_model=1024, H=4, dropout 0.1 in train mode (BASELINE.json configs[1]).  Inputs are resident in HBM before the timed
region.  One process per GPU (RCCL all-reduce of the flat gradient bucket): for N > 1 launch with torch.distributed.run, or
just `python bench.py --gpus N` -- without WORLD_SIZE in the environment the script starts its N ranks itself.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     -- the cross-modal attention kernels (V<-A: B16 H4 Sq256 Sk800; A<-V: Sq800 Sk256) with the step's padded
                  masks: flops EXECUTED per launch / mean launch time measured here with HIP events on the launch
                  stream, vs the 2.5 PFLOP/s dense bf16 MFMA peak (attention_roofline()),
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's arithmetic) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tv", type=int, default=256)
    ap.add_argument("--ta", type=int, default=800)
    ap.add_argument("--len", type=int, default=30)
    ap.add_argument("--vocab", type=int, default=10172)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--mode", choices=("warmstart", "rl"), default="warmstart",
                    help="warmstart = BASELINE configs[1] (the metric's config); rl = configs[2], the worker RL step of "
                         "train_bimodal_bl with synthetic rewards (biased KL of sampled tokens + value-head update)")
    ap.add_argument("--eager", action="store_true", help="launch kernels step by step instead of replaying the HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exploration", action="store_true",
                    help="A/B aid: the manager's exploration noise off (the reference's warmstart step has it ON, the default here)")
    ap.add_argument("--cpu-batch", type=int, default=16, help="batch of the bounded CPU-oracle sample (default: the full B=16 step)")
    return ap.parse_args()


def _time_launches(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def _pad_mask(B, S, dev):
    """key mask of the synthetic batch (bmhrl_amd.synthetic.synthetic_batch): the last (37 b) mod (S/4) positions of
    sample b are padding"""
    m = torch.ones(B, S, dtype=torch.bool)
    for b in range(B):
        r = (37 * b) % max(S // 4, 1)
        if r:
            m[b, S - r:] = False
    return m.to(dev)


def attention_roofline(dev, B, H, Tv, Ta, iters=50):
    """The two cross-modal attention launches of an encoder layer at the step's own shapes and with the step's own padded
    key masks, each timed with HIP events on the stream it is launched on (torch's current stream):

      V<-A  bmhrl_attention_shared128_fwd -> attn_fwd_kernel<128, ...>: queries = video rows (Sq = Tv), keys / values = the
            128-wide audio rows shared by all heads (absorbed-projection form).  The launch EXECUTES
            4*B*H*Sq*Sk*128 flops (half of the 4*B*Sq*Sk*1024 of the attention call it stands for);
      A<-V  bmhrl_attention_fwd -> attn_fwd_kernel<256, ...>: queries = audio rows (Sq = Ta), keys / values = video rows,
            d_k = 256: executes the full 4*B*Sq*Sk*1024.

    `frac` (the headline) = flops EXECUTED by the V<-A launch / its time / the 2.5 PFLOP/s dense bf16 MFMA peak (padded
    positions count as full work).  The A<-V launch, the call-equivalent figure of V<-A (the attention call's
    flops over kernel + the two query-side projection GEMMs the form adds, timed together) and the MFMA work actually
    issued (key tiles behind the last valid key of a batch row are not visited) are reported next to it."""
    from bmhrl_amd import ops
    dk, D, dm = 256, H * 256, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    scale = dk ** -0.5
    a_mask, v_mask = _pad_mask(B, Ta, dev), _pad_mask(B, Tv, dev)

    # ---- V<-A (head dimension 128, shared keys / values)
    Sq, Sk = Tv, Ta
    Qp = (0.5 * torch.randn(B, Sq, H, dm, generator=g)).to(dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, dm, generator=g).to(dev).to(torch.bfloat16)
    ctx = torch.empty(B, Sq, H, dm, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    va = lambda: ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, a_mask, Sk, B, H, Sq, Sk, scale, H * dm, dm, H * dm)
    sec_va = _time_launches(va, iters)
    # the call the launch stands for: Q'_h = Q_h Wk_h before, O_h = context_h Wv_h^T after (it removes the K|V projection
    # of the B*Sk audio rows), timed together with the kernel
    rows = B * Sq
    Qb = torch.randn(rows, D, generator=g).to(dev).to(torch.bfloat16)
    Wk = torch.randn(D, dm, generator=g).to(dev).to(torch.bfloat16)
    Ob = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    Qp2 = Qp.view(rows, H * dm)
    ctx2 = ctx.view(rows, H * dm)

    def va_call():
        ops.gemm(Qb, Wk, rows, dm, dk, lda=D, ldb=dm, b_trans=True, batch=(1, H), a_strides=(0, dk), b_strides=(0, dk * dm),
                 C_bf16=Qp2, ldcb=H * dm, cb_strides=(0, dm))
        va()
        ops.gemm(ctx2, Wk, rows, dk, dm, lda=H * dm, ldb=dm, batch=(1, H), a_strides=(0, dm), b_strides=(0, dk * dm),
                 C_bf16=Ob, ldcb=D, cb_strides=(0, dk))
    sec_call = _time_launches(va_call, iters)
    exec_va = 4.0 * B * H * Sq * Sk * dm
    call_va = 4.0 * B * Sq * Sk * D
    # MFMA work actually issued: the kernel visits the 64-key tiles up to the last valid key of a batch row
    visited = sum(min(Sk, -(-int(a_mask[b].sum()) // 64) * 64) for b in range(B))
    issued_va = 4.0 * H * Sq * visited * dm

    # ---- A<-V (head dimension 256)
    Sq2, Sk2 = Ta, Tv
    Q2 = torch.randn(B, Sq2, D, generator=g).to(dev).to(torch.bfloat16)
    K2 = torch.randn(B, Sk2, D, generator=g).to(dev).to(torch.bfloat16)
    V2 = torch.randn(B, Sk2, D, generator=g).to(dev).to(torch.bfloat16)
    O2 = torch.empty(B, Sq2, D, dtype=torch.bfloat16, device=dev)
    rmax2 = torch.empty(B, H, Sq2, device=dev)
    rsum2 = torch.empty(B, H, Sq2, device=dev)
    av = lambda: ops.attention_fwd(Q2, K2, V2, O2, rmax2, rsum2, v_mask, Sk2, 0, B, H, Sq2, Sk2, dk, scale, D, D, D, D)
    sec_av = _time_launches(av, iters)
    exec_av = 4.0 * B * Sq2 * Sk2 * D
    visited2 = sum(min(Sk2, -(-int(v_mask[b].sum()) // 32) * 32) for b in range(B))
    issued_av = 4.0 * Sq2 * visited2 * D

    # ---- the same two launches at the long-segment shapes of BASELINE configs[4] (B=8, Tv=1024, Ta=2048, unmasked): where
    # the fixed cost of a launch is amortised.  Secondary figures: the headline stays the step's own shape.
    def long_shapes():
        Bl, Tvl, Tal = 8, 1024, 2048
        Ql = (0.5 * torch.randn(Bl, Tvl, H, dm, generator=g)).to(dev).to(torch.bfloat16)
        Xl = torch.randn(Bl, Tal, dm, generator=g).to(dev).to(torch.bfloat16)
        cl = torch.empty(Bl, Tvl, H, dm, dtype=torch.bfloat16, device=dev)
        m1, s1 = torch.empty(Bl, H, Tvl, device=dev), torch.empty(Bl, H, Tvl, device=dev)
        t_va = _time_launches(lambda: ops.attention_shared128_fwd(Ql, Xl, cl, m1, s1, None, 0, Bl, H, Tvl, Tal, scale, H * dm, dm,
                                                                  H * dm), 20)
        Q3, K3, V3 = (torch.randn(Bl, S, D, generator=g).to(dev).to(torch.bfloat16) for S in (Tal, Tvl, Tvl))
        O3 = torch.empty(Bl, Tal, D, dtype=torch.bfloat16, device=dev)
        m3, s3 = torch.empty(Bl, H, Tal, device=dev), torch.empty(Bl, H, Tal, device=dev)
        t_av = _time_launches(lambda: ops.attention_fwd(Q3, K3, V3, O3, m3, s3, None, 0, 0, Bl, H, Tal, Tvl, dk, scale, D, D, D, D), 20)
        f_va, f_av = 4.0 * Bl * H * Tvl * Tal * dm, 4.0 * Bl * Tal * Tvl * D
        return {"shape": {"B": Bl, "Tv": Tvl, "Ta": Tal, "mask": "none"},
                "V<-A": {"launch_us": t_va * 1e6, "flops_per_launch": f_va, "frac": f_va / t_va / 2.5e15},
                "A<-V": {"launch_us": t_av * 1e6, "flops_per_launch": f_av, "frac": f_av / t_av / 2.5e15}}
    long_seq = long_shapes()

    traffic = None   # HBM bytes per launch from rocprofv3 PMC passes of this kernel at this shape (profiles/)
    tj = os.path.join(ROOT, "profiles", "r03_attn_traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        if t.get("shape") == {"B": B, "H": H, "Sq": Sq, "Sk": Sk, "d_k": dk}:
            traffic = t["traffic_bytes_per_launch"]
    tf = lambda fl, sec: fl / sec / 1e12
    return {
        "bound": "mfma",
        "kernel": "attn_fwd_kernel<128> = bmhrl_attention_shared128_fwd, the cross-modal V<-A attention (absorbed-projection "
                  "form: the 128-wide audio rows are keys and values of all heads); flops = those the launch executes",
        "achieved": tf(exec_va, sec_va), "peak": 2500.0, "unit": "TFLOP/s", "frac": tf(exec_va, sec_va) / 2500.0,
        "traffic": traffic, "launch_us": sec_va * 1e6, "flops_per_launch": exec_va,
        "shape": {"B": B, "H": H, "Sq": Sq, "Sk": Sk, "d_k": dk, "mask": "padded tails of the synthetic batch"},
        "mfma_issued_frac": tf(issued_va, sec_va) / 2500.0,
        "cross_modal_A<-V": {"kernel": "attn_fwd_sk256_kernel = bmhrl_attention_fwd for Sk <= 256 (Sq = Ta, Sk = Tv, d_k = 256; BMHRL_ATTN_SK256=0: attn_fwd_kernel<256>)",
                             "launch_us": sec_av * 1e6, "flops_per_launch": exec_av, "achieved": tf(exec_av, sec_av),
                             "frac": tf(exec_av, sec_av) / 2500.0, "mfma_issued_frac": tf(issued_av, sec_av) / 2500.0},
        "call_equivalent_V<-A": {"what": "4*B*Sq*Sk*1024 flops of the attention call the launch stands for, over the kernel + "
                                         "the Q_h Wk_h and context Wv_h^T GEMMs the form adds, timed together (the form also "
                                         "removes the 2*B*Sk*128*2048-flop K|V projection of the audio rows and its backward)",
                                 "us": sec_call * 1e6, "flops": call_va, "achieved": tf(call_va, sec_call),
                                 "frac": tf(call_va, sec_call) / 2500.0},
        "config5_shapes": long_seq,
    }


def cpu_baseline(args):
    """The oracle's warmstart step (forward + loss + backward + Adam) on the host cores, on a bounded sample: a batch
    of --cpu-batch (default: the full batch; smaller values are scaled linearly, the cost is linear in B), one warm-up +
    four timed steps -- about 15 s of CPU work on the GPU box's 16-thread share."""
    from bmhrl_amd import synthetic as syn
    from bmhrl_amd.model.bm_hrl_agent import agent_state_shapes
    from oracle import bmhrl_oracle as O
    # the GPU box gives one GPU a share of 16 host CPUs (cgroup quota, not visible in os.cpu_count()): more threads
    # than that only oversubscribe the quota
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle step on {cores} host threads ...", file=sys.stderr, flush=True)
    cfg = syn.default_cfg(dout_p=0.0, rl_att_layers=args.layers)
    Bc = max(2, min(args.cpu_batch, args.batch))
    sd = syn.fill_state_dict(agent_state_shapes(cfg, args.vocab, with_critic=False), seed=0, clone_layers=True)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    skip = ("critic.", "manager_core.", "manager.core.")
    params = [k for k in sd if not k.startswith(skip) and not (".feed_forward.fc" in k and "_fus." in k)]
    for k in params:
        sd[k] = sd[k].clone().requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in params], lr=1e-4)
    b = syn.synthetic_batch(Bc, args.tv, args.ta, args.len, args.vocab, seed=0)
    cap = b["captions"]
    trg_in, trg_y = cap[:, :-1], cap[:, 1:]
    masks = O.make_masks(b["rgb"], b["audio"], trg_in, 1)

    def one():
        opt.zero_grad()
        pred = O.agent_forward(sd, cfg, (b["rgb"] + b["flow"], b["audio"]), trg_in, masks)[0]
        O.warmstart_loss(pred, trg_y, 0.7, 1).backward()
        opt.step()

    one()
    print("[bench] cpu_baseline: warm-up done", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        one()
    dt = (time.perf_counter() - t0) / n
    steps_per_s = 1.0 / (dt * args.batch / Bc)
    return {"value": steps_per_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle warmstart step (fwd+loss+bwd+Adam, fp32, torch CPU {torch.__version__}) at batch {Bc} of "
                      f"{args.batch}, same shapes; 1 warm-up + {n} timed steps, {dt:.2f} s per sampled step; value scaled "
                      f"linearly to B={args.batch}"}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher (the reference needs none either: nn.DataParallel,
    scripts/train_rl_captioning_module.py:95-99): start the N ranks as children under torch.distributed.run -- before this
    process touches the GPU -- and leave with their exit code.  The JSON line is rank 0's."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()                 # (counting devices does not initialise the GPU)
    if n_dev < args.gpus and os.environ.get("BMHRL_BENCH_ONE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible -- refusing to report a smaller job as {args.gpus} GPUs")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] WORLD_SIZE unset: starting {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal aid for the multi-rank path on a one-GPU box: BMHRL_BENCH_BACKEND=gloo BMHRL_BENCH_ONE_DEVICE=1 runs every
    # rank on cuda:0 with gloo collectives (RCCL refuses two ranks on one device); never used by the driver
    backend = os.environ.get("BMHRL_BENCH_BACKEND", "nccl")
    if os.environ.get("BMHRL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BMHRL_BENCH_FORCE_PG=1: a one-rank RCCL process group (rehearses capture / replay next to RCCL's own threads and
    # streams on a one-GPU box; with BMHRL_SPLIT_BACKWARD=1 the step takes the multi-rank path)
    if world > 1 or os.environ.get("BMHRL_BENCH_FORCE_PG") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    from bmhrl_amd import _lib, synthetic as syn
    from bmhrl_amd.train import CaptionTrainer
    _lib.load()

    cfg = syn.default_cfg(dout_p=args.dropout, rl_att_layers=args.layers)
    rl = args.mode == "rl"
    b = syn.synthetic_batch(args.batch, args.tv, args.ta, args.len, args.vocab, seed=rank)
    rewards = syn.synthetic_rewards(args.batch, b["captions"].shape[1] - 1, seed=2 + rank).to(dev) if rl else None
    trainer = CaptionTrainer(cfg, args.vocab, dev, lr=1e-4, phase="worker" if rl else "warmstart",
                             reward_fn=(lambda sampled, captions: rewards) if rl else None,
                             exploration=False if args.no_exploration else None)
    trainer.agent.train()
    if rl:
        trainer.value_net.train()
    fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
    cap = b["captions"].to(dev)

    use_graph = not args.eager
    if use_graph:
        trainer.capture(fs, cap, warmup=2)
        step = lambda: trainer.replay()
    else:
        step = lambda: trainer.step(fs, cap)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_v = float(loss)

    if rank == 0:
        roof = attention_roofline(dev, args.batch, cfg.rl_att_heads, args.tv, args.ta)
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args)   # N=1 only (bench contract)
        out = {
            "metric": "caption-train steps/sec", "value": args.steps * world / elapsed, "unit": "steps/s (one step = one "
            "B=16 batch on one GPU; whole-job aggregate over all GPUs)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BMHRL worker RL step (BASELINE configs[2], synthetic rewards)" if rl else
                                    "BMHRL warmstart step (BASELINE configs[1])") + f": B={args.batch}/GPU Tv={args.tv} Ta={args.ta} "
                                   f"L={args.len} V={args.vocab} N={args.layers} d_model=1024 H=4 dropout={args.dropout}",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "launch": "hip-graph" if use_graph else "eager", "samples_per_s": args.steps * world * args.batch / elapsed},
            "loss": loss_v, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()      # rank 0 is still measuring the roofline kernel: keep the communicator alive until it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
