"""The per-batch steps (warmstart, worker RL, validation, decoders) through the reference's loop signatures on the GPU,
with a synthetic loader that honours the reference's batch contract and synthetic rewards
(BASELINE config 3)."""
from types import SimpleNamespace

import math

import pytest
import torch

from bmhrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu


class SynthDataset:
    pad_idx, start_idx, end_idx, phase = 1, 2, 3, "train"

    def __init__(self, cfg, V, n_batches, B, Tv, Ta, L, dev):
        self.train_vocab = SimpleNamespace(itos=[f"w{i}" for i in range(V)], vectors=None)
        self.trg_voc_size = V
        self.batches = []
        for i in range(n_batches):
            b = syn.synthetic_batch(B, Tv, Ta, L, V, seed=10 + i, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=3)
            self.batches.append({
                "video_ids": [f"v{j}" for j in range(B)], "captions": ["a b c"] * B,
                "starts": torch.zeros(B, 1), "ends": torch.ones(B, 1),
                "feature_stacks": {k: b[k].to(dev) for k in ("rgb", "flow", "audio")},
                "caption_data": SimpleNamespace(caption=b["captions"].to(dev), idx=torch.arange(B)),
            })

    def update_iterator(self):
        pass


class SynthLoader:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        return iter(self.dataset.batches)

    def __len__(self):
        return len(self.dataset.batches)


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bmhrl_amd.install  # noqa: F401  (reference import paths)
    from model.bm_hrl_agent import BMHrlAgent, BMWorkerValueFunction
    from loss.label_smoothing import LabelSmoothing
    from loss.biased_kl import BiasedKL
    dev = torch.device("cuda:0")
    cfg = syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.1)
    cfg.device = "cuda:0"
    V = 80
    ds = SynthDataset(cfg, V, 3, 4, 160, 200, 8, dev)
    agent = BMHrlAgent(cfg, ds).to(dev)
    wv = BMWorkerValueFunction(cfg).to(dev)
    return cfg, ds, SynthLoader(ds), agent, wv, LabelSmoothing(0.7, 1), BiasedKL(0.7, 1), dev


def test_warmstart_loop_trains(setup):
    from epoch_loops.captioning_bmrl_loops import warmstart_bmhrl_bl, bmhrl_validation_next_word_loop, bimodal_decoder
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    opt = torch.optim.Adam(agent.parameters(), lr=3e-4)
    models = {"captioning": (agent, opt, ls), "worker": (wv, None, None), "manager": (None, None, None)}
    before = bmhrl_validation_next_word_loop(cfg, agent, loader, bimodal_decoder, ls, 0, None, "t")
    losses = [warmstart_bmhrl_bl(cfg, models, None, loader, e, "t", None, "extra-arg-of-the-driver") for e in range(4)]
    after = bmhrl_validation_next_word_loop(cfg, agent, loader, bimodal_decoder, ls, 0, None, "t")
    assert all(torch.isfinite(torch.tensor(losses)))
    assert after < before, (before, after)             # the step actually learns the synthetic batches
    toks = bimodal_decoder(agent.eval(), ds.batches[0]["feature_stacks"], 6, 2, 3, 1, "audio_video")
    assert toks.shape[0] == 4 and toks.shape[1] <= 7 and int(toks[0, 0]) == 2


def test_worker_rl_step_runs_with_synthetic_rewards(setup):
    from epoch_loops.captioning_bmrl_loops import train_bmhrl_bl
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    cfg.rl_stabilize = True
    cfg.grad_clip = 1.0
    cfg.rl_reward_fn = lambda sampled, captions: torch.rand(sampled.shape, device=sampled.device)
    opt = torch.optim.Adam(agent.parameters(), lr=1e-4)
    wopt = torch.optim.Adam(wv.parameters(), lr=1e-4)
    models = {"captioning": (agent, opt, bkl), "worker": (wv, wopt, torch.nn.MSELoss(reduction="none")),
              "manager": (None, None, None)}
    w0 = wv.projection.weight.detach().clone()
    m0 = agent.manager.linear.weight.detach().clone()
    p0 = agent.worker.core.projection.weight.detach().clone()
    loss = train_bmhrl_bl(cfg, models, None, loader, 0, "t", None, True)
    assert torch.isfinite(torch.tensor(loss))
    assert not torch.equal(w0, wv.projection.weight)                    # value head updated
    assert torch.equal(m0, agent.manager.linear.weight)                 # manager side frozen in the worker phase
    assert not torch.equal(p0, agent.worker.core.projection.weight)
    with pytest.raises(NotImplementedError):
        train_bmhrl_bl(cfg, models, None, loader, 0, "t", None, False)   # the reference's manager branch raises too


def test_trainer_graph_replay_matches_eager(setup):
    """Whole-step HIP graph == eager step (same seeds are not shared: compare with dropout off)."""
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    c2 = syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0)
    b = ds.batches[0]
    t1 = CaptionTrainer(c2, 80, dev, exploration=False, lr=1e-3)
    t2 = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3)
    t1.agent.train(); t2.agent.train()
    cap = b["caption_data"].caption
    l_eager = [float(t1.step(b["feature_stacks"], cap)) for _ in range(4)]
    t2.capture(b["feature_stacks"], cap, warmup=1)        # one eager (un-timed) step, then the captured graph
    l_graph = [float(t2.replay()) for _ in range(3)]
    assert all(abs(a - g) < 2e-3 * abs(a) for a, g in zip(l_eager[1:], l_graph)), (l_eager, l_graph)
    assert l_eager[3] < l_eager[0]
    if t2.opt.direct_grads and t2.opt.fused_shadows:          # (the defaults; both have environment switches for A/B runs)
        # one rank: the captured Adam pass reads the gradients in place (no gather copy inside the graph) ...
        plan_ptrs = t2.opt.__dict__["_seg_plan"][2]
        assert plan_ptrs is not None and sum(p != 0 for p in plan_ptrs) > 100
        # ... and the flat gradient bucket is not written at all
        assert float(t2.opt.grad.abs().max()) == 0.0


def test_unzeroed_gradient_arena_is_only_ever_overwritten(setup):
    """Weight-gradient GEMMs that run without a K split get their output from an arena that is never zeroed
    (StepScratch.f32(zero=False), decided by the launcher's own bmhrl_gemm_splits).  Poisoned with NaN before every step,
    a captured trainer must produce the losses and weights of the unpoisoned one -- and that arena must actually be in use."""
    from bmhrl_amd import functional as F, ops
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[0]
    cap = b["caption_data"].caption
    assert ops.gemm_overwrites(1024, 1024, 480) and not ops.gemm_overwrites(128, 300, 480)
    out = []
    try:
        for poison in (False, True):
            F._ARENA_POISON = poison
            t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3)
            t.agent.train()
            losses = [float(t.step(b["feature_stacks"], cap)) for _ in range(3)]
            assert t.scratch.need_raw > 100000 and t.scratch.raw is not None
            out.append((losses, t.opt.flat.clone()))
    finally:
        F._ARENA_POISON = False
    (l0, p0), (l1, p1) = out
    assert all(math.isfinite(x) for x in l1) and bool(torch.isfinite(p1).all())
    assert all(abs(a - c) < 2e-3 * abs(a) for a, c in zip(l0, l1)), (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) < 3e-3


def test_fused_head_loss_trains_like_the_four_kernel_tail(setup):
    """The warmstart step with the worker head's one-launch loss tail (functional.request_head_loss -> ops.head_loss) against
    the same step with log_softmax + smooth_kl_fwd + token_loss_reduce + smooth_kl_bwd: the fused form must actually run
    (once per step, TokenLossFn launching nothing), and losses / weights must agree (the gradient is bit-identical, the
    loss differs by the order of a row's fp32 sum; the comparison bound is that of two runs with atomics)."""
    from bmhrl_amd import functional as F, ops
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[0]
    cap = b["caption_data"].caption
    calls = {"fused": 0, "bwd": 0}
    real_fused, real_bwd = ops.head_loss, ops.smooth_kl_bwd

    def count_fused(*a, **k):
        calls["fused"] += 1
        return real_fused(*a, **k)

    def count_bwd(*a, **k):
        calls["bwd"] += 1
        return real_bwd(*a, **k)
    out = []
    old = F.FUSED_HEAD_LOSS
    ops.head_loss, ops.smooth_kl_bwd = count_fused, count_bwd
    try:
        for fused in (False, True):
            F.FUSED_HEAD_LOSS = fused
            calls["fused"] = calls["bwd"] = 0
            t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3)
            t.agent.train()
            losses = [float(t.step(b["feature_stacks"], cap)) for _ in range(3)]
            assert (calls["fused"], calls["bwd"]) == ((3, 0) if fused else (0, 3)), calls
            out.append((losses, t.opt.flat.clone()))
    finally:
        F.FUSED_HEAD_LOSS = old
        ops.head_loss, ops.smooth_kl_bwd = real_fused, real_bwd
    (l0, p0), (l1, p1) = out
    assert all(abs(a - c) < 2e-3 * abs(a) for a, c in zip(l0, l1)), (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) < 3e-3


def test_phased_adam_graph_equals_the_plain_graph(setup):
    """CaptionTrainer.phased_adam (one rank): backward in phases inside one graph, each bucket's Adam pass on a side stream as
    soon as its gradients are complete (FlatAdam.step_part) == the plain captured step."""
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[0]
    cap = b["caption_data"].caption
    out = []
    for phased in (False, True):
        t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3)
        t.agent.train()
        t.phased_adam = phased
        t.capture(b["feature_stacks"], cap, warmup=1)
        assert t._split() == phased and t.graph_b is None
        out.append(([float(t.replay()) for _ in range(3)], t.opt.flat.clone(), int(t.opt.step_dev)))
    (l0, p0, n0), (l1, p1, n1) = out
    assert n0 == n1 == 4                                             # one warm-up step + three replays, counted once per step
    assert all(abs(a - c) < 2e-3 * abs(a) for a, c in zip(l0, l1)), (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) < 3e-3


def test_trainer_rl_graph_replay_matches_eager(setup):
    """The worker RL step (sampled tokens + synthetic rewards + value head) captured as one HIP graph == the eager step:
    the sampler adds the device seed word, so replay k draws the same tokens as eager step k."""
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[0]
    cap = b["caption_data"].caption
    rew = syn.synthetic_rewards(cap.shape[0], cap.shape[1] - 1, seed=5).to(dev)
    fn = lambda sampled, captions: rew
    mk = lambda: CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3, phase="worker",
                                reward_fn=fn, value_lr=1e-3)
    t1 = mk()
    t1.agent.train(); t1.value_net.train()
    m0 = t1.agent.manager.linear.weight.detach().clone()
    v0 = t1.vopt.flat.clone()
    l_eager = [float(t1.step(b["feature_stacks"], cap)) for _ in range(4)]
    assert torch.equal(m0, t1.agent.manager.linear.weight)          # manager side frozen in the worker phase
    assert not torch.equal(v0, t1.vopt.flat)                        # value head trained by the same backward
    t2 = mk()
    t2.agent.train(); t2.value_net.train()
    t2.capture(b["feature_stacks"], cap, warmup=1)
    assert t2.graph_b is None                                       # one graph: forward, sampling, backward, both Adams
    l_graph = [float(t2.replay()) for _ in range(3)]
    assert all(torch.isfinite(torch.tensor(l_eager)))
    assert all(abs(a - g) < 2e-3 * abs(a) for a, g in zip(l_eager[1:], l_graph)), (l_eager, l_graph)
    assert float((t1.vopt.flat - t2.vopt.flat).norm() / t1.vopt.flat.norm()) < 1e-3


def test_validation_1by1_loop_decodes_the_loader(setup, tmp_path):
    """the one-by-one validation pass (reference epoch_loops/validation_loops.py:13-137) through the incremental decoder:
    one sentence per clip in the ActivityNet-captions result dictionary, same tokens as the prefix re-run decoder"""
    import json
    from epoch_loops.validation_loops import validation_1by1_loop
    from epoch_loops.captioning_bmrl_loops import bmhrl_greedy_decoder
    from bmhrl_amd.decode import greedy_decode
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    cfg.max_len, cfg.modality, cfg.log_path = 6, "audio_video", str(tmp_path)
    cfg.reference_paths, cfg.max_prop_per_vid = ["a", "b", "c", "d"], 100
    ds.phase = "val_1"
    ds.train_vocab.itos[3] = "</s>"
    wrapped = SimpleNamespace(module=agent, eval=agent.eval)
    agent.set_inference_mode(True)
    out = validation_1by1_loop(cfg, wrapped, loader, bmhrl_greedy_decoder, 0, None)
    saved = json.load(open(out["submission_path"]))["results"]
    assert sorted(saved) == [f"v{j}" for j in range(4)] and all(len(v) == 3 for v in saved.values())
    toks = greedy_decode(agent, ds.batches[0]["feature_stacks"], 6, 2, 3, 1, "audio_video", incremental=False).cpu()
    words = [ds.train_vocab.itos[int(i)] for i in toks[0]][1:]
    words = words[:words.index("</s>")] if "</s>" in words else words
    assert saved["v0"][0]["sentence"] == " ".join(words).capitalize()
    ds.phase = "train"


def test_adam_pass_keeps_weight_shadows_current(setup):
    """bmhrl_adam_segments writes the bf16 shadow of every weight it updates: after a step each cached shadow equals the
    bf16 cast of its parameter without a refresh pass, and the losses equal those of the plain Adam kernel + refresh"""
    from bmhrl_amd.functional import SHADOWS
    from bmhrl_amd.train import CaptionTrainer, FlatAdam
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[1]
    cap = b["caption_data"].caption
    losses = {}
    for fused in (True, False):
        FlatAdam.fused_shadows = fused
        try:
            t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-3)
            t.agent.train()
            losses[fused] = [float(t.step(b["feature_stacks"], cap)) for _ in range(3)]
            if fused:
                mine = {id(p) for p in t.opt.params}
                n_checked = 0
                n_split = []
                for store, is_w in ((SHADOWS.w, True), (SHADOWS.b, False)):
                    for key, (ver, buf, refs) in store.items():
                        params = [r() for r in refs]
                        if any(p is None or id(p) not in mine for p in params):
                            continue
                        assert SHADOWS._is_current(1 if is_w else 0, key, (ver, buf, refs))      # held current ...
                        off = 0
                        for p in params:                                       # ... and really is
                            if is_w:
                                hi = p.detach().to(torch.bfloat16)
                                assert torch.equal(buf[off:off + p.shape[0], :p.shape[1]], hi)
                                part = SHADOWS.split.get(key, 0)
                                if part:          # [hi | lo | hi] shadow of the vocabulary projection (WorkerHeadFn)
                                    n_split.append(key)
                                    lo = (p.detach() - hi.float()).to(torch.bfloat16)
                                    assert torch.equal(buf[:, part:part + p.shape[1]], lo)
                                    assert torch.equal(buf[:, 2 * part:2 * part + p.shape[1]], hi)
                                    assert float(buf[:, p.shape[1]:part].abs().max()) == 0.0      # padding stays zero
                                off += p.shape[0]
                            else:
                                assert torch.equal(buf[off:off + p.numel()], p.detach().reshape(-1))
                                off += p.numel()
                            n_checked += 1
                assert n_checked > 50 and len(n_split) == 1
        finally:
            FlatAdam.fused_shadows = True
    assert all(abs(a - c) <= 2e-3 * abs(c) for a, c in zip(losses[True], losses[False])), losses


def test_shadows_made_after_capture_follow_the_replayed_updates(setup):
    """A captured step updates the weights without any host code running, and autograd's version counters do not see the
    optimizer kernel: shadow entries the captured Adam pass does not maintain itself (here: the per-module groups the decoders
    use, created AFTER the capture) must still be re-made after replays -- FlatAdam.generation carries that."""
    from bmhrl_amd.decode import greedy_decode
    from bmhrl_amd.functional import SHADOWS
    from bmhrl_amd.train import CaptionTrainer
    cfg, ds, loader, agent, wv, ls, bkl, dev = setup
    b = ds.batches[0]
    cap = b["caption_data"].caption
    t = CaptionTrainer(syn.tiny_cfg(d_model=1024, rl_att_heads=4, dout_p=0.0), 80, dev, exploration=False, lr=1e-2)
    t.agent.train()
    t.capture(b["feature_stacks"], cap, warmup=1)
    att = t.agent.bm_worker_fus.decoder.layers[0].self_att
    w = att.linear_Q2d.weight
    t.agent.eval()
    greedy_decode(t.agent, b["feature_stacks"], 4, 2, 3, 1, "audio_video")        # creates the decoders' shadow groups
    before = w.detach().clone()
    s0 = SHADOWS.weight(w, att.linear_K2d.weight, att.linear_V2d.weight).clone()
    t.agent.train()
    for _ in range(3):
        t.replay()
    assert float((w.detach() - before).abs().max()) > 1e-3                        # the replays really moved the weight
    s1 = SHADOWS.weight(w, att.linear_K2d.weight, att.linear_V2d.weight)
    D = w.shape[0]
    assert torch.equal(s1[:D, :w.shape[1]], w.detach().to(torch.bfloat16)) and not torch.equal(s0, s1)
    # and decoding after the training steps equals a decoder built from scratch on the same weights
    t.agent.eval()
    toks = greedy_decode(t.agent, b["feature_stacks"], 4, 2, 3, 1, "audio_video")
    t.agent.__dict__.pop("_incremental_decoders", None)
    SHADOWS.invalidate()
    assert torch.equal(toks, greedy_decode(t.agent, b["feature_stacks"], 4, 2, 3, 1, "audio_video"))
