"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (``/root/reference`` is mounted there, never on the GPU box):

    python tests/golden/make_golden.py

It imports the reference's importable, pure-torch files (model/*.py, loss/*.py; SURVEY.md
section 8c), loads deterministic weights from ``bmhrl_amd.synthetic`` into them and stores
inputs' seeds + expected outputs as small ``.npz`` files.  The fixtures are data only: no text
of the reference is stored.  ``tests/test_oracle_golden.py`` pins ``oracle/bmhrl_oracle.py`` to
them; the ``-m gpu`` tests pin the HIP path to the same files.
"""
import contextlib
import io
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("BMHRL_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from bmhrl_amd import synthetic as syn  # noqa: E402

quiet = contextlib.redirect_stdout(io.StringIO())


def np_(t):
    return t.detach().cpu().numpy()


def build_ref_agent(cfg, voc_size, seed):
    from model.bm_hrl_agent import BMHrlAgent
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "critic.cp")
        torch.save(syn.synthetic_critic_state(cfg.d_model_caps, seed=1), path)
        cfg.rl_critic_path = path
        ds = SimpleNamespace(trg_voc_size=voc_size, train_vocab=SimpleNamespace(vectors=None))
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            agent = BMHrlAgent(cfg, ds)
    shapes = {k: tuple(v.shape) for k, v in agent.state_dict().items()}
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=seed)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    agent.load_state_dict(sd)
    agent.eval()
    agent.set_inference_mode(True)
    return agent, shapes


def kat():
    """Known-answer cases, re-derived from the reference here."""
    from model.blocks import PositionalEncoder
    from model.multihead_attention import attention
    from model.bm_hrl_agent import Manager
    from loss.label_smoothing import LabelSmoothing
    from loss.biased_kl import BiasedKL, Reinforce
    out = {}
    out["pe8"] = PositionalEncoder(8, 0.0, seq_len=5).pos_enc_mat[0].numpy()
    out["pe20"] = PositionalEncoder(20, 0.0, seq_len=40).pos_enc_mat[0].numpy()
    Q = torch.ones(1, 1, 2, 4)
    K = torch.arange(12.).reshape(1, 1, 3, 4) / 10
    V = torch.arange(12.).reshape(1, 1, 3, 4)
    out["att_allmasked"] = np_(attention(Q, K, V, torch.tensor([[[[False, False, False]]]])))
    out["att_midmasked"] = np_(attention(Q, K, V, torch.tensor([[[[True, False, True]]]])))
    mgr = Manager("cpu", 4, 1, 0.0)
    B, L = 5, 6
    g = (1 + torch.arange(B * L, dtype=torch.float32)).reshape(B, L, 1)
    seg = torch.zeros(B, L, dtype=torch.int32)
    for b, l in [(1, 1), (1, 3), (3, 5), (4, 2)]:
        seg[b, l] = 1
    out["expand_in"] = np_(g)
    out["expand_seg"] = np_(seg)
    out["expand_out"] = np_(mgr.expand_goals(g.clone(), seg))
    B, S, V_ = 2, 3, 6
    idx = torch.arange(B * S * V_, dtype=torch.float32).reshape(B, S, V_)
    lp = torch.log_softmax((idx % 7) / 3, dim=-1)
    out["a4_lp"] = np_(lp)
    with contextlib.redirect_stderr(io.StringIO()):
        out["a4_ls"] = np_(LabelSmoothing(0.7, 1)(lp, torch.tensor([[2, 4, 1], [5, 3, 2]])))
        out["a4_ls_guard"] = np_(LabelSmoothing(0.7, 1)(lp, torch.tensor([[1, 4, 2], [5, 3, 2]])))
        out["a4_bkl"] = np_(BiasedKL(0.7, 1)(lp, torch.tensor([[2, 4, 1], [5, 3, 2]]), torch.tensor([[2, 0, 3], [1, 3, 4]]),
                                              torch.tensor([[.5, .25, 1.], [.8, 0., .1]])))
        out["a4_reinforce"] = np_(Reinforce()(torch.exp(lp), torch.tensor([[2, 0, 3], [1, 3, 4]]),
                                              torch.tensor([[.1, .2, .3], [.4, .5, .6]]), torch.tensor([[.3, .1, .0], [.2, .2, .9]])))
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)


def losses_random():
    from loss.label_smoothing import LabelSmoothing
    from loss.biased_kl import BiasedKL, Reinforce
    g = torch.Generator().manual_seed(11)
    B, S, V = 3, 7, 37
    logits = torch.randn(B, S, V, generator=g)
    trg = torch.randint(2, V, (B, S), generator=g)
    trg[0, 5:] = 1
    trg[2, 3:] = 1
    sampled = torch.randint(0, V, (B, S), generator=g)
    score = torch.rand(B, S, generator=g)
    baseline = torch.rand(B, S, generator=g) * 0.5
    out = dict(logits=np_(logits), trg=np_(trg), sampled=np_(sampled), score=np_(score), baseline=np_(baseline))
    with contextlib.redirect_stderr(io.StringIO()):
        x = logits.clone().requires_grad_(True)
        lp = torch.log_softmax(x, -1)
        ls = LabelSmoothing(0.7, 1)(lp, trg)
        n_tok = (trg != 1).sum()
        (ls.sum() / n_tok).backward()
        out["ls"] = np_(ls)
        out["ls_grad_logits"] = np_(x.grad)
        for stab in (False, True):
            x = logits.clone().requires_grad_(True)
            lp = torch.log_softmax(x, -1)
            mask = trg != 1
            p = torch.gather(torch.exp(lp), 2, sampled.unsqueeze(-1)).squeeze(-1)
            sc = (score - baseline) * mask.float() if stab else score
            amp = torch.clamp(sc * p * mask.sum(-1).reshape(-1, 1).float(), 0, 1)
            div = BiasedKL(0.7, 1)(lp, trg, sampled, amp)
            (div.sum() / (n_tok * 0.2)).backward()
            tag = "stab" if stab else "raw"
            out[f"bkl_{tag}"] = np_(div)
            out[f"bkl_{tag}_amp"] = np_(amp)
            out[f"bkl_{tag}_grad_logits"] = np_(x.grad)
        x = logits.clone().requires_grad_(True)
        probs = torch.softmax(x, -1)
        r = Reinforce()(probs, sampled, score, baseline)
        r.backward()
        out["reinforce"] = np_(r)
        out["reinforce_grad_logits"] = np_(x.grad)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


def biased_kl_forward_cases():
    """BiasedKL.forward (loss/biased_kl.py:22-53) as its callers may use it: the amplitude `biased_offset` is an ordinary
    tensor argument -- attached to the prediction (the loops' clamp(score * p(a) * n, 0, 1), :285,321-322), detached, or
    attached through some other function of the prediction.  Divergence and the gradient w.r.t. the logits for all three."""
    from loss.biased_kl import BiasedKL
    g = torch.Generator().manual_seed(23)
    B, S, V = 4, 6, 29
    logits = torch.randn(B, S, V, generator=g)
    trg = torch.randint(2, V, (B, S), generator=g)
    trg[1, 4:] = 1
    trg[3, 2:] = 1
    sampled = torch.randint(0, V, (B, S), generator=g)
    score = torch.rand(B, S, generator=g)
    up = torch.randn(B * S, generator=g)             # an arbitrary upstream gradient of the divergence's row sums
    out = dict(logits=np_(logits), trg=np_(trg), sampled=np_(sampled), score=np_(score), up=np_(up))
    with contextlib.redirect_stderr(io.StringIO()):
        for tag in ("attached", "detached", "other"):
            x = logits.clone().requires_grad_(True)
            lp = torch.log_softmax(x, -1)
            p = torch.gather(torch.exp(lp), 2, sampled.unsqueeze(-1)).squeeze(-1)
            n = (trg != 1).sum(-1).reshape(-1, 1).float()
            if tag == "other":
                amp = torch.clamp(score * torch.sqrt(p) * 0.9 + 0.05, 0, 1)
            else:
                amp = torch.clamp(score * p * n, 0, 1)
            if tag == "detached":
                amp = amp.detach()
            div = BiasedKL(0.7, 1)(lp, trg, sampled, amp)
            (div.sum(-1) * up).sum().backward()
            out[f"{tag}_amp"] = np_(amp)
            out[f"{tag}_rows"] = np_(div.sum(-1))
            out[f"{tag}_grad_logits"] = np_(x.grad)
    np.savez_compressed(os.path.join(HERE, "biased_kl_forward.npz"), **out)


def mha_cases():
    from model.multihead_attention import MultiheadedAttention
    out = {}
    g = torch.Generator().manual_seed(5)
    # (dQ, dK, H, d_model, Sq, Sk, tag)
    for dq, dk, H, D, Sq, Sk, tag in [(48, 48, 4, 64, 7, 7, "self"), (48, 24, 4, 64, 7, 9, "cross"), (8, 20, 2, 64, 6, 6, "goal")]:
        with quiet:
            m = MultiheadedAttention(dq, dk, dk, H, 0.0, D)
        sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
        m.load_state_dict(sd)
        B = 3
        q = torch.randn(B, Sq, dq, generator=g)
        kv = q if tag == "self" else torch.randn(B, Sk, dk, generator=g)
        if tag == "goal":
            mask = torch.tril(torch.ones(B, Sq, Sk, dtype=torch.bool))
            mask[1, :, 4:] = False
        else:
            mask = torch.ones(B, 1, Sk, dtype=torch.bool)
            mask[0, 0, Sk - 2:] = False
            mask[2, 0, :] = False  # a fully masked sample -> uniform attention
        y = m(q, kv, kv, mask)
        out[f"{tag}_q"], out[f"{tag}_kv"], out[f"{tag}_mask"], out[f"{tag}_out"] = np_(q), np_(kv), np_(mask), np_(y)
    np.savez_compressed(os.path.join(HERE, "mha.npz"), **out)


def agent_tiny():
    cfg = syn.tiny_cfg()
    Vsz = 50
    agent, shapes = build_ref_agent(cfg, Vsz, seed=0)
    from model.masking import make_masks
    from loss.label_smoothing import LabelSmoothing
    from loss.biased_kl import BiasedKL
    B, Tv, Ta, L = 4, 7, 9, 6
    batch = syn.synthetic_batch(B, Tv, Ta, L, Vsz, seed=7, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=3)
    cap = batch["captions"]
    trg_in, trg_y = cap[:, :-1], cap[:, 1:]
    fs = {"rgb": batch["rgb"], "flow": batch["flow"], "audio": batch["audio"]}
    masks = make_masks(fs, trg_in, "audio_video", 1)
    x = (fs["rgb"] + fs["flow"], fs["audio"])
    out = {"state_keys": np.array(sorted(shapes)), "state_shapes": np.array([str(shapes[k]) for k in sorted(shapes)])}
    with quiet, contextlib.redirect_stderr(io.StringIO()):
        pred, wf, mf, goals, seg = agent(x, trg_in, masks)
        out.update(pred=np_(pred), worker_feat=np_(wf), manager_feat=np_(mf), goals=np_(goals), seg=np_(seg))
        # encoder outputs, for layer-level checks
        V = agent.pos_enc_V(x[0]); A = agent.pos_enc_A(x[1])
        Va, Av = agent.bm_enc((V, A), masks)
        out.update(enc_v=np_(Va), enc_a=np_(Av))
        # warmstart step loss + all grads
        agent.zero_grad()
        n_tok = (trg_y != 1).sum()
        loss = LabelSmoothing(0.7, 1)(pred, trg_y).sum() / n_tok
        loss.backward()
        out["ws_loss"] = np_(loss)
        for k, p in agent.named_parameters():
            if p.grad is not None:
                out["ws_grad/" + k] = np_(p.grad)
        # mixed prediction (trg tuple + factor)
        yhat = torch.roll(trg_in, 1, dims=0)
        pm = agent(x, (trg_in, yhat), masks, 0.25)[0]
        out["yhat"] = np_(yhat)
        out["pred_mixed"] = np_(pm)
        # worker RL step with given sample / reward / baseline (worker phase: teach_worker)
        agent.teach_worker()
        agent.zero_grad()
        pred, wf, mf, goals, seg = agent(x, trg_in, masks)
        g = torch.Generator().manual_seed(3)
        sampled = torch.distributions.Categorical(torch.exp(pred)).sample() if False else torch.multinomial(
            torch.exp(pred).reshape(-1, Vsz), 1, generator=g).reshape(B, L)
        score = syn.synthetic_rewards(B, L, seed=2)
        mask = trg_y != 1
        p = torch.gather(torch.exp(pred), 2, sampled.unsqueeze(-1)).squeeze(-1)
        amp = torch.clamp(score * p * mask.sum(-1).reshape(-1, 1).float(), 0, 1)
        div = BiasedKL(0.7, 1)(pred, trg_y, sampled, amp)
        rl = div.sum() / (n_tok * 0.2)
        rl.backward()
        out.update(rl_sampled=np_(sampled), rl_score=np_(score), rl_loss=np_(rl))
        for k, p_ in agent.named_parameters():
            if p_.grad is not None:
                out["rl_grad/" + k] = np_(p_.grad)
    np.savez_compressed(os.path.join(HERE, "agent_tiny.npz"), **out)


def critic_case():
    from model.bm_hrl_agent import SegmentCritic
    cfg = syn.tiny_cfg()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "critic.cp")
        torch.save(syn.synthetic_critic_state(cfg.d_model_caps, seed=1), path)
        cfg.rl_critic_path = path
        with quiet:
            c = SegmentCritic(cfg)
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(3, 6, cfg.d_model_caps, generator=g) * 2
    np.savez_compressed(os.path.join(HERE, "critic.npz"), emb=np_(emb), out=np_(c(emb)))


def sample_clip_decode():
    """BASELINE config 1: the reference's sample clip, greedy decode on the CPU reference path."""
    from captioning_datasets.load_features import crop_a_segment
    from model.masking import make_masks
    cfg = syn.default_cfg(dout_p=0.0, rl_critic_score_threshhold=1.0)
    Vsz = 2000  # the real vocabulary needs torchtext/spaCy (absent); V is a parameter
    agent, _ = build_ref_agent(cfg, Vsz, seed=0)
    feats = {}
    for k, f in (("rgb", "rgb"), ("flow", "flow"), ("audio", "vggish")):
        a = torch.from_numpy(np.load(os.path.join(REF, "sample", f"women_long_jump_{f}.npy"))).float()
        feats[k] = crop_a_segment(a, 0, 15, 16).unsqueeze(0)
    trg = torch.full((1, 1), 2, dtype=torch.long)
    first = None
    margins, top_logp = [], []           # per step: the reference's own top-1 / top-2 gap and the chosen token's log-prob
    with torch.no_grad(), quiet:
        while trg.shape[-1] <= 12:
            masks = make_masks(feats, trg, "audio_video", 1)
            pred = agent.inference((feats["rgb"] + feats["flow"], feats["audio"]), trg, masks)
            if first is None:
                first = pred[0, -1].clone()
            t2 = torch.topk(pred[0, -1], 2).values
            margins.append(float(t2[0] - t2[1]))
            top_logp.append(float(t2[0]))
            nxt = pred[:, -1].argmax(-1, keepdim=True)
            trg = torch.cat([trg, nxt], -1)
    top2 = torch.topk(first, 2).values
    np.savez_compressed(os.path.join(HERE, "sample_clip.npz"), rgb=np_(feats["rgb"]),
                        flow=np_(feats["flow"]), audio=np_(feats["audio"]),
                        tokens=np_(trg), first_logp=np_(first), first_margin=np_(top2[0] - top2[1]), voc=np.array(Vsz),
                        margins=np.array(margins, dtype=np.float32), top_logp=np.array(top_logp, dtype=np.float32))


def detr_cases():
    """Post-norm encoder / decoder stacks (model/encoder.py, model/decoder.py).  The reference's `causal=True` branch
    builds its triangular fill on `get_device()`, which is -1 on the CPU and raises; the cases here stay on the branches
    that run on the CPU reference: a tensor query position (add_pos=True), and add_pos=False without a query mask (the
    causal fill is skipped when mask is None, model/multihead_attention.py:17)."""
    from model.blocks import PositionalEncoder
    from model.decoder import TransformerDecoder, TransformerDecoderLayer
    from model.encoder import TransformerEncoder, TransformerEncoderLayer
    g = torch.Generator().manual_seed(21)
    B, S, L, D, dC, dG, H, dff = 3, 9, 6, 64, 24, 8, 4, 48
    out = {}

    def load(m):
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(syn.fill_state_dict(shapes, seed=9))
        return m.eval()

    with quiet:
        enc = load(TransformerEncoder(TransformerEncoderLayer(D, H, dff, 0.0, embed_size=20), 2, torch.nn.LayerNorm(D)))
        dec = load(TransformerDecoder(TransformerDecoderLayer(D, H, dC, dG, dff, 0.0), 2, torch.nn.LayerNorm(dC)))
    src = torch.randn(B, S, D, generator=g)
    mask = torch.ones(B, 1, S, dtype=torch.bool)
    mask[0, 0, S - 3:] = False
    mask[2, 0, :2] = False
    memory = enc(src, mask, PositionalEncoder(D, 0.0))
    out["src"], out["mask"], out["enc_out"] = np_(src), np_(mask), np_(memory)
    tgt = torch.randn(B, L, dC, generator=g)
    qpos = 0.3 * torch.randn(B, L, dC, generator=g)
    qmask = torch.tril(torch.ones(B, L, L, dtype=torch.bool))
    qmask[1, :, 4:] = False
    objs = torch.randn(B, 5, 256, generator=g)
    goal = torch.randn(B, L, dG, generator=g)
    mem = memory[-1].detach()
    a = dec(tgt, mem, mask, PositionalEncoder(D, 0.0), qpos, qmask, None, None, None, True, objs, None)
    b = dec(tgt, mem, mask, PositionalEncoder(D, 0.0), PositionalEncoder(dC, 0.0), None, goal, qmask, PositionalEncoder(dG, 0.0),
            False, None, None)
    out.update(tgt=np_(tgt), qpos=np_(qpos), qmask=np_(qmask), objs=np_(objs), goal=np_(goal), dec_a=np_(a), dec_b=np_(b))
    np.savez_compressed(os.path.join(HERE, "detr.npz"), **out)


def detr_agent_case():
    """DetrCaption (model/det_bmhrl_agent.py) at small widths: the state dict of the reference's own module and what its forward
    produces up to the point the CPU can take it -- the three Conv1d('same') + GroupNorm blocks, the 100-query object detector
    and the video encoder.  The caption decoder's causal self attention raises on the CPU (get_device() == -1,
    model/multihead_attention.py:19), so the log-probs are pinned through the oracle (tests/test_oracle_golden.py checks the
    oracle against everything stored here; tests/golden/detr.npz holds the decoder layer itself).  model/object_detector.py
    imports torchvision.models.VisionTransformer without using it: an empty module of that name lets the file import."""
    import types
    if "torchvision" not in sys.modules:
        tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
        tvm.VisionTransformer = object
        tv.models = tvm
        sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    from model.det_bmhrl_agent import DetrCaption
    V = 41
    cfg = syn.tiny_cfg(d_model=64, d_model_video=64, d_vid=64, d_model_caps=20, rl_att_heads=4, rl_goal_d=8, dout_p=0.0)
    cfg.pre_goal_attention = False
    cfg.device = "cpu"
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "critic.cp")
        torch.save(syn.synthetic_critic_state(cfg.d_model_caps, seed=1), path)
        cfg.rl_critic_path = path
        ds = SimpleNamespace(trg_voc_size=V, train_vocab=SimpleNamespace(vectors=None))
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            agent = DetrCaption(cfg, ds)
    shapes = {k: tuple(v.shape) for k, v in agent.state_dict().items()}
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=13)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    agent.load_state_dict(sd)
    agent.eval()
    g = torch.Generator().manual_seed(31)
    B, T = 2, 11
    xv = torch.randn(B, T, cfg.d_model, generator=g)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, T - 3:] = False
    # the weights themselves are not stored (27 M values): they are syn.fill_state_dict(shapes, seed=13) + the synthetic critic
    # (seed 1), a pure function of the names and shapes recorded here -- which also pin the module's state-dict layout
    keys = sorted(agent.state_dict().keys())
    out = {"keys": np.array(keys), "shapes": np.array([",".join(str(d) for d in shapes[k]) for k in keys])}
    out["x_video"], out["V_mask"] = np_(xv), np_(mask)
    with torch.no_grad():
        vf = xv.transpose(1, 2)
        for i in range(agent.n_time):
            vf = agent.input_proj[i](vf)
            out[f"proj{i}"] = np_(vf.transpose(1, 2))
        xp = vf.transpose(1, 2)
        cls, hs, ob_mask = agent.object_detector(xp, mask)
        out["obj_logits"], out["obj_hs"], out["obj_mask"] = np_(cls), np_(hs), np_(ob_mask)
        out["memory"] = np_(agent.encoder(xp, mask, agent.pos_enc))
    np.savez_compressed(os.path.join(HERE, "detr_agent.npz"), **out)


def rl_glue_cases():
    """metrics/util.py:discontinue_reward (the file imports only torch, so it is loaded by path; its package pulls nltk).
    The segment loops of metrics/batched_meteor.py and epoch_loops/captioning_bmrl_loops.py are run by rl_loops_cases()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metrics_util", os.path.join(REF, "metrics", "util.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(17)
    out = {}
    n = 0
    for i in range(24):
        B = int(torch.randint(1, 6, (1,), generator=g))
        L = int(torch.randint(2, 13, (1,), generator=g))
        x = torch.randn(B, L, generator=g)
        seg = (torch.rand(B, L, generator=g) < 0.35).int()
        if i % 5 == 0:
            seg[0] = 0
        gamma = 0.5 + 0.45 * float(torch.rand(1, generator=g))
        n_step = [100, 3, 1][i % 3]
        out[f"x{i}"], out[f"seg{i}"], out[f"par{i}"] = np_(x), np_(seg), np.array([gamma, n_step])
        out[f"plain{i}"] = np_(ref.discontinue_reward(x.clone(), gamma, n_step).float())
        out[f"segd{i}"] = np_(ref.discontinue_reward(x.clone(), gamma, n_step, seg))
        n += 1
    out["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "rl_glue.npz"), **out)


def import_reference_loops():
    """epoch_loops/captioning_bmrl_loops.py and metrics/batched_meteor.py import nltk at module level (absent in this
    image) although none of the functions run here touches it: empty modules of that name let the imports pass.  Nothing
    of nltk is restated -- every attribute the import lines ask for is None."""
    import importlib
    import types
    for name in ("nltk", "nltk.corpus", "nltk.translate", "nltk.translate.meteor_score", "nltk.tokenize", "nltk.tokenize.treebank"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nltk"].corpus = sys.modules["nltk.corpus"]
    sys.modules["nltk.corpus"].wordnet = None
    sys.modules["nltk.translate"].meteor = None
    sys.modules["nltk.translate.meteor_score"].meteor_score = None
    sys.modules["nltk.translate.meteor_score"].single_meteor_score = None
    sys.modules["nltk.tokenize.treebank"].TreebankWordDetokenizer = None
    return importlib.import_module("epoch_loops.captioning_bmrl_loops"), importlib.import_module("metrics.batched_meteor")


class GivenScore:
    """Scorer stand-in for get_score (captioning_bmrl_loops.py:219-230): type "CIDER", returns the (B, L) reward it was
    given -- BASELINE configs[2] stubs the rewards the same way."""
    type = "CIDER"

    def __init__(self, score):
        self.score = score

    def delta_cider_worker(self, tokens, caption):
        return self.score.clone(), None

    def delta_cider_manager(self, tokens, caption, mask, segments):
        return self.score.clone(), None


def rl_loops_cases():
    """The reference's OWN biased_kl (captioning_bmrl_loops.py:271-334), both branches and both `stabilize` values, its
    get_norm_reward_factor (:414-416) and metrics/batched_meteor.py:segment_reward (:19-36) on random cases incl. rows
    without a segment (row 0 among them), padded rows, adjacent segment ends and a batch with no segment at all."""
    from loss.biased_kl import BiasedKL
    loops, meteor = import_reference_loops()
    g = torch.Generator().manual_seed(29)
    out = {}
    n = 0
    for i in range(40):
        B = int(torch.randint(1, 6, (1,), generator=g))
        L = int(torch.randint(2, 10, (1,), generator=g))
        V = int(torch.randint(7, 30, (1,), generator=g))
        logits = 2.0 * torch.randn(B, L, V, generator=g)
        trg = torch.randint(2, V, (B, L), generator=g)
        for b in range(B):
            if float(torch.rand(1, generator=g)) < 0.5:
                trg[b, int(torch.randint(1, L, (1,), generator=g)):] = 1
        seg = (torch.rand(B, L, generator=g) < 0.4).int()
        if i % 4 == 0:
            seg[0] = 0
        if i % 7 == 3 and B > 2:
            seg[1] = 0
        if i == 13:
            seg[:] = 0
        if i % 5 == 2:
            seg[-1, :2] = 1
        score = torch.rand(B, L, generator=g)
        baseline = torch.rand(B, L, generator=g) * 0.6
        mask = trg != 1
        stab = bool(i % 2)
        out.update({f"logits{i}": np_(logits), f"trg{i}": np_(trg), f"seg{i}": np_(seg), f"score{i}": np_(score),
                    f"base{i}": np_(baseline), f"stab{i}": np.array(stab)})
        crit = BiasedKL(0.7, 1)
        with contextlib.redirect_stderr(io.StringIO()):
            # manager branch
            x = logits.clone().requires_grad_(True)
            div, sc, tok, _ = loops.biased_kl(False, torch.log_softmax(x, -1), GivenScore(score), baseline.clone(), trg, None, mask, seg,
                                              "cpu", crit, stab)
            div.sum().backward()
            out.update({f"m_div{i}": np_(div), f"m_score{i}": np_(sc[0]), f"m_tok{i}": np_(tok[0]), f"m_grad{i}": np_(x.grad)})
            # worker branch: the sample is drawn from torch's global generator (Categorical.sample)
            x = logits.clone().requires_grad_(True)
            torch.manual_seed(1000 + i)
            div, sc, tok, _ = loops.biased_kl(True, torch.log_softmax(x, -1), GivenScore(score), baseline.clone(), trg, None, mask, seg,
                                              "cpu", crit, stab)
            div.sum().backward()
            out.update({f"w_div{i}": np_(div), f"w_score{i}": np_(sc[0]), f"w_tok{i}": np_(tok[0]), f"w_grad{i}": np_(x.grad)})
        out[f"nf_w{i}"] = np_(loops.get_norm_reward_factor(True, mask, seg))
        out[f"nf_m{i}"] = np_(loops.get_norm_reward_factor(False, mask, seg))
        sr, sidx = meteor.segment_reward(score, seg)
        out[f"sr{i}"], out[f"sr_idx{i}"] = np_(sr), np_(sidx)
        n += 1
    out["n"] = np.array(n)
    np.savez_compressed(os.path.join(HERE, "rl_loops.npz"), **out)


def loader_clip_arrays(i):
    """deterministic, compressible feature files of test clip i: (rgb, flow, audio) or None for a missing modality"""
    S = [7, 12, 1, 9, 5, 10][i % 6]
    Sa = [11, 4, 2, 15, 6, 3][i % 6]
    r = (np.arange(S)[:, None] * 3 + (np.arange(1024)[None, :] % 5) + i).astype(np.float32)
    f = (np.arange(S)[:, None] * 2 - (np.arange(1024)[None, :] % 3) + 0.5 * i).astype(np.float32)
    a = (np.arange(Sa)[:, None] + (np.arange(128)[None, :] % 4) * 0.25 + i).astype(np.float32)
    return r, f, a


LOADER_CLIPS = [  # (index, start, end, duration, video files present, audio file present)
    (0, 0.0, 10.0, 10.0, True, True), (1, 2.5, 7.0, 20.0, True, True), (2, 0.0, 0.3, 9.0, True, True),
    (3, 8.9, 9.0, 9.0, True, True), (4, 1.0, 4.0, 5.0, False, True), (5, 3.0, 6.0, 12.0, True, False),
    (1, 19.99, 20.0, 20.0, True, True), (3, 0.0, 0.01, 9.0, True, True)]


def loader_cases():
    """captioning_datasets/load_features.py (imports numpy / torch only): crop_a_segment over a grid incl. its edge cases,
    and load_features_from_npy on files written here; the batch assembly (captioning_dataset.py:262-290: zero row for a
    missing clip, pad_sequence with pad_idx / 0 / pad_idx) follows the reference line by line with torch's pad_sequence
    -- that module itself imports torchtext and cannot be loaded."""
    import importlib.util
    from torch.nn.utils.rnn import pad_sequence
    spec = importlib.util.spec_from_file_location("ref_load_features", os.path.join(REF, "captioning_datasets", "load_features.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(23)
    cases = []
    for i in range(400):
        S = int(torch.randint(1, 40, (1,), generator=g))
        dur = float(torch.rand(1, generator=g)) * 200 + 1
        a, b = sorted(float(x) for x in torch.rand(2, generator=g) * dur)
        if i % 7 == 0:
            b = a                                    # empty segment
        if i % 11 == 0:
            a, b = dur, dur                          # at the very end
        if i % 13 == 0:
            b = dur * 1.2                            # annotation past the end of the video
        if i % 17 == 0:
            a, b = dur * 1.1, dur * 1.3              # entirely past the end
        feat = torch.arange(S, dtype=torch.float32)[:, None].expand(S, 2)
        out = ref.crop_a_segment(feat, a, b, dur)
        cases.append([S, a, b, dur, -1 if out is None else int(out[0, 0]), 0 if out is None else out.shape[0]])
    res = {"crop": np.array(cases, dtype=np.float64)}
    pad_idx = 1
    with tempfile.TemporaryDirectory() as td:
        cfg = SimpleNamespace(video_features_path=td, audio_features_path=td)
        for i in range(6):
            r, f, a = loader_clip_arrays(i)
            present = [c for c in LOADER_CLIPS if c[0] == i]
            if all(c[4] for c in present):
                np.save(os.path.join(td, f"clip{i}_rgb.npy"), r)
                np.save(os.path.join(td, f"clip{i}_flow.npy"), f)
            if all(c[5] for c in present):
                np.save(os.path.join(td, f"clip{i}.npy"), a)
        rgbs, flows, auds = [], [], []
        for i, start, end, dur, _, _ in LOADER_CLIPS:
            st = ref.load_features_from_npy(cfg, ["i3d_features", "vggish_features"], f"clip{i}", start, end, dur, pad_idx)
            r, f, a = st["rgb"], st["flow"], st["audio"]
            if r is None and f is None:
                r, f = ref.fill_missing_features("zero", 1024), ref.fill_missing_features("zero", 1024)
            if a is None:
                a = ref.fill_missing_features("zero", 128)
            rgbs.append(r); flows.append(f); auds.append(a)
        cfg.pad_feats_up_to = {"video": 14, "audio": 16}
        full = ref.load_features_from_npy(cfg, ["i3d_features", "vggish_features"], "clip1", 0, 1, 1, pad_idx, get_full_feat=True)
        res["full_rgb"], res["full_flow"], res["full_audio"] = np_(full["rgb"]), np_(full["flow"]), np_(full["audio"])
        res["full_len"] = np.array([full["orig_feat_length"][k] for k in ("rgb", "flow", "audio")])
        res["rgb"] = np_(pad_sequence(rgbs, batch_first=True, padding_value=pad_idx))
        res["flow"] = np_(pad_sequence(flows, batch_first=True, padding_value=0))
        res["audio"] = np_(pad_sequence(auds, batch_first=True, padding_value=pad_idx))
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **res)


def value_function_cases():
    """BMWorkerValueFunction / BMManagerValueFunction (reference model/bm_hrl_agent.py:251-286), eval mode: outputs at the
    default widths (d_model_caps 300 -> 600 -> 300 -> 1) and, at width 48 (small fixture), the masked-MSE loss against a
    given score (epoch_loops/captioning_bmrl_loops.py:873-876) with every parameter gradient."""
    from model.bm_hrl_agent import BMManagerValueFunction, BMWorkerValueFunction
    out = {}
    for d, with_grads in ((300, False), (48, True)):
        cfg = syn.tiny_cfg()
        cfg.d_model_caps, cfg.rl_goal_d, cfg.dout_p = d, 64, 0.1
        g = torch.Generator().manual_seed(21 + d)
        B, L = 3, 6
        feat = torch.randn(B, L, d, generator=g)
        goal = torch.randn(B, L, 64, generator=g)
        score = torch.rand(B, L, generator=g)
        mask = (torch.rand(B, L, generator=g) < 0.8).float()
        out.update({f"d{d}/feat": np_(feat), f"d{d}/goal": np_(goal), f"d{d}/score": np_(score), f"d{d}/mask": np_(mask)})
        for name, cls in (("worker", BMWorkerValueFunction), ("manager", BMManagerValueFunction)):
            m = cls(cfg)
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict(syn.fill_state_dict(shapes, seed=31 if name == "worker" else 32))
            m.eval()
            y = m((feat, goal)) if name == "worker" else m(feat)
            out[f"d{d}/{name}/out"] = np_(y)
            out[f"d{d}/{name}/keys"] = np.array(sorted(shapes))
            if with_grads:
                loss = (torch.nn.MSELoss(reduction="none")(y.squeeze(-1), score) * mask).mean()
                loss.backward()
                out[f"d{d}/{name}/loss"] = np_(loss)
                for k, v in m.named_parameters():
                    out[f"d{d}/{name}/grad/{k}"] = np_(v.grad)
    np.savez_compressed(os.path.join(HERE, "value_fn.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--only-value" in sys.argv:
        value_function_cases()
        sys.exit(0)
    if "--only-rl-loops" in sys.argv:
        rl_loops_cases()
        sys.exit(0)
    if "--only-detr-agent" in sys.argv:
        detr_agent_case()
        sys.exit(0)
    if "--only-bkl" in sys.argv:
        biased_kl_forward_cases()
        sys.exit(0)
    if "--only-sample" in sys.argv:
        sample_clip_decode()
        sys.exit(0)
    kat()
    losses_random()
    biased_kl_forward_cases()
    mha_cases()
    critic_case()
    agent_tiny()
    sample_clip_decode()
    detr_cases()
    detr_agent_case()
    rl_glue_cases()
    rl_loops_cases()
    loader_cases()
    value_function_cases()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
