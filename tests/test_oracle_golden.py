"""Pins oracle/bmhrl_oracle.py to the fixtures the reference itself produced (tests/golden/make_golden.py)
and to the hand-typed known answers of SURVEY.md Appendix A.  CPU only."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from bmhrl_amd import synthetic as syn
from oracle import bmhrl_oracle as O

T = torch.from_numpy


def close(a, b, tol=1e-5, floor=1e-6):
    """max|a-b| <= tol * max(max|b|, floor).  Gradients that are analytically zero (key biases under
    softmax shift invariance) are fp32 noise of ~1e-9 in the reference; ``floor`` keeps them out."""
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), floor)
    assert err <= tol, err


def agent_sd(g):
    shapes = {k: eval(s) for k, s in zip(g["state_keys"].tolist(), g["state_shapes"].tolist())}
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=0)
    cfg = syn.tiny_cfg()
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    return cfg, sd


def test_posenc_known_answers(golden):
    g = golden("kat")
    close(O.posenc_table(5, 8), g["pe8"], 1e-12)
    close(O.posenc_table(40, 20), g["pe20"], 1e-12)
    # SURVEY.md Appendix A1, typed by hand
    pe1 = [0.8414709848, 0.9504152803, 0.0998334166, 0.9995000417, 0.0099998333, 0.9999950000, 0.0009999998, 0.9999999500]
    close(O.posenc_table(3, 8)[1], np.array(pe1), 1e-9)


def test_attention_masking_known_answers(golden):
    g = golden("kat")
    Q = torch.ones(1, 1, 2, 4)
    K = torch.arange(12.).reshape(1, 1, 3, 4) / 10
    V = torch.arange(12.).reshape(1, 1, 3, 4)
    a = O.sdp_attention(Q, K, V, torch.tensor([[[[False, False, False]]]]))
    close(a, g["att_allmasked"])
    close(a[0, 0, 0], np.array([4., 5., 6., 7.]))  # A2: fully masked row -> uniform mean of V
    close(O.sdp_attention(Q, K, V, torch.tensor([[[[True, False, True]]]])), g["att_midmasked"])


def test_expand_goals_known_answer(golden):
    g = golden("kat")
    out = O.expand_goals(T(g["expand_in"]), T(g["expand_seg"]))
    close(out, g["expand_out"], 0)
    a3 = [[0, 0, 0, 0, 0, 0], [8, 8, 10, 10, 0, 0], [13, 14, 15, 16, 17, 18], [24] * 6, [27, 27, 27, 28, 29, 30]]
    close(out[:, :, 0], np.array(a3, dtype=np.float32), 0)


def test_losses_known_answers(golden):
    g = golden("kat")
    lp = T(g["a4_lp"])
    ls = O.label_smoothing(lp, torch.tensor([[2, 4, 1], [5, 3, 2]]), 0.7, 1)
    close(ls, g["a4_ls"])
    close(ls.sum(1), np.array([0.3006364, 0.2642519, 0.0, 0.6130040, 0.3660113, 0.3579281], dtype=np.float32), 2e-6)
    # the idx.sum()>0 guard: only padded flat index is 0 -> row NOT zeroed
    close(O.label_smoothing(lp, torch.tensor([[1, 4, 2], [5, 3, 2]]), 0.7, 1), g["a4_ls_guard"])
    bk = O.biased_kl_loss(lp, torch.tensor([[2, 4, 1], [5, 3, 2]]), torch.tensor([[2, 0, 3], [1, 3, 4]]),
                          torch.tensor([[.5, .25, 1.], [.8, 0., .1]]), 0.7, 1)
    close(bk, g["a4_bkl"])
    assert abs(float(bk.sum()) - 1.3885437) < 1e-5
    r = O.reinforce_loss(torch.exp(lp), torch.tensor([[2, 0, 3], [1, 3, 4]]), torch.tensor([[.1, .2, .3], [.4, .5, .6]]),
                         torch.tensor([[.3, .1, .0], [.2, .2, .9]]))
    close(r, g["a4_reinforce"])
    assert abs(float(r) - 0.1814150) < 1e-6


def test_losses_random_with_grads(golden):
    g = golden("losses")
    logits, trg, sampled = T(g["logits"]), T(g["trg"]), T(g["sampled"])
    score, baseline = T(g["score"]), T(g["baseline"])
    x = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(x, -1)
    close(O.label_smoothing(lp, trg, 0.7, 1), g["ls"])
    O.warmstart_loss(lp, trg, 0.7, 1).backward()
    close(x.grad, g["ls_grad_logits"])
    for stab, tag in ((False, "raw"), (True, "stab")):
        x = logits.clone().requires_grad_(True)
        lp = torch.log_softmax(x, -1)
        div, _ = O.worker_biased_kl(lp, trg, sampled, score, baseline, trg != 1, 0.7, 1, stab)
        close(div, g[f"bkl_{tag}"])
        O.worker_rl_loss(lp, trg, sampled, score, baseline, 0.7, 1, stab).backward()
        close(x.grad, g[f"bkl_{tag}_grad_logits"])  # includes the path through the amplitude (A5)
    x = logits.clone().requires_grad_(True)
    r = O.reinforce_loss(torch.softmax(x, -1), sampled, score, baseline)
    r.backward()
    close(r, g["reinforce"])
    close(x.grad, g["reinforce_grad_logits"])


@pytest.mark.parametrize("tag,dq,dk,H", [("self", 48, 48, 4), ("cross", 48, 24, 4), ("goal", 8, 20, 2)])
def test_mha(golden, tag, dq, dk, H):
    g = golden("mha")
    D = 64
    shapes = {}
    for n, (o, i) in {"linear_Q2d": (D, dq), "linear_K2d": (D, dk), "linear_V2d": (D, dk), "linear_d2Q": (dq, D)}.items():
        shapes[f"{n}.weight"] = (o, i)
        shapes[f"{n}.bias"] = (o,)
    sd = {"m." + k: v for k, v in syn.fill_state_dict(shapes, seed=3).items()}
    q, kv, mask = T(g[f"{tag}_q"]), T(g[f"{tag}_kv"]), T(g[f"{tag}_mask"])
    close(O.mha(sd, "m", q, kv, kv, mask, H), g[f"{tag}_out"])


def test_critic(golden):
    g = golden("critic")
    cfg = syn.tiny_cfg()
    sd = {"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()}
    close(O.segment_critic(sd, "critic", T(g["emb"])), g["out"])


def _tiny_inputs(cfg):
    B, Tv, Ta, L, V = 4, 7, 9, 6, 50
    b = syn.synthetic_batch(B, Tv, Ta, L, V, seed=7, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=3)
    cap = b["captions"]
    trg_in, trg_y = cap[:, :-1], cap[:, 1:]
    masks = O.make_masks(b["rgb"], b["audio"], trg_in, 1)
    return b, trg_in, trg_y, masks


def test_agent_forward_and_warmstart_grads(golden):
    g = golden("agent_tiny")
    cfg, sd = agent_sd(g)
    for k in sd:
        if not k.startswith("critic."):
            sd[k] = sd[k].clone().requires_grad_(True)
    b, trg_in, trg_y, masks = _tiny_inputs(cfg)
    x = (b["rgb"] + b["flow"], b["audio"])
    pred, wf, mf, goals, seg = O.agent_forward(sd, cfg, x, trg_in, masks)
    close(pred, g["pred"]); close(wf, g["worker_feat"]); close(mf, g["manager_feat"]); close(goals, g["goals"])
    assert np.array_equal(seg.numpy(), g["seg"])
    assert seg.sum() > 0, "fixture must exercise expand_goals"
    loss = O.warmstart_loss(pred, trg_y, 0.7, 1)
    close(loss, g["ws_loss"])
    loss.backward()
    n = 0
    for k in g:
        if k.startswith("ws_grad/"):
            name = k[len("ws_grad/"):]
            close(sd[name].grad, g[k], 1e-4, 1e-4)
            n += 1
    assert n > 100
    # parameters the reference never reaches get no gradient (hard parts)
    assert sd["bm_worker_fus.decoder.layers.0.feed_forward.fc1.weight"].grad is None
    assert "ws_grad/bm_worker_fus.decoder.layers.0.feed_forward.fc1.weight" not in g


def test_agent_mixed_and_rl_step(golden):
    g = golden("agent_tiny")
    cfg, sd = agent_sd(g)
    b, trg_in, trg_y, masks = _tiny_inputs(cfg)
    x = (b["rgb"] + b["flow"], b["audio"])
    close(O.agent_forward(sd, cfg, x, (trg_in, T(g["yhat"])), masks, 0.25)[0], g["pred_mixed"])
    for k in sd:
        if not k.startswith("critic."):
            sd[k] = sd[k].clone().requires_grad_(True)
    pred = O.agent_forward(sd, cfg, x, trg_in, masks)[0]
    loss = O.worker_rl_loss(pred, trg_y, T(g["rl_sampled"]), T(g["rl_score"]), None, 0.7, 1, False)
    close(loss, g["rl_loss"])
    loss.backward()
    for k in g:
        if k.startswith("rl_grad/"):
            close(sd[k[len("rl_grad/"):]].grad, g[k], 1e-4, 1e-4)


def test_sample_clip_greedy_decode(golden):
    """BASELINE config 1: reference sample clip, greedy decode (tokens produced by the reference)."""
    g = golden("sample_clip")
    cfg = syn.default_cfg(dout_p=0.0, rl_critic_score_threshhold=1.0)
    from bmhrl_amd.model.bm_hrl_agent import agent_state_shapes
    V = int(g["voc"])
    sd = syn.fill_state_dict(agent_state_shapes(cfg, V, with_critic=False), seed=0)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    rgb, flow, audio = T(g["rgb"]), T(g["flow"]), T(g["audio"])
    toks = O.greedy_decode(sd, cfg, rgb, flow, audio, 12, 2, 3, 1)
    assert np.array_equal(toks.numpy(), g["tokens"])


def test_detr_layers(golden):
    """Post-norm encoder / decoder stacks (model/encoder.py, model/decoder.py) against the reference's outputs."""
    g = golden("detr")
    enc, dec, d = syn.detr_tiny_modules()
    esd = {"enc." + k: v for k, v in enc.state_dict().items()}
    dsd = {"dec." + k: v for k, v in dec.state_dict().items()}
    H = d["H"]
    src, mask = T(g["src"]), T(g["mask"])
    mem_all = O.detr_stack(esd, "enc", 2, src, lambda p, x: O.detr_encoder_layer(esd, p, x, mask, H), True)
    close(mem_all, g["enc_out"])
    mem = T(g["enc_out"])[-1]
    tgt, qpos, qmask, objs, goal = (T(g[k]) for k in ("tgt", "qpos", "qmask", "objs", "goal"))
    a = O.detr_stack(dsd, "dec", 2, tgt, lambda p, x: O.detr_decoder_layer(dsd, p, x, mem, mask, qpos, qmask, None, None,
                                                                             True, objs, H), True)
    close(a, g["dec_a"])
    b = O.detr_stack(dsd, "dec", 2, tgt, lambda p, x: O.detr_decoder_layer(dsd, p, x, mem, mask, None, None, goal, qmask,
                                                                             False, None, H), True)
    close(b, g["dec_b"])


def test_value_functions_match_the_reference(golden):
    """oracle.value_function against BMWorkerValueFunction / BMManagerValueFunction of the reference
    (model/bm_hrl_agent.py:251-286; tests/golden/value_fn.npz)"""
    z = golden("value_fn")
    for d in (300, 48):
        feat = torch.from_numpy(z[f"d{d}/feat"])
        for name, seed in (("worker", 31), ("manager", 32)):
            keys = [str(k) for k in z[f"d{d}/{name}/keys"]]
            shapes = {"value_function.fc1.weight": (2 * d, d), "value_function.fc1.bias": (2 * d,),
                      "value_function.fc2.weight": (d, 2 * d), "value_function.fc2.bias": (d,),
                      "projection.weight": (1, d), "projection.bias": (1,)}
            assert sorted(shapes) == keys
            sd = syn.fill_state_dict(shapes, seed=seed)
            out = O.value_function(sd, feat)
            ref = torch.from_numpy(z[f"d{d}/{name}/out"])
            assert out.shape == ref.shape == (3, 6, 1)
            assert float((out - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def test_biased_kl_loops_against_the_reference(golden):
    """tests/golden/rl_loops.npz holds outputs of the reference's OWN epoch_loops/captioning_bmrl_loops.py:biased_kl (both
    branches, run with given rewards), get_norm_reward_factor and metrics/batched_meteor.py:segment_reward: the oracle's
    manager_biased_kl / manager_segment_loop / worker_biased_kl / segment_reward_loop are pinned to them."""
    g = golden("rl_loops")
    n = int(g["n"])
    assert n >= 40
    seen_empty_row0 = seen_no_segment = False
    for i in range(n):
        logits, trg, seg = T(g[f"logits{i}"]), T(g[f"trg{i}"]), T(g[f"seg{i}"])
        score, base, stab = T(g[f"score{i}"]), T(g[f"base{i}"]), bool(g[f"stab{i}"])
        mask = trg != 1
        seen_empty_row0 |= int(seg[0].sum()) == 0 and int(seg.sum()) > 0
        seen_no_segment |= int(seg.sum()) == 0
        # manager branch
        x = logits.clone().requires_grad_(True)
        div, sc, tok, _ = O.manager_biased_kl(torch.log_softmax(x, -1), trg, score, base, mask, seg, 0.7, 1, stab)
        div.sum().backward()
        assert torch.equal(tok, T(g[f"m_tok{i}"]))
        close(div, g[f"m_div{i}"], 1e-5)
        close(sc, g[f"m_score{i}"], 1e-6)
        close(x.grad, g[f"m_grad{i}"], 1e-5)
        # worker branch on the tokens the reference drew
        x = logits.clone().requires_grad_(True)
        div, sc = O.worker_biased_kl(torch.log_softmax(x, -1), trg, T(g[f"w_tok{i}"]), score, base, mask, 0.7, 1, stab)
        div.sum().backward()
        close(div, g[f"w_div{i}"], 1e-5)
        close(sc, g[f"w_score{i}"], 1e-6)
        close(x.grad, g[f"w_grad{i}"], 1e-5)
        # norm factors (:414-416) as the oracle forms them inside the two branches
        assert torch.equal(mask.sum(-1).reshape(-1, 1), T(g[f"nf_w{i}"]))
        assert torch.equal(seg.sum(-1).reshape(-1, 1), T(g[f"nf_m{i}"]))
        sr, idx = O.segment_reward_loop(score, seg)
        close(sr, g[f"sr{i}"], 1e-6)
        assert torch.equal(idx, T(g[f"sr_idx{i}"]))
    assert seen_empty_row0 and seen_no_segment


@pytest.mark.parametrize("tag", ["attached", "detached", "other"])
def test_biased_kl_forward_amplitude_is_an_argument(golden, tag):
    """oracle.biased_kl_loss against the reference's BiasedKL.forward (tests/golden/biased_kl_forward.npz, loss/biased_kl.py:22-53)
    with the amplitude attached to the prediction, detached, and attached through another function of it"""
    g = golden("biased_kl_forward")
    T = lambda k: torch.from_numpy(g[k])
    x = T("logits").clone().requires_grad_(True)
    lp = torch.log_softmax(x, -1)
    p = torch.gather(torch.exp(lp), 2, T("sampled").unsqueeze(-1)).squeeze(-1)
    n = (T("trg") != 1).sum(-1).reshape(-1, 1).float()
    amp = torch.clamp(T("score") * torch.sqrt(p) * 0.9 + 0.05, 0, 1) if tag == "other" else torch.clamp(T("score") * p * n, 0, 1)
    if tag == "detached":
        amp = amp.detach()
    div = O.biased_kl_loss(lp, T("trg"), T("sampled"), amp, 0.7, 1)
    (div.sum(-1) * T("up")).sum().backward()
    assert torch.allclose(div.sum(-1).detach(), T(f"{tag}_rows"), atol=1e-6)
    assert torch.allclose(x.grad, T(f"{tag}_grad_logits"), atol=1e-6)


def _detr_agent_state(g):
    keys = [str(k) for k in g["keys"]]
    shapes = {k: tuple(int(d) for d in str(s).split(",") if d != "") for k, s in zip(keys, g["shapes"])}
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=13)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(20, seed=1).items()})
    return sd, shapes


def test_detr_agent_input_projection_object_detector_and_encoder(golden):
    """oracle.conv1d_same_groupnorm / object_detect / the post-norm encoder stack against the reference's own DetrCaption
    (tests/golden/detr_agent.npz, model/det_bmhrl_agent.py:169-176 + model/object_detector.py:33-46)"""
    g = golden("detr_agent")
    sd, _ = _detr_agent_state(g)
    x, mask = torch.from_numpy(g["x_video"]), torch.from_numpy(g["V_mask"])
    vf = x
    for i in range(3):
        vf = O.conv1d_same_groupnorm(sd, f"input_proj.{i}", vf)
        assert torch.allclose(vf, torch.from_numpy(g[f"proj{i}"]), atol=2e-5), i
    logits, hs, ob_mask = O.object_detect(sd, "object_detector", vf, mask, 41)
    assert torch.allclose(hs, torch.from_numpy(g["obj_hs"]), atol=1e-4)
    assert torch.allclose(logits, torch.from_numpy(g["obj_logits"]), atol=1e-4)
    assert torch.equal(ob_mask, torch.from_numpy(g["obj_mask"]))
    mem = O.detr_stack(sd, "encoder", 3, vf, lambda q, t: O.detr_encoder_layer(sd, q, t, mask, 4), True, False)
    assert torch.allclose(mem, torch.from_numpy(g["memory"]), atol=1e-4)
