"""CPU oracle for the BMHRL hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a functional (state-dict driven) fp32 restatement of the reference's
bimodal transformer forward and of its per-token losses.  It exists only so that
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
can check / time the HIP path against the reference's arithmetic.  Nothing under
``bmhrl_amd/`` may import it: the product path fails loudly when the HIP library
is missing instead of falling back to this code.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the reference's own
``model/*.py`` and ``loss/*.py`` in the build container (CPU, torch 2.10) and wrote the
fixtures in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against them (and against the known-answer values of SURVEY.md
Appendix A) to <=1e-5.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  The code is written against a flat ``dict`` of tensors that
uses the reference's state-dict key names, so no ``nn.Module`` of the reference is
re-created here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
def posenc_table(seq_len: int, d_model: int) -> Tensor:
    """Sinusoid table, float64.  model/blocks.py:95-103.

    Quirk kept: every column uses ITS OWN index in the exponent (odd columns are
    cos(pos / 10000^(i/d)) with i odd, not (i-1)/d).
    """
    pos = np.arange(seq_len, dtype=np.float64)[:, None]
    col = np.arange(d_model, dtype=np.float64)[None, :]
    angle = pos / np.power(10000.0, col / d_model)
    tab = np.where((np.arange(d_model) % 2 == 0)[None, :], np.sin(angle), np.cos(angle))
    return torch.from_numpy(tab)


def add_posenc(x: Tensor) -> Tensor:
    """x + PE[:S] (dropout omitted: parity runs are eval / p=0).  model/blocks.py:105-112."""
    if x.dim() != 3:
        return x
    _, S, d = x.shape
    return x + posenc_table(S, d).to(x.dtype).unsqueeze(0)


def linear(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def layer_norm(sd: SD, name: str, x: Tensor) -> Tensor:
    """nn.LayerNorm over the last dim, eps 1e-5, affine.  model/blocks.py:132."""
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def sdp_attention(Q: Tensor, K: Tensor, V: Tensor, mask: Optional[Tensor]) -> Tensor:
    """softmax(Q K^T / sqrt(d_k), masked with -1e9) V.  model/multihead_attention.py:7-31.

    Masked logits are set to -1e9 (not -inf), so a fully masked row becomes the
    uniform distribution over all Sk keys.
    """
    d_k = Q.shape[-1]
    s = Q.matmul(K.transpose(-1, -2)) / math.sqrt(d_k)
    if mask is not None:
        s = s.masked_fill(~mask.bool(), -1e9)
    return torch.softmax(s, dim=-1).matmul(V)


def mha(sd: SD, name: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, mask: Optional[Tensor], H: int) -> Tensor:
    """MultiheadedAttention.forward.  model/multihead_attention.py:60-92."""
    B, Sq, _ = q_in.shape
    Q = linear(sd, name + ".linear_Q2d", q_in)
    K = linear(sd, name + ".linear_K2d", k_in)
    V = linear(sd, name + ".linear_V2d", v_in)
    D = Q.shape[-1]
    dk = D // H

    def split(t):
        return t.reshape(B, -1, H, dk).permute(0, 2, 1, 3)

    m = mask.unsqueeze(1) if mask is not None else None  # same mask for every head
    o = sdp_attention(split(Q), split(K), split(V), m)
    o = o.permute(0, 2, 1, 3).reshape(B, Sq, D)
    return linear(sd, name + ".linear_d2Q", o)


def ffn(sd: SD, name: str, x: Tensor) -> Tensor:
    """fc2(relu(fc1 x)).  model/blocks.py:175-187."""
    return linear(sd, name + ".fc2", torch.relu(linear(sd, name + ".fc1", x)))


# --------------------------------------------------------------------------------------
# encoder / fusion
# --------------------------------------------------------------------------------------
def encoder_layer(sd: SD, p: str, M1: Tensor, M2: Tensor, m1_mask: Tensor, m2_mask: Tensor, H: int):
    """BMEncoderLayer.forward (pre-LN residuals).  model/bm_hrl_agent.py:344-384.

    Cross attention: query = LN(own stream), key/value = the OTHER stream after its
    self-attention residual, NOT normalised (:362,:364).
    """
    n1 = layer_norm(sd, p + ".res_layers_M1.0.norm", M1)
    M1 = M1 + mha(sd, p + ".self_att_M1", n1, n1, n1, m1_mask, H)
    n2 = layer_norm(sd, p + ".res_layers_M2.0.norm", M2)
    M2 = M2 + mha(sd, p + ".self_att_M2", n2, n2, n2, m2_mask, H)

    q1 = layer_norm(sd, p + ".res_layers_M1.1.norm", M1)
    M1m2 = M1 + mha(sd, p + ".bi_modal_att_M1", q1, M2, M2, m2_mask, H)
    q2 = layer_norm(sd, p + ".res_layers_M2.1.norm", M2)
    M2m1 = M2 + mha(sd, p + ".bi_modal_att_M2", q2, M1, M1, m1_mask, H)

    M1m2 = M1m2 + ffn(sd, p + ".feed_forward_M1", layer_norm(sd, p + ".res_layers_M1.2.norm", M1m2))
    M2m1 = M2m1 + ffn(sd, p + ".feed_forward_M2", layer_norm(sd, p + ".res_layers_M2.2.norm", M2m1))
    return M1m2, M2m1


def bm_encoder(sd: SD, p: str, V: Tensor, A: Tensor, masks: Dict[str, Tensor], H: int, N: int):
    """BMEncoder.forward: N stacked layers, no final norm.  model/bm_hrl_agent.py:224-235."""
    for n in range(N):
        V, A = encoder_layer(sd, f"{p}.encoder.layers.{n}", V, A, masks["V_mask"], masks["A_mask"], H)
    return V, A


def fusion_layer(sd: SD, p: str, C: Tensor, mem_a: Tensor, mem_v: Tensor, masks: Dict[str, Tensor], H: int) -> Tensor:
    """BMFusionLayer.forward.  model/bm_hrl_agent.py:73-117.

    ``feed_forward`` exists in the state dict (:66) but is never applied.
    """
    n = layer_norm(sd, p + ".res_layer_self_att.norm", C)
    C = C + mha(sd, p + ".self_att", n, n, n, masks["C_mask"], H)
    Ca = C + mha(sd, p + ".enc_att_A", layer_norm(sd, p + ".res_layer_enc_att_A.norm", C), mem_a, mem_a, masks["A_mask"], H)
    Cv = C + mha(sd, p + ".enc_att_V", layer_norm(sd, p + ".res_layer_enc_att_V.norm", C), mem_v, mem_v, masks["V_mask"], H)
    Ca = layer_norm(sd, p + ".normCA", Ca)
    Cv = layer_norm(sd, p + ".normCV", Cv)
    g = torch.sigmoid(torch.clamp(sd[p + ".a_v_constant"], -2.0, 2.0))
    return g * Cv + (1.0 - g) * Ca


def bm_fusion(sd: SD, p: str, C: Tensor, mem_a: Tensor, mem_v: Tensor, masks, H: int, N: int) -> Tensor:
    """BMFusion.forward.  model/bm_hrl_agent.py:128-130."""
    for n in range(N):
        C = fusion_layer(sd, f"{p}.decoder.layers.{n}", C, mem_a, mem_v, masks, H)
    return C


# --------------------------------------------------------------------------------------
# critic (frozen LSTM -> AReLU -> GRU -> AReLU -> Linear)
# --------------------------------------------------------------------------------------
def arelu(x: Tensor, alpha: Tensor, beta: Tensor) -> Tensor:
    """model/bm_hrl_agent.py:13-23."""
    a = torch.clamp(alpha, 0.01, 0.99)
    b = 1.0 + torch.sigmoid(beta)
    return torch.relu(x) * b - torch.relu(-x) * a


def _lstm(sd: SD, p: str, x: Tensor, layers: int) -> Tensor:
    """batch_first multi-layer LSTM, zero initial state, torch gate order (i,f,g,o)."""
    B, L, _ = x.shape
    for l in range(layers):
        w_ih, w_hh = sd[f"{p}.weight_ih_l{l}"], sd[f"{p}.weight_hh_l{l}"]
        b = sd[f"{p}.bias_ih_l{l}"] + sd[f"{p}.bias_hh_l{l}"]
        Hd = w_hh.shape[1]
        h = x.new_zeros(B, Hd)
        c = x.new_zeros(B, Hd)
        xin = x.matmul(w_ih.t()) + b
        outs = []
        for t in range(L):
            g = xin[:, t] + h.matmul(w_hh.t())
            i, f, gg, o = g.split(Hd, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, dim=1)
    return x


def _gru(sd: SD, p: str, x: Tensor, layers: int) -> Tensor:
    """batch_first multi-layer GRU, zero initial state, torch gate order (r,z,n)."""
    B, L, _ = x.shape
    for l in range(layers):
        w_ih, w_hh = sd[f"{p}.weight_ih_l{l}"], sd[f"{p}.weight_hh_l{l}"]
        b_ih, b_hh = sd[f"{p}.bias_ih_l{l}"], sd[f"{p}.bias_hh_l{l}"]
        Hd = w_hh.shape[1]
        h = x.new_zeros(B, Hd)
        xin = x.matmul(w_ih.t()) + b_ih
        outs = []
        for t in range(L):
            hh = h.matmul(w_hh.t()) + b_hh
            xr, xz, xn = xin[:, t].split(Hd, dim=1)
            hr, hz, hn = hh.split(Hd, dim=1)
            r = torch.sigmoid(xr + hr)
            z = torch.sigmoid(xz + hz)
            n = torch.tanh(xn + r * hn)
            h = (1.0 - z) * n + z * h
            outs.append(h)
        x = torch.stack(outs, dim=1)
    return x


def segment_critic(sd: SD, p: str, emb: Tensor) -> Tensor:
    """SegmentCritic.forward, (B,L,300)->(B,L,1), no grad.  model/bm_hrl_agent.py:204-215."""
    with torch.no_grad():
        h = _lstm(sd, p + ".lstm", emb, 4)
        h = arelu(h, sd[p + ".relu.alpha"], sd[p + ".relu.beta"])
        h = _gru(sd, p + ".gru", h, 2)
        h = arelu(h, sd[p + ".relu2.alpha"], sd[p + ".relu2.beta"])
        return linear(sd, p + ".lin", h)


def segment_labels(score: Tensor, threshold: float) -> Tensor:
    """(sigmoid(critic) > thr).squeeze().int().  model/bm_hrl_agent.py:638-640."""
    return (torch.sigmoid(score) > threshold).squeeze().int()


# --------------------------------------------------------------------------------------
# manager / worker
# --------------------------------------------------------------------------------------
def expand_goals(goals: Tensor, seg: Tensor) -> Tensor:
    """Manager.expand_goals restated as an out-of-place loop.  model/bm_hrl_agent.py:415-429.

    Non-zero labels are visited in row-major order.  Each one copies the goal at the
    segment end over [previous end + 1 .. end].  When the row changes, the PREVIOUS
    row's tail is zeroed; the previous row starts at 0, so an initial run of rows
    without labels is not touched except row 0, which is zeroed from column 0 when
    the first visited row is not row 0.  The last visited row keeps its tail.
    The reference writes in place; autograd sees the same values as this version.
    """
    out = goals.clone()
    seg2 = seg.reshape(goals.shape[0], goals.shape[1]) if seg.dim() != 2 else seg
    prev_b, prev_end = 0, 0
    for b, l in torch.nonzero(seg2).tolist():
        if b != prev_b:
            out[prev_b, prev_end:] = 0
            prev_end = 0
            prev_b = b
        out[b, prev_end:l + 1] = out[b, l]
        prev_end = l + 1
    return out


def manager(sd: SD, p: str, feat: Tensor, seg: Tensor) -> Tensor:
    """Manager.forward with exploration off.  model/bm_hrl_agent.py:437-454 (core unused, :438)."""
    return expand_goals(linear(sd, p + ".linear", feat), seg)


def worker(sd: SD, p: str, feat: Tensor, goals: Tensor, c_mask: Tensor) -> Tensor:
    """Worker.forward: goal attention (H=2) + Linear(364->V) + log_softmax.  :480-487, :456-466."""
    gc = mha(sd, p + ".goal_attention", goals, feat, feat, c_mask, 2)
    logits = linear(sd, p + ".core.projection", torch.cat([feat, gc], dim=-1))
    return torch.log_softmax(logits, dim=-1)


# --------------------------------------------------------------------------------------
# masks (model/masking.py:3-55)
# --------------------------------------------------------------------------------------
def make_masks(rgb: Tensor, audio: Tensor, captions: Optional[Tensor], pad_idx: int) -> Dict[str, Tensor]:
    """audio_video modality: V_mask from rgb[:,:,0] (before flow is added), A_mask from
    audio[:,:,0], C_mask = key-pad & lower-triangular.  model/masking.py:18-25,44-50."""
    masks = {"V_mask": (rgb[:, :, 0] != 0).unsqueeze(1), "A_mask": (audio[:, :, 0] != 0).unsqueeze(1)}
    if captions is not None:
        L = captions.shape[-1]
        tril = torch.tril(torch.ones(1, L, L, dtype=torch.bool))
        masks["C_mask"] = (captions != pad_idx).unsqueeze(-2) & tril
    return masks


# --------------------------------------------------------------------------------------
# whole agent
# --------------------------------------------------------------------------------------
def agent_forward(sd: SD, cfg, x: Tuple[Tensor, Tensor], trg, masks: Dict[str, Tensor], factor: float = 1.0):
    """BMHrlAgent.forward / prediction / mixed_prediction / predict_with_features.

    model/bm_hrl_agent.py:611-661.  x = (V = rgb+flow, A); returns
    (log-probs (B,L,V), worker_feat, manager_feat, goals, segment_labels).
    """
    xv, xa = x
    d_caps = sd["emb_C.embedder.weight"].shape[1]
    emb = lambda t: F.embedding(t, sd["emb_C.embedder.weight"]) * math.sqrt(d_caps)  # model/blocks.py:44-48
    if isinstance(trg, tuple):
        y, yhat = trg
        C = emb(y) * (1 - factor) + emb(yhat) * factor
    else:
        C = emb(trg)
    V = add_posenc(xv)
    A = add_posenc(xa)
    seg = segment_labels(segment_critic(sd, "critic", C), cfg.rl_critic_score_threshhold)
    C = add_posenc(C)
    H, N = cfg.rl_att_heads, cfg.rl_att_layers
    enc_v, enc_a = bm_encoder(sd, "bm_enc", V, A, masks, H, N)
    w_feat = bm_fusion(sd, "bm_worker_fus", C, enc_a, enc_v, masks, H, N)
    m_feat = bm_fusion(sd, "bm_manager_fus", C, enc_a, enc_v, masks, H, N)
    goals = manager(sd, "manager", m_feat, seg)
    pred = worker(sd, "worker", w_feat, goals, masks["C_mask"])
    return pred, w_feat, m_feat, goals, seg


def value_function(sd: SD, feat: Tensor) -> Tensor:
    """BM{Worker,Manager}ValueFunction.forward: Linear(relu(FFN(feat))).  :263-269, :282-286."""
    return linear(sd, "projection", torch.relu(ffn(sd, "value_function", feat)))


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def _kl_none(logp: Tensor, target: Tensor) -> Tensor:
    """F.kl_div(logp, target, reduction='none') = xlogy(t, t) - t*logp (0 where t == 0)."""
    return torch.xlogy(target, target) - target * logp


def label_smoothing(pred: Tensor, target: Tensor, smoothing: float, pad_idx: int) -> Tensor:
    """LabelSmoothing.forward -> unreduced (B*S, V).  loss/label_smoothing.py:12-32.

    Quirk kept: pad rows are zeroed only when the SUM of the padded flat indices is
    positive, so a batch whose only pad target sits at flat index 0 is not zeroed.
    """
    B, S, V = pred.shape
    lp = pred.reshape(-1, V)
    t = target.reshape(-1).long()
    dist = torch.full_like(lp, smoothing / (V - 2))
    dist[torch.arange(t.numel()), t] = 1.0 - smoothing
    dist[:, pad_idx] = 0
    pad_rows = torch.nonzero(t == pad_idx).reshape(-1)
    if pad_rows.numel() > 0 and int(pad_rows.sum()) > 0:
        dist[pad_rows] = 0
    return _kl_none(lp, dist)


def biased_kl_loss(pred: Tensor, trg: Tensor, biased_trg: Tensor, biased_offset: Tensor, smoothing: float, pad_idx: int) -> Tensor:
    """BiasedKL.forward -> unreduced (B*S, V).  loss/biased_kl.py:22-53.

    Order kept: smoothed prior, scatter 0.3*(1-amp) at the target, zero the pad column,
    ADD 0.3*amp at the sampled token, zero pad rows (same index-sum guard), +1e-8, KL.
    Differentiable w.r.t. ``biased_offset`` (the caller leaves it attached).
    """
    B, S, V = pred.shape
    keep = 1.0 - smoothing
    lp = pred.reshape(-1, V)
    t = trg.reshape(-1).long()
    a = biased_trg.reshape(-1).long()
    off = biased_offset.reshape(-1)
    rows = torch.arange(t.numel())
    dist = torch.full_like(lp, smoothing / (V - 2))
    dist = dist.index_put((rows, t), (keep * (1.0 - off)).to(lp.dtype))
    col = torch.ones(V, dtype=lp.dtype)
    col[pad_idx] = 0
    dist = dist * col
    dist = dist + torch.zeros_like(lp).index_put((rows, a), (keep * off).to(lp.dtype))
    pad_rows = torch.nonzero(t == pad_idx).reshape(-1)
    if pad_rows.numel() > 0 and int(pad_rows.sum()) > 0:
        rowmask = torch.ones(t.numel(), 1, dtype=lp.dtype)
        rowmask[pad_rows] = 0
        dist = dist * rowmask
    return _kl_none(lp, dist + 1e-8)


def reinforce_loss(probs: Tensor, action: Tensor, value: Tensor, critic_value: Tensor) -> Tensor:
    """Reinforce.forward.  loss/biased_kl.py:69-81."""
    eps = 1e-5
    p = torch.clamp(probs, eps, 1 - eps)
    pa = torch.gather(p, -1, action.long().unsqueeze(-1)).squeeze(-1)
    adv = value - critic_value
    return -(adv.detach() * torch.log(pa)).mean() + (adv ** 2).mean()


def amplitude(score: Tensor, sampled_probs: Tensor, norm_factor: Tensor) -> Tensor:
    """clamp(score * p(a) * n, 0, 1).  epoch_loops/captioning_bmrl_loops.py:409-411."""
    return torch.clamp(score.float() * sampled_probs.float() * norm_factor.float(), 0, 1)


def worker_biased_kl(pred: Tensor, trg: Tensor, sampled: Tensor, score: Tensor, baseline: Tensor, loss_mask: Tensor,
                     smoothing: float, pad_idx: int, stabilize: bool) -> Tuple[Tensor, Tensor]:
    """Worker branch of ``biased_kl`` with the sample and the reward given.

    epoch_loops/captioning_bmrl_loops.py:271-334 (train_worker=True): p = exp(pred)
    gathered at the sampled token; norm factor = tokens per row (:414-416); optional
    baseline subtraction (:318-319); amplitude stays attached to ``pred``.
    Returns (unreduced divergence (B*S, V), score used for the value loss).
    """
    p = torch.gather(torch.exp(pred), 2, sampled.unsqueeze(-1)).squeeze(-1)
    n = loss_mask.sum(dim=-1).reshape(-1, 1)
    if stabilize:
        score = (score - baseline) * loss_mask.float()
    amp = amplitude(score, p, n)
    return biased_kl_loss(pred, trg, sampled, amp, smoothing, pad_idx), score


def manager_biased_kl(pred: Tensor, trg: Tensor, score: Tensor, baseline: Tensor, loss_mask: Tensor, segments: Tensor,
                      smoothing: float, pad_idx: int, stabilize: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Manager branch of ``biased_kl`` with the reward given (epoch_loops/captioning_bmrl_loops.py:271-334,
    train_worker=False): arg-max tokens (:283-284), ``score * segments`` (:300), per-segment product of the arg-max
    probabilities and per-segment sum of the expected scores by the Python loop (:301-316, ``manager_segment_loop``),
    optional baseline subtraction (:318-319), norm factor = segments per row (:414-416); the amplitude stays attached to
    ``pred`` through every probability of the product.
    Returns (unreduced divergence (B*S, V), score, arg-max tokens, amplitude)."""
    probs = torch.exp(pred)
    sampled = torch.argmax(probs, dim=-1)
    p = torch.gather(probs, 2, sampled.unsqueeze(-1)).squeeze(-1)
    n = segments.sum(dim=-1).reshape(-1, 1)
    score = score.float() * segments.float()
    seg_prob, expected = manager_segment_loop(p, baseline.float(), segments)
    if stabilize:
        score = (score - expected) * loss_mask.float()
    amp = amplitude(score, seg_prob, n)
    return biased_kl_loss(pred, trg, sampled, amp, smoothing, pad_idx), score, sampled, amp


def warmstart_loss(pred: Tensor, trg_y: Tensor, smoothing: float, pad_idx: int) -> Tensor:
    """sum(LabelSmoothing) / n_tokens.  epoch_loops/captioning_bmrl_loops.py:1156-1158."""
    n_tokens = (trg_y != pad_idx).sum()
    return label_smoothing(pred, trg_y, smoothing, pad_idx).sum() / n_tokens


def worker_rl_loss(pred, trg_y, sampled, score, baseline, smoothing, pad_idx, stabilize) -> Tensor:
    """sum(divergence) / (n_tokens * 4/20).  epoch_loops/captioning_bmrl_loops.py:829-833,846-862."""
    loss_mask = trg_y != pad_idx
    div, _ = worker_biased_kl(pred, trg_y, sampled, score, baseline, loss_mask, smoothing, pad_idx, stabilize)
    return div.sum() / (loss_mask.sum() * (4.0 / 20.0))


def masked_value_loss(expected: Tensor, score: Tensor, loss_mask: Tensor) -> Tensor:
    """mean(MSE(expected, score) * mask).  epoch_loops/captioning_bmrl_loops.py:873-876."""
    return (((expected - score.float()) ** 2) * loss_mask.float()).mean()


def greedy_decode(sd: SD, cfg, rgb: Tensor, flow: Tensor, audio: Tensor, max_len: int, start_idx: int, end_idx: int, pad_idx: int) -> Tensor:
    """Arg-max autoregressive decode with the whole model re-run per token.

    epoch_loops/captioning_bmrl_loops.py:127-152 (3-argument ``inference``, :653-654).
    """
    B = audio.shape[0]
    trg = torch.full((B, 1), start_idx, dtype=torch.long)
    done = torch.zeros(B, 1, dtype=torch.bool)
    with torch.no_grad():
        while trg.shape[-1] <= max_len and not bool(done.all()):
            masks = make_masks(rgb, audio, trg, pad_idx)
            pred = agent_forward(sd, cfg, (rgb + flow, audio), trg, masks)[0]
            nxt = pred[:, -1].argmax(dim=-1, keepdim=True)
            trg = torch.cat([trg, nxt], dim=-1)
            done = done | (nxt == end_idx)
    return trg


# --------------------------------------------------------------------------------------
# post-norm (DETR-style) layers -- model/encoder.py, model/decoder.py
# --------------------------------------------------------------------------------------
def causal_mask(mask: Optional[Tensor], causal: bool) -> Optional[Tensor]:
    """model/multihead_attention.py:17-22: with `causal`, logits above the diagonal are filled with -1e9 before the mask
    is applied -- the same as masking with (mask AND lower-triangular).  Skipped when there is no mask at all (:17)."""
    if mask is None or not causal:
        return mask
    S = mask.shape[-1]
    return mask.bool() & torch.ones(S, S, dtype=torch.bool).tril().unsqueeze(0)


def detr_encoder_layer(sd: SD, p: str, src: Tensor, mask: Optional[Tensor], H: int) -> Tensor:
    """TransformerEncoderLayer.forward_post, model/encoder.py:59-69 (pos = PositionalEncoder, dropout off)."""
    qk = add_posenc(src)
    src = layer_norm(sd, p + ".norm1", src + mha(sd, p + ".self_attn", qk, qk, src, mask, H))
    ff = linear(sd, p + ".linear2", torch.relu(linear(sd, p + ".linear1", src)))
    return layer_norm(sd, p + ".norm2", src + ff)


def detr_decoder_layer(sd: SD, p: str, tgt: Tensor, memory: Tensor, memory_mask, query_pos, query_mask, goal, goal_mask,
                       add_pos: bool, detected_objects, H: int) -> Tensor:
    """TransformerDecoderLayer.forward_post, model/decoder.py:66-100.  `query_pos` is None for "PositionalEncoder"
    (add_pos=False, causal self attention) or a tensor added to tgt (add_pos=True).  norm1 is applied to tgt before the
    self-attention branch is added (:77-78); `forward` always passes obj_mask=None (:106)."""
    qk = add_posenc(tgt) if not add_pos else tgt + query_pos
    branch = mha(sd, p + ".self_attn", qk, qk, tgt, causal_mask(query_mask, not add_pos), H)
    tgt = layer_norm(sd, p + ".norm1", tgt) + branch
    tgt = layer_norm(sd, p + ".norm2", tgt + mha(sd, p + ".multihead_attn", qk, add_posenc(memory), memory, memory_mask, H))
    if goal is not None:
        branch = mha(sd, p + ".goal_attention", add_posenc(tgt), add_posenc(goal), goal, goal_mask, H)
        tgt = layer_norm(sd, p + ".norm4", tgt + branch)
    if detected_objects is not None:
        tgt = layer_norm(sd, p + ".norm5", tgt + mha(sd, p + ".detected_attention", qk, detected_objects, detected_objects, None, H))
    ff = linear(sd, p + ".linear2", torch.relu(linear(sd, p + ".linear1", tgt)))
    return layer_norm(sd, p + ".norm3", tgt + ff)


def detr_stack(sd: SD, p: str, n_layers: int, x: Tensor, layer_fn, has_norm: bool, return_intermediate: bool = True) -> Tensor:
    """TransformerEncoder.forward / TransformerDecoder.forward, model/encoder.py:20-37, model/decoder.py:17-37: with a
    final norm the last intermediate becomes norm(norm(output)) (output is normalised, then normalised again on append)."""
    inter = []
    for i in range(n_layers):
        x = layer_fn(f"{p}.layers.{i}", x)
        inter.append(x)
    if has_norm:
        x = layer_norm(sd, p + ".norm", x)
        inter[-1] = layer_norm(sd, p + ".norm", x)
    return torch.stack(inter) if return_intermediate else x


def conv1d_same_groupnorm(sd: SD, p: str, x: Tensor, groups: int = 32) -> Tensor:
    """One block of DetrCaption.input_proj, model/det_bmhrl_agent.py:79-86 applied at :169-174: Conv1d(C, C, k,
    padding='same') over the time axis of x (B, T, C) followed by GroupNorm(groups, C) (eps 1e-5); `p` = "input_proj.i".
    'same' padding: k - 1 zeros in total, (k - 1) // 2 in front (the extra one of an even kernel goes behind)."""
    w, b = sd[p + ".0.weight"], sd[p + ".0.bias"]
    k = w.shape[-1]
    left = (k - 1) // 2
    xt = torch.nn.functional.pad(x.transpose(1, 2), (left, k - 1 - left))
    y = torch.nn.functional.conv1d(xt, w, b)
    y = torch.nn.functional.group_norm(y, groups, sd[p + ".1.weight"], sd[p + ".1.bias"], 1e-5)
    return y.transpose(1, 2)


def object_detect(sd: SD, p: str, x: Tensor, mask: Tensor, voc_size: int):
    """ObjectDetect.forward, model/object_detector.py:33-46: 256-wide projection, 6 post-norm encoder layers (H = 4), 6 decoder
    layers over 100 learned queries (tgt = 0, query positions added: add_pos=True), class logits, the detached query states and
    the "no object" mask (arg-max class == voc_size)."""
    src = linear(sd, p + ".input_projection", x)
    memory = detr_stack(sd, p + ".encoder", 6, src, lambda q, t: detr_encoder_layer(sd, q, t, mask, 4), True, False)
    qpos = sd[p + ".query_embed.weight"].unsqueeze(0).repeat(x.shape[0], 1, 1)
    hs = detr_stack(sd, p + ".decoder", 6, torch.zeros_like(qpos),
                    lambda q, t: detr_decoder_layer(sd, q, t, memory, mask, qpos, None, None, None, True, None, 4), True, False)
    logits = linear(sd, p + ".class_embed", hs)
    return logits, hs.detach(), (logits.argmax(-1) == voc_size).detach()


def detr_caption_forward(sd: SD, cfg, x_video: Tensor, trg: Tensor, masks: Dict[str, Tensor]):
    """DetrCaption.forward with use_manager = False and pre_goal_attention = False, model/det_bmhrl_agent.py:158-208:
    (log-probs, worker features[..., :300], encoder memory, class logits).  The end token (3) is embedded as padding (1), :161-162;
    the caption decoder runs causal self attention (add_pos=False) -- the branch the CPU reference cannot execute
    (model/multihead_attention.py:19) -- and attends the detached object queries without a mask (model/decoder.py:106)."""
    H, V = cfg.rl_att_heads, sd["linear.weight"].shape[0]
    trg = trg.clone()
    trg[trg == 3] = 1
    C = sd["emb_C.embedder.weight"][trg] * math.sqrt(cfg.d_model_caps)
    vf = x_video
    for i in range(3):
        vf = conv1d_same_groupnorm(sd, f"input_proj.{i}", vf)
    cls, hs, _ = object_detect(sd, "object_detector", vf, masks["V_mask"], V)
    memory = detr_stack(sd, "encoder", 3, vf, lambda q, t: detr_encoder_layer(sd, q, t, masks["V_mask"], H), True, False)
    feat = detr_stack(sd, "worker_decoder", 3, C,
                      lambda q, t: detr_decoder_layer(sd, q, t, memory, masks["V_mask"], None, masks["C_mask"], None, None, False, hs, H),
                      True, False)
    pred = torch.log_softmax(linear(sd, "linear", feat), -1)
    return pred, feat[:, :, :300], memory, cls


# --------------------------------------------------------------------------------------
# host-side RL glue of the reference, restated as the loops they are
# discontinue_reward_loop is pinned to the reference's own function (tests/golden/rl_glue.npz: metrics/util.py loads
# without nltk); the two segment loops and both branches of biased_kl are pinned to outputs of the reference's own
# epoch_loops/captioning_bmrl_loops.py:biased_kl and metrics/batched_meteor.py:segment_reward (tests/golden/rl_loops.npz,
# made by tests/golden/make_golden.py:rl_loops_cases; tests/test_oracle_golden.py::test_biased_kl_loops_against_the_reference).
# --------------------------------------------------------------------------------------
def manager_segment_loop(sampled_probs: Tensor, expected_scores: Tensor, segments: Tensor) -> Tuple[Tensor, Tensor]:
    """epoch_loops/captioning_bmrl_loops.py:301-316 (manager branch of biased_kl): per segment (positions after the
    previous segment end up to and including a marked position) the product of the sampled probabilities and the sum of
    the expected scores, written over the whole segment.  When the loop moves on to a later batch row, the tail of the
    row it leaves (after its last segment end) is zeroed in both tensors -- the initial `old_b = 0` makes that row 0,
    from position 0, when row 0 has no segment; the last row with segments keeps its tail.
    Returns (segment_prob, expected_scores'); expected_scores is not modified in place here."""
    B, L = sampled_probs.shape
    es = expected_scores.clone()
    sp = torch.zeros(B, L, dtype=torch.float32)
    old_l = old_b = 0
    for b, l in torch.nonzero(segments).tolist():
        if b != old_b:
            es[old_b, old_l:] = 0
            sp[old_b, old_l:] = 0
            old_b, old_l = b, 0
        sp[b, old_l:l + 1] = torch.prod(sampled_probs[b, old_l:l + 1])
        es[b, old_l:l + 1] = torch.sum(es[b, old_l:l + 1])
        old_l = l + 1
    return sp, es


def segment_reward_loop(reward: Tensor, sections: Tensor) -> Tuple[Tensor, Tensor]:
    """metrics/batched_meteor.py:19-36: every segment gets the sum of its rewards on all of its positions, positions
    behind the last segment end stay 0.  Returns (segment_reward, segment_indices)."""
    B, L = reward.shape
    out = torch.zeros(B, L, dtype=torch.float32)
    idx = torch.nonzero(sections)
    old_l = old_b = 0
    for b, l in idx.tolist():
        if b != old_b:
            out[b, old_l:] = 0
            old_b, old_l = b, 0
        out[b, old_l:l + 1] = torch.sum(reward[b, old_l:l + 1])
        old_l = l + 1
    return out, idx


def discontinue_reward_loop(cider_diff: Tensor, gamma: float, n_step: int = 100, segments: Optional[Tensor] = None) -> Tensor:
    """metrics/util.py:53-88.  Without segments: discounted return over at most n_step following positions.  With
    segments: the reward of a segment end plus the discounted rewards of the later segment ends of its row is written
    over the segment -- but the reference guards the accumulation with `i < segment_idx.shape[0]` (== 2, the width of one
    index pair), so only the first two segment ends of the whole batch see later rewards; kept.  On a row change the
    tail of the row left behind is zeroed; the last row keeps its tail."""
    if segments is None:
        B, L = cider_diff.shape
        out = torch.zeros(B, L, dtype=torch.float32)
        for b in range(B):
            for t in range(L):
                acc = 0.0
                for i in range(min(n_step, L - t)):
                    acc = acc + (gamma ** i) * float(cider_diff[b, t + i])
                out[b, t] = acc
        return out
    cd = cider_diff.clone()
    idx = torch.nonzero(segments).tolist()
    old_l = old_b = 0
    for i, (b, l) in enumerate(idx):
        disc = cd[b, l].clone()
        if old_b != b:
            cd[old_b, old_l:] = 0
            old_l, old_b = 0, b
        if i < 2:
            for n, (b2, l2) in enumerate(idx[i + 1:]):
                if b == b2:
                    disc = disc + (gamma ** (n + 1)) * cd[b2, l2]
                else:
                    break
        cd[b, old_l:l + 1] = disc
        old_l = l + 1
    return cd


# --------------------------------------------------------------------------------------
# Feature loader.  Pinned to the reference's own functions by tests/golden/loader.npz
# (captioning_datasets/load_features.py imports only numpy / torch and runs here).
# --------------------------------------------------------------------------------------
def crop_a_segment_loop(feature: Tensor, start: float, end: float, duration: float) -> Optional[Tensor]:
    """captioning_datasets/load_features.py:14-35: rows [int(S*start/duration), int(S*end/duration)) of a (S, D) stack; an
    empty range becomes one row ([S:S] -> [S-1:S] at the very end, [i:i] -> [i:i+1] elsewhere); None when nothing is left
    (e.g. start beyond the stack)."""
    S = feature.shape[0]
    a = int(S * (start / duration))
    b = int(S * (end / duration))
    if a == b:
        if a == S:
            a -= 1
        else:
            b += 1
    out = feature[a:b, :]
    return None if len(out) == 0 else out


def batch_feature_stacks_loop(samples, pad_idx: float) -> Dict[str, Tensor]:
    """captioning_datasets/captioning_dataset.py:262-290: `samples` is a list of (rgb, flow, audio) stacks, each (S_i, D)
    or None for a clip whose file is missing / whose crop is empty (-> one zero row, :268-278).  rgb and audio are padded
    to the longest clip of the batch with pad_idx, flow with 0 (it is summed onto rgb later)."""
    rgb, flow, aud = [], [], []
    for r, f, a in samples:
        if r is None and f is None:
            r, f = torch.zeros(1, 1024), torch.zeros(1, 1024)
        if a is None:
            a = torch.zeros(1, 128)
        rgb.append(r.float()); flow.append(f.float()); aud.append(a.float())

    def pad(seqs, value):
        T = max(s.shape[0] for s in seqs)
        out = torch.full((len(seqs), T, seqs[0].shape[1]), float(value))
        for i, s in enumerate(seqs):
            out[i, :s.shape[0]] = s
        return out
    return {"rgb": pad(rgb, pad_idx), "flow": pad(flow, 0.0), "audio": pad(aud, pad_idx)}
