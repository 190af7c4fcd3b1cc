"""BMHrlAgent and its parts with the reference's public surface (model/bm_hrl_agent.py): same constructor
(`cfg`, `train_dataset`), `forward / warmstart / inference / teach_* / set_inference_mode / save_model / load_model`,
attribute names and the 307-key state-dict layout, so reference checkpoints load and the reference's driver
(scripts/train_rl_captioning_module.py) can construct it unchanged.  All arithmetic of the bimodal encoder, the two
fusion stacks, manager and worker runs in the gfx950 HIP kernels (bmhrl_amd.functional).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import ExpandGoalsFn, FusionTailFn, GateFn, LayerNormFn, LinearFn, WorkerHeadFn
from .blocks import LayerStack, PositionalEncoder, PositionwiseFeedForward, ResidualConnection, VocabularyEmbedder, clone
from .multihead_attention import MultiheadedAttention


class AReLU(nn.Module):
    """reference model/bm_hrl_agent.py:13-23"""

    def __init__(self, alpha=0.90, beta=2.0):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor([alpha]))
        self.beta = nn.Parameter(torch.tensor([beta]))

    def forward(self, x):
        return F.relu(x) * (1 + torch.sigmoid(self.beta)) - F.relu(-x) * torch.clamp(self.alpha, min=0.01, max=0.99)


class ModelBase(nn.Module):
    """reference :26-37 -- checkpoint = plain state_dict at <dir>/<name>.pt"""

    def __init__(self, name):
        super().__init__()
        self.name = name

    def save_model(self, checkpoint_dir):
        torch.save(self.state_dict(), checkpoint_dir + f"/{self.name}.pt")

    def load_model(self, checkpoint_dir):
        self.load_state_dict(torch.load(checkpoint_dir + f"/{self.name}.pt"))


class BMEncoderLayer(nn.Module):
    """reference :328-384.  Three pre-norm residual blocks per modality: self attention, cross attention whose
    keys/values are the OTHER stream after its self-attention block (un-normalised), feed forward."""

    def __init__(self, d_model_M1, d_model_M2, d_model, d_ff_M1, d_ff_M2, dout_p, H):
        super().__init__()
        self.self_att_M1 = MultiheadedAttention(d_model_M1, d_model_M1, d_model_M1, H, dout_p, d_model)
        self.self_att_M2 = MultiheadedAttention(d_model_M2, d_model_M2, d_model_M2, H, dout_p, d_model)
        self.bi_modal_att_M1 = MultiheadedAttention(d_model_M1, d_model_M2, d_model_M2, H, dout_p, d_model)
        self.bi_modal_att_M2 = MultiheadedAttention(d_model_M2, d_model_M1, d_model_M1, H, dout_p, d_model)
        self.feed_forward_M1 = PositionwiseFeedForward(d_model_M1, d_ff_M1, dout_p)
        self.feed_forward_M2 = PositionwiseFeedForward(d_model_M2, d_ff_M2, dout_p)
        self.res_layers_M1 = clone(ResidualConnection(d_model_M1, dout_p), 3)
        self.res_layers_M2 = clone(ResidualConnection(d_model_M2, dout_p), 3)

    # The two modality streams of a layer are independent inside each of its three blocks (they only exchange the
    # self-attention outputs).  On the GPU the audio half runs as a parallel branch on a second HIP stream: its K=128
    # GEMMs are bandwidth-bound, the video half is MFMA-bound, and either alone leaves tails of idle CUs (two GEMM
    # chains as parallel graph branches measure 11-21 % faster than back to back on MI355X, tests/bench_streams.py).
    modality_side_stream = True
    _side = None

    def _fork(self, device):
        cls = BMEncoderLayer
        if cls._side is None or cls._side.device != device:
            cls._side = torch.cuda.Stream(device=device)
        return torch.cuda.current_stream(), cls._side

    # Attentions whose keys / values are rows narrower than a head (the 128-wide audio stream against d_k = 256) run in
    # the absorbed-projection form (functional.MemAttnFn): scores_h = (Q_h Wk_h) A^T, context_h = P_h A -- one key /
    # value tile for all heads, half the attention FLOPs, no K|V projection of the 12 800 audio rows.
    absorb_narrow_memory = True

    def _self_att_M2(self, M2, M2_mask):
        att, norm = self.self_att_M2, self.res_layers_M2[0].norm
        if self.absorb_narrow_memory and M2.is_cuda and att.d_model_K == 128 and att.d_k > 128:   # (the fused kernel's width)
            return att.fused_memory(M2, None, M2_mask, norm, emit_bf16=True)
        return att.fused(M2, None, M2_mask, norm, residual=True, emit_bf16=True)

    def _cross_M1(self, M1, M2, M2_mask):
        att, norm = self.bi_modal_att_M1, self.res_layers_M1[1].norm
        if self.absorb_narrow_memory and M1.is_cuda and att.d_model_K == 128 and att.d_k > 128:
            return att.fused_memory(M1, M2, M2_mask, norm)
        return att.fused(M1, M2, M2_mask, norm, residual=True)

    def forward(self, x, masks):
        M1, M2 = x
        M1_mask, M2_mask = masks
        if not (M1.is_cuda and self.modality_side_stream):
            M1 = self.self_att_M1.fused(M1, None, M1_mask, self.res_layers_M1[0].norm, residual=True, emit_bf16=True)
            M2 = self._self_att_M2(M2, M2_mask)
            M1m2 = self._cross_M1(M1, M2, M2_mask)
            M2m1 = self.bi_modal_att_M2.fused(M2, M1, M1_mask, self.res_layers_M2[1].norm, residual=True)
            M1m2 = self.feed_forward_M1.fused(M1m2, self.res_layers_M1[2].norm)
            M2m1 = self.feed_forward_M2.fused(M2m1, self.res_layers_M2[2].norm)
            return M1m2, M2m1
        main, side = self._fork(M1.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            M2 = self._self_att_M2(M2, M2_mask)
        # (emit_bf16: each self-attention output is the other modality's attention memory -- its bf16 copy leaves the epilogue
        #  that produces it instead of a cast launch on the consumer's chain)
        M1 = self.self_att_M1.fused(M1, None, M1_mask, self.res_layers_M1[0].norm, residual=True, emit_bf16=True)
        main.wait_stream(side)          # both self-attention outputs are needed by both cross attentions
        side.wait_stream(main)
        with torch.cuda.stream(side):
            M2m1 = self.bi_modal_att_M2.fused(M2, M1, M1_mask, self.res_layers_M2[1].norm, residual=True)
            M2m1 = self.feed_forward_M2.fused(M2m1, self.res_layers_M2[2].norm)
        M1m2 = self._cross_M1(M1, M2, M2_mask)
        M1m2 = self.feed_forward_M1.fused(M1m2, self.res_layers_M1[2].norm)
        main.wait_stream(side)
        return M1m2, M2m1


class BMEncoder(nn.Module):
    """reference :218-235"""

    def __init__(self, d_model_M1, d_model_M2, d_model, d_ff_M1, d_ff_M2, dout_p, H, N):
        super().__init__()
        self.encoder = LayerStack(BMEncoderLayer(d_model_M1, d_model_M2, d_model, d_ff_M1, d_ff_M2, dout_p, H), N)

    def forward(self, x, masks):
        V, A = x
        return self.encoder((V, A), (masks['V_mask'], masks['A_mask']))


class BMFusionLayer(nn.Module):
    """reference :54-117.  `feed_forward` is constructed (state-dict keys) but never applied, as in the reference."""

    def __init__(self, d_model_A, d_model_V, d_model_C, d_model, d_ff_c, dout_p, H):
        super().__init__()
        self.res_layer_self_att = ResidualConnection(d_model_C, dout_p)
        self.self_att = MultiheadedAttention(d_model_C, d_model_C, d_model_C, H, dout_p, d_model)
        self.res_layer_enc_att_A = ResidualConnection(d_model_C, dout_p)
        self.res_layer_enc_att_V = ResidualConnection(d_model_C, dout_p)
        self.enc_att_A = MultiheadedAttention(d_model_C, d_model_A, d_model_A, H, dout_p, d_model)
        self.enc_att_V = MultiheadedAttention(d_model_C, d_model_V, d_model_V, H, dout_p, d_model)
        self.feed_forward = PositionwiseFeedForward(d_model_C, d_ff_c, dout_p)
        self.normCA = nn.LayerNorm(d_model_C)
        self.normCV = nn.LayerNorm(d_model_C)
        self.a_v_constant = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x, masks):
        C, memory = x
        Av, Va = memory
        kvc = masks.get('_kv_cache')        # decoding only: per-clip cache of the memory K|V projections (decode.py)
        C = self.self_att.fused(C, None, masks['C_mask'], self.res_layer_self_att.norm, residual=True)
        if C.is_cuda and self.branch_side_stream:
            # the audio- and video-memory attentions both start from the same C: parallel branches up to the gate
            main = torch.cuda.current_stream()
            cls = BMFusionLayer
            if cls._side is None or cls._side.device != C.device:
                cls._side = torch.cuda.Stream(device=C.device)
            side = cls._side
            side.wait_stream(main)
            with torch.cuda.stream(side):
                Ca = self._memory_att(self.enc_att_A, C, Av, masks['A_mask'], self.res_layer_enc_att_A.norm, kvc)
            Cv = self._memory_att(self.enc_att_V, C, Va, masks['V_mask'], self.res_layer_enc_att_V.norm, kvc)
            main.wait_stream(side)
            return self._tail(Cv, Ca), memory
        Ca = self._memory_att(self.enc_att_A, C, Av, masks['A_mask'], self.res_layer_enc_att_A.norm, kvc)
        Cv = self._memory_att(self.enc_att_V, C, Va, masks['V_mask'], self.res_layer_enc_att_V.norm, kvc)
        return self._tail(Cv, Ca), memory

    branch_side_stream = True
    _side = None
    absorb_memory_projections = True
    # normCA, normCV and the gate as one launch (functional.FusionTailFn); its parameter gradients are per-block atomics, so
    # BMHRL_DETERMINISTIC takes the separate LayerNorm / gate kernels (ordered sums)
    _fused_tail_env = os.environ.get("BMHRL_FUSED_TAIL", "1") == "1"

    class _FusedTail:            # (read lazily: the deterministic switch is the library's reading of the variable, ops.deterministic)
        def __get__(self, obj, owner):
            from .. import ops
            return owner._fused_tail_env and not ops.deterministic()

    fused_tail = _FusedTail()

    def _tail_params(self):
        return (self.normCA.weight, self.normCA.bias, self.normCV.weight, self.normCV.bias, self.a_v_constant)

    def _tail(self, Cv, Ca):
        """reference :107-114"""
        if self.fused_tail and Cv.is_cuda and Cv.shape[-1] <= 512:
            return FusionTailFn.apply(Cv, Ca, 1, False, *self._tail_params())
        Ca = LayerNormFn.apply(Ca, self.normCA.weight, self.normCA.bias)
        Cv = LayerNormFn.apply(Cv, self.normCV.weight, self.normCV.bias)
        return GateFn.apply(Cv, Ca, self.a_v_constant)

    def _memory_att(self, att, C, mem, mask, norm, kv_cache):
        """30 caption positions against a 256- / 800-long memory: the K/V projections are folded into the query side
        (functional.MemAttnFn) instead of projecting the whole memory in every layer of both stacks.  Decoding keeps the
        projected form: there the projection is computed once per clip and reused for every token (kv_cache); so does
        every other no-grad call, which keeps memoised and re-run decoding bit-identical."""
        if kv_cache is None and self.absorb_memory_projections and C.is_cuda and torch.is_grad_enabled():
            return att.fused_memory(C, mem, mask, norm)
        return att.fused(C, mem, mask, norm, residual=True, kv_cache=kv_cache)


def _att_params(att, norm):
    return (norm.weight, norm.bias) + att._params()


def fusion_pair(fw, fm, C, Av, Va, masks):
    """Both fusion stacks (reference :523,528: same layers, different weights, same inputs) layer by layer with every
    product of the two stacks in one launch -- see functional.PairMemAttnFn.  Same arithmetic as BMFusion.forward on each
    stack; returns (worker features, manager features)."""
    from ..functional import FusionTailFn, PairGateFn, PairMemAttnFn, PairRowFn, PairSelfAttnFn
    C2 = torch.stack([C, C])                                     # (2, B, L, d_caps): stack 0 = worker, 1 = manager
    if '_pair' in masks:                                         # (the trainer builds the doubled masks with the single ones)
        cm2, am2, vm2 = masks['_pair']
    else:
        cm2, am2, vm2 = (torch.cat([masks[k], masks[k]]) for k in ('C_mask', 'A_mask', 'V_mask'))
    fused_tail = BMFusionLayer.fused_tail and C.shape[-1] <= 512
    n_layers = len(fw.decoder.layers)
    for li, (lw, lm) in enumerate(zip(fw.decoder.layers, fm.decoder.layers)):
        H = lw.self_att.H
        p = lw.self_att.dout_p if lw.training else 0.0
        C2 = PairSelfAttnFn.apply(C2, cm2, H, p, *_att_params(lw.self_att, lw.res_layer_self_att.norm),
                                  *_att_params(lm.self_att, lm.res_layer_self_att.norm))
        side = None
        if BMFusionLayer.branch_side_stream:                     # audio- and video-memory attentions as parallel branches
            main = torch.cuda.current_stream()
            side = BMFusionLayer._side
            if side is None or side.device != C.device:
                side = BMFusionLayer._side = torch.cuda.Stream(device=C.device)
            side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
            Ca2 = PairMemAttnFn.apply(C2, Av, am2, H, p, *_att_params(lw.enc_att_A, lw.res_layer_enc_att_A.norm),
                                      *_att_params(lm.enc_att_A, lm.res_layer_enc_att_A.norm))
            if not fused_tail:
                Ca2 = PairRowFn.apply(Ca2, lw.normCA.weight, lw.normCA.bias, lm.normCA.weight, lm.normCA.bias)
        Cv2 = PairMemAttnFn.apply(C2, Va, vm2, H, p, *_att_params(lw.enc_att_V, lw.res_layer_enc_att_V.norm),
                                  *_att_params(lm.enc_att_V, lm.res_layer_enc_att_V.norm))
        if not fused_tail:
            Cv2 = PairRowFn.apply(Cv2, lw.normCV.weight, lw.normCV.bias, lm.normCV.weight, lm.normCV.bias)
        if side is not None:
            main.wait_stream(side)
        if fused_tail:       # both stacks' normCA + normCV + gate: one launch (it was six)
            # (after the last layer the two stacks part: returned as two tensors, see FusionTailFn)
            C2 = FusionTailFn.apply(Cv2, Ca2, 2, li == n_layers - 1, *lw._tail_params(), *lm._tail_params())
        else:
            C2 = PairGateFn.apply(Cv2, Ca2, lw.a_v_constant, lm.a_v_constant)
    return C2[0], C2[1]


class BMFusion(nn.Module):
    """reference :120-130"""

    def __init__(self, d_model_A, d_model_V, d_model_C, d_model, d_ff_c, dout_p, H, N):
        super().__init__()
        self.decoder = LayerStack(BMFusionLayer(d_model_A, d_model_V, d_model_C, d_model, d_ff_c, dout_p, H), N)

    def forward(self, x, masks):
        C, _ = self.decoder(x, masks)
        return C


class SegmentCritic(nn.Module):
    """Frozen LSTM(4) -> AReLU -> GRU(2) -> AReLU -> Linear segment scorer, reference :186-215.
    On the GPU it runs on the fp32 HIP kernels of csrc/critic.hip (score_and_labels: one launch per (layer + time)
    diagonal, both matrices of a cell on the f32-input MFMA, fused score / threshold head) so the int segment labels equal
    the reference's exactly; the
    nn.LSTM / nn.GRU modules only hold the parameters (reference state-dict keys) and serve CPU tensors.
    `cfg.rl_critic_path` is loaded when given; None leaves the (frozen) default initialisation in place."""

    def __init__(self, cfg):
        super().__init__()
        self.name = "SegmentCritic"
        d = cfg.d_model_caps
        self.lstm = nn.LSTM(d, 2 * d, num_layers=4, batch_first=True)
        self.gru = nn.GRU(2 * d, 2 * d, num_layers=2, batch_first=True)
        self.lin = nn.Linear(2 * d, 1)
        self.relu = AReLU()
        self.relu2 = AReLU()
        for p in self.parameters():
            p.requires_grad = False
        path = getattr(cfg, "rl_critic_path", None)
        if path is not None:
            self.load_state_dict(torch.load(path))

    def forward(self, emb):
        with torch.no_grad():
            if emb.is_cuda:
                return self.score_and_labels(emb, 0.0)[0]
            h, _ = self.lstm(emb)          # host tensors (tests of the module wiring): torch's own RNN
            h, _ = self.gru(self.relu(h))
            return self.lin(self.relu2(h))

    wavefront = True       # False: one GEMM + L step launches per layer (the first native form; kept for A/B and tests)
    # time steps a layer trails the one below: W_ih is read once per chunk (ops.rnn_wavefront).  r03 re-measured the step at
    # chunk 1 / 2 / 3 / 5 on one box: 5.634 / 5.637 / 5.65 / 5.651 ms -- the chunk machinery no longer pays; default 1 (the
    # plain wavefront, no projection scratch), the chunked form stays selectable and tested
    wave_chunk = int(os.environ.get("BMHRL_WAVE_CHUNK", "1"))

    def score_and_labels(self, emb, threshold):
        """HIP path, fp32.  Returns (score (B, L, 1), labels (B, L) int32 = sigmoid(score) > threshold).
        The six recurrent layers run as a wavefront over (layer, time) (ops.rnn_wavefront: L + 5 launches, every cell of
        a diagonal in one grid, input projection fused into the cell) instead of layer after layer (6 x (1 + L))."""
        from .. import ops
        B, L, d = emb.shape
        dev = emb.device
        H = self.lstm.hidden_size
        rows = B * L
        x = emb.detach().contiguous().view(rows, d)
        stacks = (("lstm", self.lstm, 4, 4, self.relu), ("gru", self.gru, 3, 2, self.relu2))
        if self.wavefront and d % 4 == 0:
            layers = []
            for kind, rnn, gates, n_layers, act in stacks:
                for l in range(n_layers):
                    last = l == n_layers - 1
                    seq = torch.empty(rows, H, device=dev)
                    layers.append(dict(
                        w_ih=getattr(rnn, f"weight_ih_l{l}"), w_hh=getattr(rnn, f"weight_hh_l{l}"),
                        b_ih=getattr(rnn, f"bias_ih_l{l}"), b_hh=getattr(rnn, f"bias_hh_l{l}"),
                        in_seq=x, in_ld=x.shape[1], in_dim=x.shape[1], gates=gates, seq_out=seq,
                        h=[torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)],
                        c=[torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)] if gates == 4 else None,
                        arelu_alpha=act.alpha if last else None, arelu_beta=act.beta if last else None,
                        xproj=torch.empty(rows, gates * H, device=dev) if self.wave_chunk > 1 else None))
                    x = seq
            ops.rnn_wavefront(layers, B, L, H, chunk=self.wave_chunk)
        else:
            hb = [torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)]
            cb = [torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)]
            for kind, rnn, gates, n_layers, act in stacks:
                for l in range(n_layers):
                    w_ih, w_hh = getattr(rnn, f"weight_ih_l{l}"), getattr(rnn, f"weight_hh_l{l}")
                    b_ih, b_hh = getattr(rnn, f"bias_ih_l{l}"), getattr(rnn, f"bias_hh_l{l}")
                    K = x.shape[1]
                    xproj = torch.empty(rows, gates * H, device=dev)
                    ops.gemm_f32(x, w_ih, b_ih, b_hh if gates == 4 else None, xproj, rows, gates * H, K)
                    seq = torch.empty(rows, H, device=dev)
                    last = l == n_layers - 1
                    for t in range(L):
                        ops.rnn_step(gates, xproj, w_hh, b_hh if gates == 3 else None, hb[(t + 1) & 1], cb[(t + 1) & 1], hb[t & 1],
                                     cb[t & 1] if gates == 4 else None, seq, act.alpha if last else None,
                                     act.beta if last else None, B, L, H, t)
                    x = seq
        score = torch.empty(B, L, 1, device=dev)
        labels = torch.empty(B, L, dtype=torch.int32, device=dev)
        ops.critic_head(x, self.lin.weight, self.lin.bias, float(threshold), score, labels, rows, H)
        return score, labels


class LinearCore(nn.Module):
    """reference :387-396 (registered for its state-dict keys; Manager.forward does not use it, :438)"""

    def __init__(self, d_model_caps, d_goal, dout_p):
        super().__init__()
        self.linear = nn.Linear(d_model_caps, d_goal)
        self.dropout = nn.Dropout(dout_p)


class Manager(nn.Module):
    """reference :399-454: goals = expand_goals(dropout(linear(feat)) [+ exploration noise], segment labels)"""

    def __init__(self, device, d_model_caps, d_goal, dout_p, core=None, exploration=True):
        super().__init__()
        self.device = device
        self.core = core if core is not None else LinearCore(d_model_caps, d_goal, dout_p)
        self.linear = nn.Linear(d_model_caps, d_goal)
        self.dropout = nn.Dropout(dout_p)
        self.dout_p = dout_p
        self.exploration = exploration
        self.d_goal = d_goal
        self.mean_factor = 10
        self.std_factor = 5
        self.last_noise = None

    def forward(self, x, critic_mask):
        p = self.dout_p if self.training else 0.0
        g = LinearFn.apply(x, self.linear.weight, self.linear.bias, False, p)
        if self.exploration:
            # one (d_goal,) Gaussian vector N(mean/10, std/5) - 0.5*mean/10 for all tokens (reference :444-452), drawn from the
            # library's counter RNG inside the expand_goals launch (no torch generator: the captured step stays valid and
            # the vector changes with every replay); last_noise keeps the vector of the latest call for inspection
            if self.last_noise is None or self.last_noise.device != g.device:
                self.last_noise = torch.zeros(self.d_goal, device=g.device)
            return ExpandGoalsFn.apply(g, critic_mask, (float(self.mean_factor), float(self.std_factor)), self.last_noise)
        return ExpandGoalsFn.apply(g, critic_mask)


class WorkerCore(nn.Module):
    """reference :456-466"""

    def __init__(self, voc_size, d_in, d_goal):
        super().__init__()
        self.projection = nn.Linear(d_in + d_goal, voc_size)


class Worker(nn.Module):
    """reference :468-487: log_softmax(projection(cat[x, goal_attention(goal, x, x)]))"""

    def __init__(self, voc_size, d_in, d_goal, dout_p, d_model, core=None):
        super().__init__()
        self.core = core if core is not None else WorkerCore(voc_size, d_in, d_goal)
        self.goal_attention = MultiheadedAttention(d_goal, d_in, d_in, 2, dout_p, d_model)

    def forward(self, x, goal, mask):
        gc = self.goal_attention.fused(goal, x, mask)
        return WorkerHeadFn.apply(x, gc, self.core.projection.weight, self.core.projection.bias)


def _value_head(ffn: PositionwiseFeedForward, projection: nn.Linear, feat):
    p = ffn.dout_p if ffn.training else 0.0
    h = LinearFn.apply(feat, ffn.fc1.weight, ffn.fc1.bias, True, p)
    h = LinearFn.apply(h, ffn.fc2.weight, ffn.fc2.bias, True, 0.0)   # fc2, then the head's ReLU
    return LinearFn.apply(h, projection.weight, projection.bias, False, 0.0)


class BMWorkerValueFunction(ModelBase):
    """reference :251-269: Linear(relu(FFN(worker_feat))); the goal input is ignored."""

    def __init__(self, cfg):
        super().__init__("bm_worker_value_function")
        d = cfg.d_model_caps
        self.value_function = PositionwiseFeedForward(d, d * 2, cfg.dout_p)
        self.projection = nn.Linear(d, 1)
        self.activation = nn.ReLU()

    def forward(self, x):
        w_feat, _ = x
        return _value_head(self.value_function, self.projection, w_feat)


class BMManagerValueFunction(ModelBase):
    """reference :272-286"""

    def __init__(self, cfg):
        super().__init__("bm_manager_value_function")
        d = cfg.d_model_caps
        self.value_function = PositionwiseFeedForward(d, d * 2, cfg.dout_p)
        self.projection = nn.Linear(d, 1)
        self.activation = nn.ReLU()

    def forward(self, x):
        return _value_head(self.value_function, self.projection, x)


class BMHrlAgent(nn.Module):
    """reference :491-661"""

    def __init__(self, cfg, train_dataset):
        super().__init__()
        self.name = "bm_hrl_agent"
        self.d_video, self.d_audio = cfg.d_vid, cfg.d_aud
        self.d_proj = getattr(cfg, "rl_projection_d", None)
        self.d_model_caps, self.d_model = cfg.d_model_caps, cfg.d_model
        self.att_heads, self.att_layers = cfg.rl_att_heads, cfg.rl_att_layers
        self.dout_p = cfg.dout_p
        self.d_goal = cfg.rl_goal_d
        self.voc_size = train_dataset.trg_voc_size
        self.device = torch.device(cfg.device)
        self.critic_score_threshhold = cfg.rl_critic_score_threshhold

        self.pos_enc_A = PositionalEncoder(cfg.d_model_audio, cfg.dout_p)
        self.pos_enc_V = PositionalEncoder(cfg.d_model_video, cfg.dout_p)
        self.pos_enc_C = PositionalEncoder(cfg.d_model_caps, cfg.dout_p)
        self.critic = SegmentCritic(cfg)
        self.emb_C = VocabularyEmbedder(train_dataset.trg_voc_size, cfg.d_model_caps)
        self.emb_C.init_word_embeddings(train_dataset.train_vocab.vectors, cfg.unfreeze_word_emb)
        self.bm_enc = BMEncoder(self.d_video, self.d_audio, self.d_model, cfg.rl_ff_v, cfg.rl_ff_a, self.dout_p,
                                self.att_heads, self.att_layers)
        self.bm_worker_fus = BMFusion(cfg.d_model_audio, cfg.d_model_video, cfg.d_model_caps, cfg.d_model, cfg.rl_ff_c,
                                      self.dout_p, self.att_heads, self.att_layers)
        self.bm_manager_fus = BMFusion(cfg.d_model_audio, cfg.d_model_video, cfg.d_model_caps, cfg.d_model, cfg.rl_ff_c,
                                       self.dout_p, self.att_heads, self.att_layers)
        self.manager_core = LinearCore(cfg.d_model_caps, cfg.rl_goal_d, cfg.dout_p)
        self.manager = Manager(self.device, self.d_model_caps, self.d_goal, self.dout_p, self.manager_core)
        self.worker = Worker(self.voc_size, self.d_model_caps, self.d_goal, self.dout_p, self.d_model)

        self.teach_warmstart()
        self.warmstarting = True
        self.teaching_worker = True
        self.sigmoid_epoch_offset = torch.tensor(-1)
        self.worker_modules = [self.bm_enc, self.bm_worker_fus, self.worker]
        self.manager_modules = [self.bm_manager_fus, self.manager]

    # ---- checkpoints (reference :547-553)
    def save_model(self, checkpoint_dir):
        torch.save(self.state_dict(), checkpoint_dir + f"/{self.name}.pt")

    def load_model(self, checkpoint_dir):
        self.load_state_dict(torch.load(checkpoint_dir + f"/{self.name}.pt"))

    # ---- phase switches (reference :555-593)
    @staticmethod
    def _set_module_grads(modules, enable):
        for m in modules:
            for p in m.parameters():
                p.requires_grad = enable

    def teach_warmstart(self):
        self.warmstarting = True
        self._set_module_grads([self.worker, self.bm_worker_fus, self.manager, self.bm_manager_fus], True)

    def teach_worker(self):
        self.warmstarting = False
        self.teaching_worker = True
        self._set_module_grads(self.worker_modules, True)
        self._set_module_grads(self.manager_modules, False)
        self.manager.exploration = False

    def teach_manager(self):
        self.warmstarting = False
        self.teaching_worker = False
        self._set_module_grads(self.worker_modules, False)
        self._set_module_grads(self.manager_modules, True)
        self.manager.exploration = True

    def set_inference_mode(self, inference):
        self.manager.exploration = not inference

    # ---- forward (reference :596-661)
    def warmstart(self, x, trg, mask):
        return self.prediction(x, trg, mask)

    critic_side_stream = True
    # training: layer l of the worker and of the manager fusion stack run as ONE set of launches (functional.PairMemAttnFn)
    pair_fusion_stacks = os.environ.get("BMHRL_PAIR_STACKS", "1") == "1"

    _critic_streams = {}

    def _side_stream(self, device):
        """ONE critic stream per device for the whole process (like the layers' fork streams): torch hands streams out of a
        pool of 32 per device, round robin -- a stream per agent would, after enough agents, be the very stream another fork
        (or the graph capture itself) runs on."""
        st = BMHrlAgent._critic_streams.get(device)
        if st is None:
            st = BMHrlAgent._critic_streams[device] = torch.cuda.Stream(device=device)
        return st

    def _segment_labels(self, emb):
        if emb.is_cuda:
            # same values as (sigmoid(critic(C)) > thr).squeeze().int() of the reference (:638-640); the squeeze only
            # changes the shape when B == 1 or L == 1
            return self.critic.score_and_labels(emb, self.critic_score_threshhold)[1].squeeze()
        seg = torch.sigmoid(self.critic(emb))
        return (seg > self.critic_score_threshhold).squeeze().int()

    def _encode(self, x):
        """x = (V, A) or ((rgb, flow), A): the rgb + flow add is fused into the positional-encoding kernel (K1)."""
        xv, xa = x
        V = self.pos_enc_V(xv[0], xv[1]) if isinstance(xv, (tuple, list)) else self.pos_enc_V(xv)
        return V, self.pos_enc_A(xa)

    def prediction(self, x, trg, mask):
        emb, C = self.emb_C.embed_posenc(trg, self.pos_enc_C)
        return self.predict_with_features(emb, None, None, mask, C, x=x)      # (the critic is forked before x is encoded)

    def mixed_prediction(self, x, trgs, mask, mix_factor):
        y_trg, yhat_trg = trgs
        emb, C = self.emb_C.embed_posenc(y_trg, self.pos_enc_C, yhat_trg, float(mix_factor))
        return self.predict_with_features(emb, None, None, mask, C, x=x)

    # ---- memoised decoding.  The reference's greedy decoders re-run the whole agent -- encoder
    # included -- for every generated token (epoch_loops/captioning_bmrl_loops.py:61-76,127-152); the encoder output and
    # the fusion layers' K|V projections of it depend on the clip only, so they are computed once.
    def encode_memory(self, x, mask):
        """Encoder output (Va, Av) of a clip batch; x as in forward(), mask needs V_mask / A_mask."""
        V, A = self._encode(x)
        return self.bm_enc((V, A), mask)

    def inference_from_memory(self, memory, trg, mask, kv_cache=None):
        """log-probs (B, L, V) for the captions `trg` given encode_memory()'s result -- same values as inference()."""
        emb, C = self.emb_C.embed_posenc(trg, self.pos_enc_C)
        if kv_cache is not None:
            mask = dict(mask)
            mask['_kv_cache'] = kv_cache
        return self.predict_with_features(emb, None, None, mask, C, memory=memory)[0]

    def predict_with_features(self, C_emb, V, A, mask, C=None, memory=None, x=None):
        # The frozen critic only feeds the segment labels the manager needs at the very end: it runs on a side HIP
        # stream (a parallel branch of the captured graph) next to the encoder / fusion kernels.
        side = None
        if C_emb.is_cuda and self.critic_side_stream:
            main = torch.cuda.current_stream()
            side = self._side_stream(C_emb.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                segment_labels = self._segment_labels(C_emb)
            C_emb.record_stream(side)
        else:
            segment_labels = self._segment_labels(C_emb)
        if C is None:
            C = self.pos_enc_C(C_emb)
        if x is not None:                      # raw features: their positional encoding runs after the critic's fork -- the
            V, A = self._encode(x)             # critic's 45 dependent launches are the longest chain of the forward
        # Va: video stream (B,Tv,d_vid), Av: audio stream (B,Ta,d_aud)
        Va, Av = self.bm_enc((V, A), mask) if memory is None else memory
        # (worker and manager stacks as two parallel branches were measured slower than back to back: 10.8 -> 11.4 ms/step;
        # the branches that pay off are inside the layers: BMEncoderLayer.modality_side_stream, BMFusionLayer.branch_side_stream)
        if self.pair_fusion_stacks and C.is_cuda and torch.is_grad_enabled():
            worker_feat, manager_feat = fusion_pair(self.bm_worker_fus, self.bm_manager_fus, C, Av, Va, mask)
        else:
            worker_feat = self.bm_worker_fus((C, (Av, Va)), mask)
            manager_feat = self.bm_manager_fus((C, (Av, Va)), mask)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        goals = self.manager(manager_feat, segment_labels)
        pred = self.worker(worker_feat, goals, mask["C_mask"])
        return pred, worker_feat, manager_feat, goals, segment_labels

    def inference(self, x, trg, mask, *hidden):
        """3-argument reference form returns the log-probs; the 5-argument call sites of the reference's greedy
        decoders (epoch_loops/captioning_bmrl_loops.py:72,147) get (log-probs, None, None)."""
        pred = self.prediction(x, trg, mask)[0]
        return (pred, None, None) if hidden else pred

    def forward(self, x, trg, mask, factor=1):
        if type(trg) is tuple:
            return self.mixed_prediction(x, trg, mask, factor)
        return self.prediction(x, trg, mask)


class _OutOfHotPath(nn.Module):
    """Names of the reference's model/bm_hrl_agent.py that its driver and model/det_bmhrl_agent.py import but that belong to
    other model families (the unimodal AHRL / VHRL ablation agents, reference :40-51, :133-183, :238-248, :289-325,
    :664-809; SURVEY.md section 2 row 3: OUT OF SCOPE).  They exist so that `from model.bm_hrl_agent import ...` of the
    reference's own files resolves after `import bmhrl_amd.install`; constructing one raises."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__} (unimodal / projection variants of the reference) is outside the bimodal hot path "
            "this package replaces; use the reference's own class for it")


class ModalityProjection(_OutOfHotPath):
    pass


class UnimodalFusion(_OutOfHotPath):
    pass


class UnimodalFusionLayer(_OutOfHotPath):
    pass


class UnimodalEncoder(_OutOfHotPath):
    pass


class UnimodalEncoderLayer(_OutOfHotPath):
    pass


class UnimodalAgent(_OutOfHotPath):
    pass


class AudioAgent(UnimodalAgent):
    pass


class VideoAgent(UnimodalAgent):
    pass


def agent_state_shapes(cfg, voc_size, with_critic=True):
    """name -> shape of the agent's state dict without building tensors (meta device)."""
    from types import SimpleNamespace
    ds = SimpleNamespace(trg_voc_size=voc_size, train_vocab=SimpleNamespace(vectors=None))
    saved = getattr(cfg, "rl_critic_path", None)
    cfg.rl_critic_path = None
    try:
        with torch.device("meta"):
            m = BMHrlAgent(cfg, ds)
    finally:
        cfg.rl_critic_path = saved
    return {k: tuple(v.shape) for k, v in m.state_dict().items() if with_critic or not k.startswith("critic.")}
