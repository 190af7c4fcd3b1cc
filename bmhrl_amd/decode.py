"""Greedy decoding with the reference's decoder signature (epoch_loops/captioning_bmrl_loops.py:41-43,127-152):
arg-max autoregressive decode that stops when every sample has produced </s> or at max_len.

Three forms, same tokens:

  * memoise=False -- the reference's own schedule: the whole agent (encoder included) re-run for every token;
  * memoise=True  -- encoder output and the fusion layers' memory K|V projections computed once per clip batch, the caption
    side re-run over the whole prefix for every token;
  * incremental   -- IncrementalDecoder below: on top of the per-clip memory K|V, every caption-side
    layer keeps the K|V rows of the tokens decoded so far, the frozen critic carries its LSTM / GRU state, and a token costs
    one row per sample through both fusion stacks, the manager and the worker.  All shapes of a token step are static, the
    position is a device word, so the step is captured once into a HIP graph and replayed max_len times.
"""
import math

import torch

from . import ops
from .functional import SHADOWS, ExpandGoalsFn, GateFn, LayerNormFn, LinearFn, WorkerHeadFn, _attn_core_fwd, pad8
from .model.masking import make_masks

_BF16 = torch.bfloat16


def greedy_decode(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality, return_first=False, memoise=True,
                  incremental=None):
    """memoise=True (default) runs the encoder and the fusion layers' memory projections once per clip batch instead of
    once per generated token; the tokens and log-probs are the same as with the reference's full re-run (memoise=False).
    incremental (default: on for the HIP agent in eval mode on a GPU with both modalities) additionally decodes through
    IncrementalDecoder.  Models without encode_memory() (anything but the HIP BMHrlAgent) always take the full re-run."""
    with torch.no_grad():
        B = feature_stacks['audio'].shape[0]
        device = feature_stacks['audio'].device
        memoise = memoise and hasattr(model, "encode_memory") and not model.training
        if incremental is None:
            incremental = IncrementalDecoder.enabled
        if incremental and memoise and device.type == "cuda" and modality == "audio_video" and max_len >= 1:
            dec = IncrementalDecoder.for_batch(model, feature_stacks, max_len, start_idx, end_idx, pad_idx)
            if dec.begin(feature_stacks):
                return dec.run(return_first)
        done = torch.zeros(B, 1, dtype=torch.bool, device=device)
        trg = torch.full((B, 1), start_idx, dtype=torch.long, device=device)
        first = None
        x = ((feature_stacks['rgb'], feature_stacks['flow']), feature_stacks['audio'])
        memory, kv_cache = None, {}
        while trg.size(-1) <= max_len and not bool(done.all()):
            masks = make_masks(feature_stacks, trg, modality, pad_idx)
            if memoise:
                if memory is None:
                    memory = model.encode_memory(x, masks)
                preds = model.inference_from_memory(memory, trg, masks, kv_cache)
            else:
                preds = model.inference(x, trg, masks)
            if first is None:
                first = preds[:, -1].clone()
            nxt = preds[:, -1].argmax(dim=-1, keepdim=True)
            trg = torch.cat([trg, nxt], dim=-1)
            done = done | (nxt == end_idx)
    return (trg, first) if return_first else trg


def bimodal_decoder(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality):
    return greedy_decode(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality)


bmhrl_greedy_decoder = bimodal_decoder


class IncrementalDecoder:
    """One decoded token = one row per sample through the caption side of BMHrlAgent (model/bm_hrl_agent.py:596-661),
    with everything that does not depend on the newest token kept in HBM between tokens:

      mem_kv[stack][layer][A|V]  (B, cap, 2*d_model) bf16   K|V projections of the encoder output (per clip)
      self_kv[stack][layer]      (B, Lc, 2*d_model)  bf16   K|V rows of the caption self attention (one row appended per token)
      goal_kv                    (B, Lc, 2*d_model)  bf16   K|V rows of the worker's goal attention over the worker features
      critic h / c               6 x (B, 600) fp32          LSTM(4) / GRU(2) state of the frozen segment critic
      labels (B, Lc) int32, goals_raw (B, Lc, d_goal) fp32  the manager's inputs to expand_goals (which reads whole rows)

    The fusion stack is causal (C_mask = key padding & lower triangle, model/masking.py:13-15), the critic is a forward
    recurrence and the memory attentions / LayerNorms / gate act per position, so position t of every intermediate only
    depends on tokens <= t: appending rows reproduces the full re-run.  expand_goals is not causal (a segment's last goal is
    copied back over the segment, with the row-transition quirks of the reference's loop, :415-429); it runs over the whole
    (B, Lc) label buffer every token -- positions after t carry label 0, which is what the re-run sees as "no later token".

    Static shapes (memory padded to a capacity with masked tails, Lc = padded max_len + 1, the position a device word)
    make the token step one HIP graph.  begin() returns False when a sample has no valid memory key at all (softmax of a
    fully masked row is uniform over the UNPADDED keys in the reference, :22) and the caller falls back."""

    enabled = True
    use_graph = True
    check_every = 4             # host looks at `done` every this many tokens (the result is trimmed to the exact length)
    _cache_attr = "_incremental_decoders"

    @classmethod
    def for_batch(cls, agent, fs, max_len, start_idx, end_idx, pad_idx):
        B, Tv = fs['rgb'].shape[:2]
        Ta = fs['audio'].shape[1]
        key = (B, -(-Tv // 64) * 64, -(-Ta // 64) * 64, int(max_len), int(start_idx), int(end_idx), int(pad_idx), fs['rgb'].device)
        cache = agent.__dict__.setdefault(cls._cache_attr, {})
        dec = cache.get(key)
        if dec is None:
            if len(cache) >= 8:                         # shapes of a validation set fall into a few capacity buckets
                cache.pop(next(iter(cache)))
            dec = cache[key] = cls(agent, *key[:7], device=key[7])
        return dec

    def __init__(self, agent, B, tv_cap, ta_cap, max_len, start_idx, end_idx, pad_idx, device):
        self.agent = agent
        self.B, self.tv_cap, self.ta_cap, self.max_len = B, tv_cap, ta_cap, max_len
        self.start_idx, self.end_idx, self.pad_idx = start_idx, end_idx, pad_idx
        self.dev = dev = torch.device(device)
        self.dC, self.D = agent.d_model_caps, agent.d_model
        self.Lc = Lc = pad8(max_len + 1)
        D, dC = self.D, self.dC
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
        self.t = z(1, dtype=torch.int64)
        self.tok = z(B, dtype=torch.int64)
        self.out = z(B, max_len + 1, dtype=torch.int64)
        self.done = z(B, dtype=torch.bool)
        self.valid = z(B, 1, Lc, dtype=torch.uint8)
        self.a_mask = z(B, 1, ta_cap, dtype=torch.uint8)
        self.v_mask = z(B, 1, tv_cap, dtype=torch.uint8)
        self.labels = z(B, Lc, dtype=torch.int32)
        self.goals_raw = z(B, Lc, agent.d_goal)
        self.logp = z(B, 1, agent.voc_size)
        self.emb_scale = math.sqrt(dC)
        self.pe = agent.pos_enc_C.table(dev)
        n_layers = len(agent.bm_worker_fus.decoder.layers)
        mk = lambda rows: z(B, rows, 2 * D, dtype=_BF16)
        self.stacks = []
        for fus in (agent.bm_worker_fus, agent.bm_manager_fus):
            self.stacks.append(dict(fus=fus, self_kv=[mk(Lc) for _ in range(n_layers)],
                                    mem_a=[mk(ta_cap) for _ in range(n_layers)], mem_v=[mk(tv_cap) for _ in range(n_layers)]))
        self.goal_kv = z(B, Lc, 2 * agent.worker.goal_attention.d_model, dtype=_BF16)
        cr = agent.critic
        Hc = cr.lstm.hidden_size
        self.critic_layers = []
        for rnn, gates, n, act in ((cr.lstm, 4, 4, cr.relu), (cr.gru, 3, 2, cr.relu2)):
            for l in range(n):
                self.critic_layers.append(dict(
                    gates=gates, w_ih=getattr(rnn, f"weight_ih_l{l}"), w_hh=getattr(rnn, f"weight_hh_l{l}"),
                    b_ih=getattr(rnn, f"bias_ih_l{l}"), b_hh=getattr(rnn, f"bias_hh_l{l}"),
                    act=act if l == n - 1 else None, h=z(B, Hc), c=z(B, Hc), h_new=z(B, Hc), c_new=z(B, Hc),
                    # the step kernel reads the carried state only at positions > 0 of its window: the newest token is
                    # position 1 of a two-position window whose position 0 is never touched
                    xproj=z(B, 2, gates * Hc), seq=z(B, 2, Hc)))
        self.labels2 = z(B, 2, dtype=torch.int32)
        self.Hc = Hc
        self.graph = None
        self._shadow_sig = None
        self.steps_run = 0
        with torch.no_grad():
            self._reset()
            self._token_step()                      # eager once: weight shadows, allocator warm-up
            if self.use_graph:
                self._capture()
            self._reset()

    # ------------------------------------------------------------------ per clip
    def _reset(self):
        self.t.zero_()
        self.tok.fill_(self.start_idx)
        self.out.fill_(self.pad_idx)
        self.out[:, 0] = self.start_idx
        self.done.zero_()
        self.valid.zero_()
        self.labels.zero_()
        self.goals_raw.zero_()
        torch._foreach_zero_([l[k] for l in self.critic_layers for k in ("h", "c")])
        self.steps_run = 0

    def begin(self, fs) -> bool:
        """encoder + memory K|V projections of a clip batch into the static buffers; False = take the re-run path"""
        ag, B, D = self.agent, self.B, self.D
        masks = make_masks(fs, None, "audio_video", self.pad_idx)
        am, vm = masks['A_mask'], masks['V_mask']
        if not bool((am.any(-1) & vm.any(-1)).all()):
            return False
        Ta, Tv = am.shape[-1], vm.shape[-1]
        self.a_mask.zero_(); self.a_mask[:, :, :Ta] = am
        self.v_mask.zero_(); self.v_mask[:, :, :Tv] = vm
        x = ((fs['rgb'], fs['flow']), fs['audio'])
        Va, Av = ag.encode_memory(x, masks)                    # (B, Tv, d_vid), (B, Ta, d_aud)
        for mem, T, cap, name, att_name in ((Av, Ta, self.ta_cap, "mem_a", "enc_att_A"), (Va, Tv, self.tv_cap, "mem_v", "enc_att_V")):
            d = mem.shape[-1]
            mb = torch.zeros(B, cap, pad8(d), dtype=_BF16, device=self.dev)
            mb[:, :T, :d] = mem
            for st in self.stacks:
                for li, layer in enumerate(st["fus"].decoder.layers):
                    att = getattr(layer, att_name)
                    w_kv = SHADOWS.weight(att.linear_K2d.weight, att.linear_V2d.weight)
                    ops.gemm(mb, w_kv, B * cap, 2 * D, d, lda=mb.shape[-1], ldb=w_kv.shape[1], C_bf16=st[name][li], ldcb=2 * D,
                             bias=SHADOWS.bias(att.linear_K2d.bias, att.linear_V2d.bias))
        self._reset()
        if self.graph is not None and self._shadows() != self._shadow_sig:
            self._capture()                # weights were re-materialised since the capture (new shadow buffers)
        return True

    def run(self, return_first=False):
        first = None
        for i in range(self.max_len):
            self.step()
            if i == 0 and return_first:
                first = self.logp[:, 0].clone()
            if (i + 1) % self.check_every == 0 and i + 1 < self.max_len and bool(self.done.all()):
                break
        trg = self.result()
        return (trg, first) if return_first else trg

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._token_step()
        self.steps_run += 1

    def result(self):
        """tokens up to the step at which every sample had produced </s> (the reference's loop condition, :61-76)"""
        n = self.steps_run
        out = self.out[:, :n + 1]
        is_end = out[:, 1:] == self.end_idx
        if n and bool(is_end.any(1).all()):
            n = int(is_end.float().argmax(1).max()) + 1
        return out[:, :n + 1].clone()

    # ------------------------------------------------------------------ graph
    def _weights(self):
        """(weight groups, bias groups) whose bf16 / concatenated shadows the token step reads"""
        ag = self.agent
        ws, bs = [], []
        for st in self.stacks:
            for layer in st["fus"].decoder.layers:
                a = layer.self_att
                ws += [(a.linear_Q2d.weight, a.linear_K2d.weight, a.linear_V2d.weight), (a.linear_d2Q.weight,)]
                bs += [(a.linear_Q2d.bias, a.linear_K2d.bias, a.linear_V2d.bias)]
                for m in (layer.enc_att_A, layer.enc_att_V):
                    ws += [(m.linear_Q2d.weight,), (m.linear_d2Q.weight,), (m.linear_K2d.weight, m.linear_V2d.weight)]
                    bs += [(m.linear_K2d.bias, m.linear_V2d.bias)]
        g = ag.worker.goal_attention
        ws += [(g.linear_Q2d.weight,), (g.linear_K2d.weight, g.linear_V2d.weight), (g.linear_d2Q.weight,),
               (ag.manager.linear.weight,)]
        bs += [(g.linear_K2d.bias, g.linear_V2d.bias)]
        return ws, bs

    def _shadows(self):
        """refreshes the shadows the captured kernels read (bf16 weights are re-cast in place when a parameter changed,
        concatenated biases are rebuilt) and returns their addresses: a different address means the graph holds a dead
        pointer and is captured again"""
        ws, bs = self._weights()
        # (+ the parameters the step reads directly -- LayerNorm, biases, embedding table, critic weights: their storages
        # move when the module is re-materialised, e.g. by .to(device) or by a trainer that re-points them into its bucket)
        return tuple(SHADOWS.weight(*w).data_ptr() for w in ws) + tuple(SHADOWS.bias(*b).data_ptr() for b in bs) + \
            (SHADOWS.weight_split3(self.agent.worker.core.projection.weight).data_ptr(),) + \
            tuple(p.data_ptr() for p in self.agent.parameters())

    def _capture(self):
        self._shadow_sig = self._shadows()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._token_step()
        self.graph = g

    # ------------------------------------------------------------------ one token
    def _attend(self, att, norm, x, kv, mask, Sk, residual, append):
        """x (B, dq) fp32 -> [x +] d2Q(attention(Q2d(LN?(x)), K|V rows in `kv`)) (B, dq) fp32.  append: this is a self
        attention -- the K|V projection of the (normalised) row is written at position t of `kv` first."""
        B, dev = self.B, self.dev
        D, H = att.d_model, att.H
        dq = x.shape[1]
        ldx = pad8(dq)
        xb = ops.bf16_zeros(B, dq, dev)
        if norm is not None:
            ops.layernorm_fwd(x, norm.weight.detach(), norm.bias.detach(), xb, ldx, None, None, None, B, dq)
        else:
            ops.cast_bf16(x, dq, xb, ldx, B, dq)
        if append is not None:
            w = SHADOWS.weight(att.linear_Q2d.weight, *append[0])
            n_out = w.shape[0]
            bias = SHADOWS.bias(att.linear_Q2d.bias, *append[1])
            q = torch.empty(B, n_out, dtype=_BF16, device=dev)
            ops.gemm(xb, w, B, n_out, dq, lda=ldx, ldb=w.shape[1], C_bf16=q, ldcb=n_out, bias=bias)
            kv.index_copy_(1, self.t, q[:, D:].unsqueeze(1))
            ldq = n_out
        else:
            w = SHADOWS.weight(att.linear_Q2d.weight)
            q = torch.empty(B, D, dtype=_BF16, device=dev)
            ops.gemm(xb, w, B, D, dq, lda=ldx, ldb=w.shape[1], C_bf16=q, ldcb=D, bias=att.linear_Q2d.bias.detach())
            ldq = D
        o, _ = _attn_core_fwd(q, 0, ldq, kv, 0, 2 * D, kv, D, 2 * D, mask, Sk, 0, B, H, 1, Sk, D // H, 0.0, 0)
        w_o = SHADOWS.weight(att.linear_d2Q.weight)
        y = torch.empty(B, dq, device=dev)
        ops.gemm(o, w_o, B, dq, D, lda=D, ldb=w_o.shape[1], C_f32=y, ldc=dq, bias=att.linear_d2Q.bias.detach(),
                 residual=x if residual else None, ldr=dq)
        return y

    def _critic_step(self, emb):
        B, Hc = self.B, self.Hc
        x = emb
        for l in self.critic_layers:
            g = l["gates"]
            ops.gemm_f32(x, l["w_ih"], l["b_ih"], l["b_hh"] if g == 4 else None, l["xproj"][:, 1], B, g * Hc, x.shape[1])
            act = l["act"]
            ops.rnn_step(g, l["xproj"], l["w_hh"], l["b_hh"] if g == 3 else None, l["h"], l["c"], l["h_new"],
                         l["c_new"] if g == 4 else None, l["seq"], act.alpha if act is not None else None,
                         act.beta if act is not None else None, B, 2, Hc, 1)
            x = l["seq"][:, 1]
        torch._foreach_copy_([l[k] for l in self.critic_layers for k in ("h", "c")],
                             [l[k] for l in self.critic_layers for k in ("h_new", "c_new")])
        cr = self.agent.critic
        ops.critic_head(self.critic_layers[-1]["seq"], cr.lin.weight, cr.lin.bias, float(self.agent.critic_score_threshhold),
                        None, self.labels2, 2 * B, Hc)
        self.labels.index_copy_(1, self.t, self.labels2[:, 1:2])

    def _token_step(self):
        ag, B, t = self.agent, self.B, self.t
        emb = ag.emb_C.embedder.weight.detach().index_select(0, self.tok) * self.emb_scale     # (B, dC): the critic's input
        C0 = emb + self.pe.index_select(0, t)
        self.valid.index_copy_(2, t, (self.tok != self.pad_idx).to(torch.uint8).view(B, 1, 1))
        self._critic_step(emb.contiguous())
        feats = []
        for st in self.stacks:
            C = C0
            for li, layer in enumerate(st["fus"].decoder.layers):
                a = layer.self_att
                C = self._attend(a, layer.res_layer_self_att.norm, C, st["self_kv"][li], self.valid, self.Lc, True,
                                 ((a.linear_K2d.weight, a.linear_V2d.weight), (a.linear_K2d.bias, a.linear_V2d.bias)))
                Ca = self._attend(layer.enc_att_A, layer.res_layer_enc_att_A.norm, C, st["mem_a"][li], self.a_mask, self.ta_cap,
                                  True, None)
                Cv = self._attend(layer.enc_att_V, layer.res_layer_enc_att_V.norm, C, st["mem_v"][li], self.v_mask, self.tv_cap,
                                  True, None)
                C = layer._tail(Cv, Ca)            # normCA, normCV, gate: the kernel of the full forward (bit-identical)
            feats.append(C)
        w_feat, m_feat = feats
        # manager (:437-454, exploration off while decoding): the newest raw goal joins the buffer, expand_goals reads whole rows
        g = LinearFn.apply(m_feat, ag.manager.linear.weight, ag.manager.linear.bias, False, 0.0)
        self.goals_raw.index_copy_(1, t, g.unsqueeze(1))
        goal = ExpandGoalsFn.apply(self.goals_raw, self.labels).index_select(1, t).squeeze(1)      # (B, d_goal)
        # worker (:480-487): goal attention over the worker features decoded so far, then the vocabulary head
        ga = ag.worker.goal_attention
        xb = ops.bf16_zeros(B, self.dC, self.dev)
        ops.cast_bf16(w_feat, self.dC, xb, xb.shape[1], B, self.dC)
        w_kv = SHADOWS.weight(ga.linear_K2d.weight, ga.linear_V2d.weight)
        kv_t = torch.empty(B, w_kv.shape[0], dtype=_BF16, device=self.dev)
        ops.gemm(xb, w_kv, B, w_kv.shape[0], self.dC, lda=xb.shape[1], ldb=w_kv.shape[1], C_bf16=kv_t, ldcb=w_kv.shape[0],
                 bias=SHADOWS.bias(ga.linear_K2d.bias, ga.linear_V2d.bias))
        self.goal_kv.index_copy_(1, t, kv_t.unsqueeze(1))
        gc = self._attend(ga, None, goal.contiguous(), self.goal_kv, self.valid, self.Lc, False, None)
        logp = WorkerHeadFn.apply(w_feat.view(B, 1, -1), gc.view(B, 1, -1), ag.worker.core.projection.weight,
                                  ag.worker.core.projection.bias)
        self.logp.copy_(logp)
        nxt = logp.view(B, -1).argmax(-1)
        self.out.index_copy_(1, t + 1, nxt.view(B, 1))
        self.done.logical_or_(nxt == self.end_idx)
        self.tok.copy_(nxt)
        t.add_(1)
