"""MultiheadedAttention with the reference's constructor, attributes and state-dict keys
(model/multihead_attention.py:34-92); forward runs the fused HIP path of bmhrl_amd.functional.MHAFn."""
import torch
import torch.nn as nn

from ..functional import AttnCoreFn, LinearFn, MemAttnFn, MHAFn


class MultiheadedAttention(nn.Module):

    def __init__(self, d_model_Q, d_model_K, d_model_V, H, dout_p=0.0, d_model=None):
        super().__init__()
        self.d_model_Q, self.d_model_K, self.d_model_V = d_model_Q, d_model_K, d_model_V
        self.H = H
        self.d_model = d_model if d_model is not None else d_model_Q
        self.dout_p = dout_p
        self.d_k = self.d_model // H
        assert self.d_model % H == 0
        self.linear_Q2d = nn.Linear(d_model_Q, self.d_model)
        self.linear_K2d = nn.Linear(d_model_K, self.d_model)
        self.linear_V2d = nn.Linear(d_model_V, self.d_model)
        self.linear_d2Q = nn.Linear(self.d_model, d_model_Q)
        self.dropout = nn.Dropout(dout_p)

    def _params(self):
        return (self.linear_Q2d.weight, self.linear_Q2d.bias, self.linear_K2d.weight, self.linear_K2d.bias,
                self.linear_V2d.weight, self.linear_V2d.bias, self.linear_d2Q.weight, self.linear_d2Q.bias)

    def fused(self, x, kv, mask, norm=None, residual=False, res_dropout=None, kv_cache=None, emit_bf16=False):
        """[x +] drop(MHA(LN?(x), kv, kv)).  kv=None -> self attention on the normalised x.
        The reference applies dout_p twice (attention output, residual branch); both use this module's rate
        unless the residual connection's own rate is given."""
        if kv_cache is not None and torch.is_grad_enabled():
            raise RuntimeError("kv_cache is an inference-time cache of the memory projections: use it under torch.no_grad()")
        p = self.dout_p if self.training else 0.0
        ln_w = norm.weight if norm is not None else None
        ln_b = norm.bias if norm is not None else None
        return MHAFn.apply(x, kv, ln_w, ln_b, *self._params(), mask, self.H, p, residual, kv_cache, emit_bf16)

    def fused_memory(self, x, mem, mask, norm, emit_bf16=False):
        """x + drop(MHA(LN(x), mem, mem)) for few queries against a long memory: same function as fused(x, mem, ...,
        residual=True), evaluated without ever projecting the memory (functional.MemAttnFn)."""
        p = self.dout_p if self.training else 0.0
        return MemAttnFn.apply(x, mem, norm.weight, norm.bias, *self._params(), mask, self.H, p, emit_bf16)   # mem None: self attention

    def forward(self, Q, K, V, mask, causal=False):
        """Reference signature: Q (B,Sq,Dq), K (B,Sk,Dk), V (B,Sk,Dv), mask (B,1,Sk) or (B,Sq,Sk) -> (B,Sq,Dq).
        K is V (every call site of the bimodal agent) runs the fused path; distinct K and V inputs or `causal=True`
        (the post-norm layers of model/encoder.py / model/decoder.py) run projection GEMMs + the attention core."""
        if K is V and not causal:
            return self.fused(Q, None if K is Q else K, mask)
        p = self.dout_p if self.training else 0.0
        if causal and mask is not None:
            # model/multihead_attention.py:19-22: a lower-triangular fill (sized by the key axis) on top of the mask
            Sk = mask.shape[-1]
            mask = mask.bool() & torch.ones(Sk, Sk, dtype=torch.bool, device=mask.device).tril().unsqueeze(0)
        q = LinearFn.apply(Q, self.linear_Q2d.weight, self.linear_Q2d.bias, False, 0.0)
        k = LinearFn.apply(K, self.linear_K2d.weight, self.linear_K2d.bias, False, 0.0)
        v = LinearFn.apply(V, self.linear_V2d.weight, self.linear_V2d.bias, False, 0.0)
        o = AttnCoreFn.apply(q, k, v, mask, self.H, p)
        return LinearFn.apply(o, self.linear_d2Q.weight, self.linear_d2Q.bias, False, 0.0)
