"""Post-norm transformer encoder with the reference's names and state-dict keys (model/encoder.py:10-76), on the HIP
kernels: projections / FFN through bmhrl_gemm (bf16 MFMA), attention through the flash / materialised attention core,
LayerNorm through bmhrl_layernorm_*.  Used by the reference's DETR-mode agent; the bimodal agent does not call it."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import LayerNormFn, LinearFn
from .multihead_attention import MultiheadedAttention
from .utils import _get_activation_fn, _get_clones


def add_norm(x, branch, norm, p, training):
    """norm(x + dropout(branch)) -- the post-norm residual step of model/encoder.py:63-64, 66-67."""
    return LayerNormFn.apply(x + F.dropout(branch, p, training), norm.weight, norm.bias)


def feed_forward(x, linear1, linear2, p):
    """linear2(dropout(relu(linear1 x))), bias / ReLU / dropout fused into the first GEMM's epilogue."""
    h = LinearFn.apply(x, linear1.weight, linear1.bias, True, p)
    return LinearFn.apply(h, linear2.weight, linear2.bias, False, 0.0)


def run_stack(layers, norm, return_intermediate, output, call):
    """Shared by encoder and decoder (model/encoder.py:23-37, model/decoder.py:20-37).  With a final norm the last
    intermediate is replaced by norm(norm(output)): the reference normalises `output` first and then appends
    `self.norm(output)` again; reproduced."""
    intermediate = []
    for layer in layers:
        output = call(layer, output)
        if return_intermediate:
            intermediate.append(output)
    if norm is not None:
        output = LayerNormFn.apply(output, norm.weight, norm.bias)
        if return_intermediate:
            intermediate.pop()
            intermediate.append(LayerNormFn.apply(output, norm.weight, norm.bias))
    return torch.stack(intermediate) if return_intermediate else output


class TransformerEncoder(nn.Module):

    def __init__(self, encoder_layer, num_layers, norm=None, cfg=None, return_intermediate=True):
        super().__init__()
        self.cfg = cfg
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, src, mask, pos_enc):
        return run_stack(self.layers, self.norm, self.return_intermediate, src,
                         lambda layer, x: layer(x, src_mask=mask, pos=pos_enc))


class TransformerEncoderLayer(nn.Module):

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=True,
                 embed_size=300):
        super().__init__()
        self.self_attn = MultiheadedAttention(d_model, d_model, d_model, nhead, dropout, d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.embed = nn.Linear(d_model, embed_size)      # never applied by the reference either; kept for the checkpoint keys
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before         # ignored by the reference as well: always post-norm

    def forward_post(self, src, mask, pos):
        """model/encoder.py:59-69: queries and keys carry the position code, values do not."""
        p = lambda d: d.p if self.training else 0.0  # noqa: E731
        qk = pos(src)
        src = add_norm(src, self.self_attn(qk, qk, src, mask), self.norm1, p(self.dropout1), self.training)
        ff = feed_forward(src, self.linear1, self.linear2, p(self.dropout))
        return add_norm(src, ff, self.norm2, p(self.dropout2), self.training)

    def forward(self, src, src_mask, pos):
        return self.forward_post(src, src_mask, pos)
