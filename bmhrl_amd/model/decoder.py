"""Post-norm transformer decoder with the reference's names and state-dict keys (model/decoder.py:7-107) on the HIP
kernels (see encoder.py).  Quirks kept: norm1 is applied to the decoder input BEFORE the self-attention branch is added
(:77-78); memory keys carry the position code, memory values do not; `forward` drops the caller's obj_mask (:106)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import LayerNormFn
from .encoder import add_norm, feed_forward, run_stack
from .multihead_attention import MultiheadedAttention
from .utils import _get_activation_fn, _get_clones


class TransformerDecoder(nn.Module):

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=True):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos=None, add_pos=False,
                detected_objects=None, obj_mask=None):
        return run_stack(self.layers, self.norm, self.return_intermediate, tgt,
                         lambda layer, x: layer(x, memory, mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos,
                                                add_pos, detected_objects, obj_mask))


class TransformerDecoderLayer(nn.Module):

    def __init__(self, d_model, nhead, d_model_C, d_goal, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadedAttention(d_model_C, d_model_C, d_model_C, nhead, dropout, d_model)
        self.multihead_attn = MultiheadedAttention(d_model_C, d_model, d_model, nhead, dropout, d_model)
        self.detected_attention = MultiheadedAttention(d_model_C, 256, 256, nhead, dropout, d_model)
        self.linear1 = nn.Linear(d_model_C, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model_C)
        for i in range(1, 6):
            setattr(self, f"norm{i}", nn.LayerNorm(d_model_C))
        for i in range(1, 6):
            setattr(self, f"dropout{i}", nn.Dropout(dropout))
        self.goal_attention = MultiheadedAttention(d_model_C, d_goal, d_goal, nhead, dropout, d_model)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        self.positional_encoding = nn.Parameter()        # empty in the reference too (checkpoint key of shape (0,))

    def forward_post(self, tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos,
                     detected_objects=None, add_pos=False, obj_mask=None):
        tr = self.training
        p = lambda d: d.p if tr else 0.0  # noqa: E731
        if not add_pos:                                   # query_pos is a PositionalEncoder; causal self attention
            causal, qk = True, query_pos(tgt)
        else:                                             # query_pos is a tensor of learned / given positions
            causal, qk = False, tgt + query_pos
        branch = self.self_attn(qk, qk, tgt, query_mask, causal=causal)
        tgt = LayerNormFn.apply(tgt, self.norm1.weight, self.norm1.bias) + F.dropout(branch, p(self.dropout1), tr)
        tgt = add_norm(tgt, self.multihead_attn(qk, pos(memory), memory, memory_mask), self.norm2, p(self.dropout2), tr)
        if goal is not None:
            branch = self.goal_attention(query_pos(tgt), goal_pos(goal), goal, goal_mask)
            tgt = add_norm(tgt, branch, self.norm4, p(self.dropout4), tr)
        if detected_objects is not None:
            branch = self.detected_attention(qk, detected_objects, detected_objects, obj_mask)
            tgt = add_norm(tgt, branch, self.norm5, p(self.dropout5), tr)
        ff = feed_forward(tgt, self.linear1, self.linear2, p(self.dropout))
        return add_norm(tgt, ff, self.norm3, p(self.dropout3), tr)

    def forward(self, tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos, add_pos=False,
                detected_objects=None, obj_mask=None):
        return self.forward_post(tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos,
                                 add_pos=add_pos, detected_objects=detected_objects, obj_mask=None)
