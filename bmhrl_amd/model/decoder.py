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

    OBJECT_WIDTH = 256          # width of the object detector's query states (model/object_detector.py)

    def __init__(self, d_model, nhead, d_model_C, d_goal, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        attend = lambda d_k, d_v: MultiheadedAttention(d_model_C, d_k, d_v, nhead, dropout, d_model)      # noqa: E731
        # registration order of the reference (:42-62): three attentions, the feed-forward pair, five norms, five dropouts,
        # the goal attention, the (empty) positional_encoding parameter
        self.self_attn, self.multihead_attn = attend(d_model_C, d_model_C), attend(d_model, d_model)
        self.detected_attention = attend(self.OBJECT_WIDTH, self.OBJECT_WIDTH)
        self.linear1, self.dropout = nn.Linear(d_model_C, dim_feedforward), nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model_C)
        for kind, make in (("norm", lambda: nn.LayerNorm(d_model_C)), ("dropout", lambda: nn.Dropout(dropout))):
            for i in range(1, 6):
                setattr(self, f"{kind}{i}", make())
        self.goal_attention = attend(d_goal, d_goal)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        self.positional_encoding = nn.Parameter()        # empty in the reference too (checkpoint key of shape (0,))

    def forward_post(self, tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos,
                     detected_objects=None, add_pos=False, obj_mask=None):
        train = self.training
        rate = lambda d: d.p if train else 0.0  # noqa: E731
        # queries / keys of the self attention: a PositionalEncoder (causal attention over the caption) or learned positions
        qk, causal = (tgt + query_pos, False) if add_pos else (query_pos(tgt), True)
        own = self.self_attn(qk, qk, tgt, query_mask, causal=causal)
        # (the reference normalises the INPUT and then adds the branch: :77-78)
        x = LayerNormFn.apply(tgt, self.norm1.weight, self.norm1.bias) + F.dropout(own, rate(self.dropout1), train)
        x = add_norm(x, self.multihead_attn(qk, pos(memory), memory, memory_mask), self.norm2, rate(self.dropout2), train)
        if goal is not None:
            x = add_norm(x, self.goal_attention(query_pos(x), goal_pos(goal), goal, goal_mask), self.norm4, rate(self.dropout4), train)
        if detected_objects is not None:
            x = add_norm(x, self.detected_attention(qk, detected_objects, detected_objects, obj_mask), self.norm5, rate(self.dropout5),
                         train)
        return add_norm(x, feed_forward(x, self.linear1, self.linear2, rate(self.dropout)), self.norm3, rate(self.dropout3), train)

    def forward(self, tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos, add_pos=False,
                detected_objects=None, obj_mask=None):
        # (the caller's obj_mask is dropped, as the reference does at :106)
        return self.forward_post(tgt, memory, memory_mask, pos, query_pos, query_mask, goal, goal_mask, goal_pos,
                                 add_pos=add_pos, detected_objects=detected_objects, obj_mask=None)
