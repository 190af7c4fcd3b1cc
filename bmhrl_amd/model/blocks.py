"""Transformer blocks with the reference's names and state-dict layout (model/blocks.py)."""
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from ..functional import EmbedFn, FFNFn, PosEncFn


def clone(module, N):
    """N deep copies (identical initial weights), reference model/blocks.py:23-24"""
    return nn.ModuleList([deepcopy(module) for _ in range(N)])


class LayerStack(nn.Module):
    """reference model/blocks.py:10-21"""

    def __init__(self, layer, N):
        super().__init__()
        self.layers = clone(layer, N)

    def forward(self, x, masks):
        for layer in self.layers:
            x = layer(x, masks)
        return x


def posenc_table(seq_len, d_model):
    """float64 sinusoid table; every column uses its own index in the exponent (reference model/blocks.py:95-103)."""
    pos = np.arange(seq_len, dtype=np.float64)[:, None]
    col = np.arange(d_model, dtype=np.float64)[None, :]
    ang = pos / np.power(10000.0, col / d_model)
    return np.where((np.arange(d_model) % 2 == 0)[None, :], np.sin(ang), np.cos(ang))


class PositionalEncoder(nn.Module):
    """x + PE[:S], dropout.  The fp32 table is built once and cached per device (the reference converts its
    float64 table on every call, model/blocks.py:109).  No parameters, no registered buffers (as the reference)."""

    def __init__(self, d_model, dout_p, seq_len=3660):
        super().__init__()
        self.d_model = d_model
        self.dout_p = dout_p
        self.dropout = nn.Dropout(dout_p)
        self.pos_enc_mat = torch.from_numpy(posenc_table(seq_len, d_model)).unsqueeze(0)
        self._pe32 = {}

    def table(self, device):
        t = self._pe32.get(device)
        if t is None:
            t = self.pos_enc_mat[0].to(device=device, dtype=torch.float32).contiguous()
            self._pe32[device] = t
        return t

    def forward(self, x, x2=None):
        """x (+ x2) + PE; x2 lets the caller fuse rgb + flow into the same pass (K1)."""
        if x.dim() != 3:
            return x if x2 is None else x + x2
        p = self.dout_p if self.training else 0.0
        return PosEncFn.apply(x, x2, self.table(x.device), p)


class VocabularyEmbedder(nn.Module):
    """reference model/blocks.py:35-67 (GloVe of the caption width is loaded frozen unless unfreeze_word_emb)."""

    def __init__(self, voc_size, emb_dim):
        super().__init__()
        self.voc_size = voc_size
        self.emb_dim = emb_dim
        self.embedder = nn.Embedding(voc_size, emb_dim)

    def init_word_embeddings(self, weight_matrix, emb_weights_req_grad=True):
        if weight_matrix is None:
            return
        _, dim = weight_matrix.shape
        if dim != self.emb_dim:
            raise NotImplementedError("pretrained vectors of another width need the projection variant (not on the hot path)")
        self.embedder = nn.Embedding.from_pretrained(weight_matrix)
        self.embedder.weight.requires_grad = emb_weights_req_grad

    def embed_posenc(self, tok, pos_enc, tok2=None, mix=0.0):
        """-> (emb * sqrt(d) [critic input], emb * sqrt(d) + PE with dropout)"""
        p = pos_enc.dout_p if pos_enc.training else 0.0
        return EmbedFn.apply(self.embedder.weight, tok, tok2, mix, pos_enc.table(tok.device), p)

    def forward(self, x):
        return self.embedder(x) * np.sqrt(self.emb_dim)


class ResidualConnection(nn.Module):
    """Holds the pre-norm of x + dropout(sublayer(LN(x))) (reference model/blocks.py:128-144); the fused sublayers
    (MultiheadedAttention.fused, PositionwiseFeedForward.fused) take this module's `norm`."""

    def __init__(self, size, dout_p):
        super().__init__()
        self.norm = nn.LayerNorm(size)
        self.dropout = nn.Dropout(dout_p)


class PositionwiseFeedForward(nn.Module):
    """reference model/blocks.py:164-187"""

    def __init__(self, d_model, d_ff, dout_p):
        super().__init__()
        self.d_model, self.d_ff, self.dout_p = d_model, d_ff, dout_p
        self.fc1 = nn.Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dout_p)

    def fused(self, x, norm):
        """x + drop(fc2(drop(relu(fc1(LN(x))))))"""
        p = self.dout_p if self.training else 0.0
        return FFNFn.apply(x, norm.weight, norm.bias, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, p)
