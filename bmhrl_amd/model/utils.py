"""Helpers of the post-norm (DETR-style) layers: reference model/utils.py."""
from copy import deepcopy

import torch.nn as nn


def _get_activation_fn(activation):
    """Only "relu" has a fused HIP epilogue (GEMM + bias + ReLU + dropout); the reference's other names
    (model/utils.py:4-12: gelu, glu) are rejected instead of silently running something else."""
    if activation == "relu":
        return "relu"
    if activation in ("gelu", "glu"):
        raise NotImplementedError(f"activation {activation!r}: only relu is built (the reference's configs use relu)")
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _get_clones(module, N):
    return nn.ModuleList([deepcopy(module) for _ in range(N)])
