"""Helpers of the post-norm (DETR-style) layers: reference model/utils.py."""
from copy import deepcopy

import torch.nn as nn

_BUILT = {"relu"}                 # activations with a fused HIP epilogue (GEMM + bias + ReLU + dropout)
_KNOWN = {"relu", "gelu", "glu"}  # names the reference accepts (model/utils.py:4-12)


def _get_activation_fn(activation):
    """The name itself when the activation is built; the reference's other names are rejected instead of silently running
    something else, unknown names fail with the reference's message."""
    if activation in _BUILT:
        return activation
    if activation in _KNOWN:
        raise NotImplementedError(f"activation {activation!r}: only relu is built (the reference's configs use relu)")
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _get_clones(module, N):
    """N independent deep copies as a ModuleList (the layers of a stack)"""
    return nn.ModuleList(deepcopy(module) for _ in range(N))
