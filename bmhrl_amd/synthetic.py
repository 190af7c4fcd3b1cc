"""Deterministic synthetic weights and batches.

Everything here is numpy-seeded so the same values appear in the build container (where
the golden fixtures are generated against the reference) and on the GPU box.
"""
from __future__ import annotations

import math
import zlib
from types import SimpleNamespace
from typing import Dict, Tuple

import numpy as np
import torch


def default_cfg(**over) -> SimpleNamespace:
    """The cfg attribute bag BMHrlAgent reads (reference main.py defaults)."""
    cfg = SimpleNamespace(
        d_vid=1024, d_aud=128, d_model_video=1024, d_model_audio=128, d_model_caps=300, d_model=1024,
        rl_projection_d=512, rl_att_heads=4, rl_att_layers=2, rl_goal_d=64, rl_ff_c=2048, rl_ff_v=1024,
        rl_ff_a=512, dout_p=0.1, rl_critic_score_threshhold=0.25, unfreeze_word_emb=False,
        rl_critic_path=None, device="cpu", device_ids=[0], smoothing=0.7, rl_stabilize=False,
        modality="audio_video", max_len=30, grad_clip=None, pad_idx=1, start_idx=2, end_idx=3,
    )
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def tiny_cfg(**over) -> SimpleNamespace:
    """Small dims used by the golden fixtures (every odd size of the real model kept odd)."""
    base = dict(d_vid=48, d_aud=24, d_model_video=48, d_model_audio=24, d_model_caps=20, d_model=64,
                rl_att_heads=4, rl_att_layers=2, rl_goal_d=8, rl_ff_c=40, rl_ff_v=48, rl_ff_a=32, dout_p=0.0)
    base.update(over)
    return default_cfg(**base)


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def det_tensor(name: str, shape, seed: int = 0, clone_layers: bool = False) -> torch.Tensor:
    """A deterministic fp32 tensor for state-dict entry ``name`` (scaled like torch default inits)."""
    key = name
    if clone_layers:
        import re
        key = re.sub(r"\.layers\.\d+\.", ".layers.0.", name)
    g = _rng(key, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if "norm" in name.lower() and leaf == "weight":
        v = 1.0 + 0.1 * g.uniform(-1, 1, shape)
    elif "norm" in name.lower() and leaf == "bias":
        v = 0.05 * g.uniform(-1, 1, shape)
    elif leaf == "a_v_constant":
        v = 0.6 * g.uniform(-1, 1, shape)
    elif leaf == "alpha":
        v = 0.9 + 0.05 * g.uniform(-1, 1, shape)
    elif leaf == "beta":
        v = 2.0 + 0.1 * g.uniform(-1, 1, shape)
    elif name.endswith("embedder.weight"):
        v = g.standard_normal(shape)
    elif len(shape) >= 2:
        b = 1.0 / math.sqrt(shape[-1])
        v = g.uniform(-b, b, shape)
    else:
        v = 0.05 * g.uniform(-1, 1, shape)
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, clone_layers: bool = False) -> Dict[str, torch.Tensor]:
    sd = {k: det_tensor(k, s, seed, clone_layers) for k, s in shapes.items()}
    # the reference registers the same LinearCore twice (manager_core == manager.core)
    for k in list(sd):
        if k.startswith("manager.core."):
            twin = "manager_core." + k[len("manager.core."):]
            if twin in sd:
                sd[k] = sd[twin]
    return sd


def critic_shapes(d_caps: int) -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of the frozen SegmentCritic (reference model/bm_hrl_agent.py:186-196)."""
    h = 2 * d_caps
    s: Dict[str, Tuple[int, ...]] = {}
    for l in range(4):
        s[f"lstm.weight_ih_l{l}"] = (4 * h, d_caps if l == 0 else h)
        s[f"lstm.weight_hh_l{l}"] = (4 * h, h)
        s[f"lstm.bias_ih_l{l}"] = (4 * h,)
        s[f"lstm.bias_hh_l{l}"] = (4 * h,)
    for l in range(2):
        s[f"gru.weight_ih_l{l}"] = (3 * h, h)
        s[f"gru.weight_hh_l{l}"] = (3 * h, h)
        s[f"gru.bias_ih_l{l}"] = (3 * h,)
        s[f"gru.bias_hh_l{l}"] = (3 * h,)
    s["lin.weight"] = (1, h)
    s["lin.bias"] = (1,)
    for r in ("relu", "relu2"):
        s[f"{r}.alpha"] = (1,)
        s[f"{r}.beta"] = (1,)
    return s


def synthetic_critic_state(d_caps: int, seed: int = 1) -> Dict[str, torch.Tensor]:
    """Stand-in for the critic checkpoint the reference downloads (BASELINE config 3 stubs it)."""
    return {k: det_tensor("critic." + k, s, seed) for k, s in critic_shapes(d_caps).items()}


def synthetic_batch(B: int, Tv: int, Ta: int, L: int, V: int, seed: int = 0, d_vid: int = 1024, d_aud: int = 128,
                    min_len: int = 8, pad_tails: bool = True) -> Dict[str, torch.Tensor]:
    """Synthetic I3D + VGGish batch honouring the loader contract.

    rgb, flow ~ U[0,1) (B,Tv,d_vid); audio ~ U[0,1) (B,Ta,d_aud); the last r_b = (37 b) mod (T/4) rows
    of sample b are zero padding; captions (B, L+1): <s>=2, tokens in [4,V), </s>=3 at l_b, pad=1 after.
    """
    g = np.random.Generator(np.random.PCG64(seed))
    rgb = g.random((B, Tv, d_vid), dtype=np.float32)
    flow = g.random((B, Tv, d_vid), dtype=np.float32)
    audio = g.random((B, Ta, d_aud), dtype=np.float32)
    # strictly non-zero first feature on valid rows (masks are built from column 0)
    rgb[:, :, 0] = np.maximum(rgb[:, :, 0], 1e-3)
    audio[:, :, 0] = np.maximum(audio[:, :, 0], 1e-3)
    if pad_tails:
        for b in range(B):
            rv = (37 * b) % max(Tv // 4, 1)
            ra = (37 * b) % max(Ta // 4, 1)
            if rv:
                rgb[b, Tv - rv:] = 0
                flow[b, Tv - rv:] = 0
            if ra:
                audio[b, Ta - ra:] = 0
    cap = np.full((B, L + 1), 1, dtype=np.int64)
    cap[:, 0] = 2
    lo = min(min_len, L)
    for b in range(B):
        lb = int(g.integers(lo, L + 1))  # position of </s>, in [lo, L]
        cap[b, 1:lb] = g.integers(4, V, size=max(lb - 1, 0))
        cap[b, lb] = 3
    return {"rgb": torch.from_numpy(rgb), "flow": torch.from_numpy(flow), "audio": torch.from_numpy(audio),
            "captions": torch.from_numpy(cap)}


def synthetic_rewards(B: int, L: int, seed: int = 2) -> torch.Tensor:
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.random((B, L), dtype=np.float32))


def detr_tiny_modules():
    """The post-norm encoder / decoder stacks at the size of tests/golden/detr.npz, with the fixture's weights
    (fill_state_dict seed 9).  Returns (encoder, decoder, dims)."""
    import torch.nn as nn
    from .model.decoder import TransformerDecoder, TransformerDecoderLayer
    from .model.encoder import TransformerEncoder, TransformerEncoderLayer
    dims = dict(B=3, S=9, L=6, D=64, dC=24, dG=8, H=4, dff=48)
    enc = TransformerEncoder(TransformerEncoderLayer(64, 4, 48, 0.0, embed_size=20), 2, nn.LayerNorm(64))
    dec = TransformerDecoder(TransformerDecoderLayer(64, 4, 24, 8, 48, 0.0), 2, nn.LayerNorm(24))
    for m in (enc, dec):
        m.load_state_dict(fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=9))
    return enc, dec, dims
