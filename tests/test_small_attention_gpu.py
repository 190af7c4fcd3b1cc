"""One-launch attention core for short sequences (csrc/small_attention.hip: caption self attention, goal attention) against
the batched-GEMM path it replaces and against plain fp32 torch (model/multihead_attention.py:7-31 semantics: -1e9 fill,
dropout on the output, no score gradient at masked keys)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _case(dev, B, H, Sq, Sk, dk, mask_kind, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    D = H * dk
    qkv = (torch.randn(B * max(Sq, Sk), 3 * D, generator=g) * 0.5).to(dev).bfloat16()
    dO = (torch.randn(B * Sq, D, generator=g) * 0.5).to(dev).bfloat16()
    if mask_kind == "none":
        m = None
    elif mask_kind == "key":
        m = (torch.rand(B, 1, Sk, generator=g) > 0.3)
        m[:, :, 0] = True
        m[0] = False                                  # a fully masked sample: uniform attention, no score gradient
    else:                                             # causal & padding, (B, Sq, Sk)
        m = torch.tril(torch.ones(Sq, Sk, dtype=torch.bool)).unsqueeze(0).repeat(B, 1, 1)
        m[1, :, Sk // 2:] = False
    return qkv, dO, (None if m is None else m.to(dev))


def _run(small, qkv, dO, mask, B, H, Sq, Sk, dk, p_drop, seed, with_bias):
    from bmhrl_amd import functional as F
    D = H * dk
    dev = qkv.device
    old = F.SMALL_ATTN
    F.SMALL_ATTN = small
    try:
        m8, msb, msq = F._mask_u8(mask)
        Q, K, V = qkv[:B * Sq], qkv[:B * Sk], qkv[:B * Sk]
        O, stats = F._attn_core_fwd(Q, 0, 3 * D, K, D, 3 * D, V, 2 * D, 3 * D, m8, msb, msq, B, H, Sq, Sk, dk, p_drop, seed)
        assert stats[0] == "mat"
        dQ = torch.zeros(B * Sq, 3 * D, dtype=torch.bfloat16, device=dev)
        dKV = torch.zeros(B * Sk, 3 * D, dtype=torch.bfloat16, device=dev)
        db = torch.zeros(3 * D, device=dev) if with_bias else None
        kw = dict(db_q=(db, 0), db_k=(db, D), db_v=(db, 2 * D)) if with_bias else {}
        F._attn_core_bwd(dO, O, stats, Q, 0, 3 * D, K, D, 3 * D, V, 2 * D, 3 * D, dQ, 0, 3 * D, dKV, D, 3 * D, dKV, 2 * D, 3 * D,
                         m8, msb, msq, B, H, Sq, Sk, dk, p_drop, **kw)
        return O, stats[1], dQ[:, :D], dKV[:, D:2 * D], dKV[:, 2 * D:], db
    finally:
        F.SMALL_ATTN = old


def _reference(qkv, dO, mask, B, H, Sq, Sk, dk):
    D = H * dk
    q = qkv[:B * Sq, :D].float().view(B, Sq, H, dk).transpose(1, 2).requires_grad_(True)
    k = qkv[:B * Sk, D:2 * D].float().view(B, Sk, H, dk).transpose(1, 2).requires_grad_(True)
    v = qkv[:B * Sk, 2 * D:].float().view(B, Sk, H, dk).transpose(1, 2).requires_grad_(True)
    s = q @ k.transpose(-1, -2) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B * Sq, D)
    o.backward(dO.float())
    back = lambda t, S: t.grad.transpose(1, 2).reshape(B * S, D)
    return o.detach(), p.detach(), back(q, Sq), back(k, Sk), back(v, Sk)


@pytest.mark.parametrize("B,H,Sq,Sk,dk,mask_kind", [
    (32, 4, 30, 30, 256, "causal"), (16, 2, 30, 30, 512, "causal"), (3, 2, 7, 19, 64, "key"), (2, 3, 32, 32, 128, "none"),
    (4, 4, 1, 5, 256, "key"), (2, 2, 30, 12, 256, "none")])
def test_small_attention_equals_gemm_path_and_reference(B, H, Sq, Sk, dk, mask_kind):
    dev = torch.device("cuda:0")
    from bmhrl_amd import ops
    assert ops.small_attention_ok(Sq, Sk, dk) and not ops.small_attention_ok(33, 30, 256) and not ops.small_attention_ok(30, 30, 96)
    qkv, dO, mask = _case(dev, B, H, Sq, Sk, dk, mask_kind, seed=B + Sq + dk)
    got = _run(True, qkv, dO, mask, B, H, Sq, Sk, dk, 0.0, 0, True)
    old = _run(False, qkv, dO, mask, B, H, Sq, Sk, dk, 0.0, 0, True)
    ref = _reference(qkv, dO, mask, B, H, Sq, Sk, dk)
    names = ("O", "P", "dQ", "dK", "dV")
    for n, a, b, r in zip(names, got, old, ref):
        if n == "P":
            a, b = a[..., :Sk], b[..., :Sk]
            assert float(got[1][..., Sk:].float().abs().max()) == 0.0 if got[1].shape[-1] > Sk else True
        assert rel_err(a, r) < 1.5e-2, (n, rel_err(a, r))
        assert rel_err(a, b) < 1.5e-2, (n, rel_err(a, b))
    # bias gradients = column sums of dQ | dK | dV
    D = H * dk
    cs = torch.cat([got[2].float().sum(0), got[3].float().sum(0), got[4].float().sum(0)])
    assert rel_err(got[5], cs) < 2e-3 and rel_err(got[5], old[5]) < 3e-2
    if mask_kind == "key":      # the fully masked sample: uniform probabilities, zero dQ / dK
        assert rel_err(got[1][0, :, :, :Sk], torch.full_like(got[1][0, :, :, :Sk].float(), 1.0 / Sk)) < 1e-2
        assert float(got[2][:Sq].float().abs().max()) == 0.0 and float(got[3][:Sk].float().abs().max()) == 0.0


def test_small_attention_dropout_ids_match_the_gemm_epilogue():
    """same seed -> the same elements are dropped as by the context GEMM's epilogue (the out-projection's backward
    regenerates the mask from those ids)"""
    dev = torch.device("cuda:0")
    B, H, Sq, Sk, dk = 8, 4, 30, 30, 256
    qkv, dO, mask = _case(dev, B, H, Sq, Sk, dk, "causal", seed=11)
    a = _run(True, qkv, dO, mask, B, H, Sq, Sk, dk, 0.3, 12345, False)[0]
    b = _run(False, qkv, dO, mask, B, H, Sq, Sk, dk, 0.3, 12345, False)[0]
    assert torch.equal(a == 0, b == 0) and 0.25 < float((a == 0).float().mean()) < 0.35
    assert rel_err(a, b) < 1.5e-2
