"""bmhrl_attention_bwd_scores256 (csrc/attention_bwd256.hip): P / delta / dS of the head-dimension-256 attentions with at most
256 keys in one launch -- against a float64 restatement of the softmax backward (autograd of model/multihead_attention.py:7-31)
on the same bf16 inputs, and the whole attention backward through it against the GEMM-epilogue path it replaces."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _inputs(B, H, Sq, Sk, kind, dev, packed):
    dk = 256
    D = H * dk
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    if packed:      # Q | K | V as column blocks of one (B*S, 3D) buffer (video self attention: Sq == Sk)
        QKV = torch.randn(B, Sq, 3 * D, generator=g).to(torch.bfloat16).to(dev)
        Q, K, V = QKV, QKV, QKV
        lds, offs = (3 * D,) * 3, (0, D, 2 * D)
        q, k, v = QKV[..., :D], QKV[..., D:2 * D], QKV[..., 2 * D:]
    else:
        Q = torch.randn(B, Sq, D, generator=g).to(torch.bfloat16).to(dev)
        K = torch.randn(B, Sk, D, generator=g).to(torch.bfloat16).to(dev)
        V = torch.randn(B, Sk, D, generator=g).to(torch.bfloat16).to(dev)
        lds, offs = (D,) * 3, (0, 0, 0)
        q, k, v = Q, K, V
    dO = (torch.randn(B, Sq, D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    mask = None
    if kind == "pad":
        mask = torch.ones(B, 1, Sk, dtype=torch.uint8)
        mask[0, 0, Sk - Sk // 3:] = 0
        mask[-1, 0, 5:9] = 0
    elif kind == "allmasked":
        mask = torch.ones(B, 1, Sk, dtype=torch.uint8)
        mask[0] = 0
    if mask is not None:
        mask = mask.to(dev).contiguous()
    return (Q, K, V), (q, k, v), lds, offs, dO, mask


def _reference(q, k, v, dO, mask, H, scale):
    B, Sq, D = dO.shape
    Sk = k.shape[1]
    dk = D // H
    heads = lambda t, S: t.double().reshape(B, S, H, dk).transpose(1, 2)
    qh, kh, vh, doh = heads(q, Sq), heads(k, Sk), heads(v, Sk), heads(dO, Sq)
    s = (qh @ kh.transpose(-1, -2)) * scale
    keep = torch.ones(B, 1, 1, Sk, dtype=torch.bool, device=s.device) if mask is None else (mask.view(B, 1, 1, Sk) != 0)
    s = s.masked_fill(~keep, -1e9)
    p = torch.softmax(s, -1)
    dp = doh @ vh.transpose(-1, -2)
    delta = (p * dp).sum(-1, keepdim=True)
    ds = (p * (dp - delta) * scale).masked_fill(~keep, 0.0)
    return s, p, ds


@pytest.mark.parametrize("B,H,Sq,Sk,kind,packed", [(2, 4, 256, 256, "pad", True), (2, 4, 800, 256, "pad", False),
                                                    (2, 4, 200, 250, "pad", False), (1, 2, 130, 37, "none", False),
                                                    (3, 4, 128, 8, "none", False), (2, 2, 129, 161, "allmasked", False),
                                                    (1, 4, 1, 1, "none", False), (2, 4, 33, 96, "pad", False)])
def test_scores256_match_the_softmax_backward(dev, B, H, Sq, Sk, kind, packed):
    from bmhrl_amd import ops
    dk, D = 256, H * 256
    scale = 1 / math.sqrt(dk)
    (Q, K, V), (q, k, v), lds, offs, dO, mask = _inputs(B, H, Sq, Sk, kind, dev, packed)
    s, p_ref, ds_ref = _reference(q, k, v, dO, mask, H, scale)
    # statistics as the forward kernel leaves them: any (m, l) with P = exp(s - m) / l; give the max a lag on some rows
    m = s.max(-1).values
    m[..., ::3] -= 2.5
    m = m.float().double()                 # (the statistics are fp32: form the row sum against the rounded max)
    l = torch.exp(s - m[..., None]).sum(-1)
    rmax, rsum = m.float().contiguous(), l.float().contiguous()
    Skp = ops.pad8(Sk)
    assert ops.attention_bwd_scores256_ok(Sq, Sk, dk, 0)
    P = torch.full((B, H, Sq, Skp), float("nan"), dtype=torch.bfloat16, device=dev)
    dS = torch.full((B, H, Sq, Skp), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.attention_bwd_scores256(Q, lds[0], K, lds[1], V, lds[2], dO, D, rmax, rsum, mask, Sk if mask is not None else 0, P, dS, Skp,
                                B, H, Sq, Sk, scale, q_off=offs[0], k_off=offs[1], v_off=offs[2])
    torch.cuda.synchronize()
    assert torch.isfinite(P.float()).all() and torch.isfinite(dS.float()).all()
    if Skp > Sk:        # padding columns are written, as zero
        assert float(P[..., Sk:].float().abs().max()) == 0.0 and float(dS[..., Sk:].float().abs().max()) == 0.0
    ep = float((P[..., :Sk].double() - p_ref).abs().max())
    assert ep < 4e-3 * max(float(p_ref.max()), 1e-3) + 1e-6, ep               # bf16 P
    # dS against the reference; tolerance: bf16 rounding of P and dS relative to the row's largest |dS|
    eds = float((dS[..., :Sk].double() - ds_ref).abs().max() / ds_ref.abs().max().clamp_min(1e-12))
    assert eds < 1.5e-2, eds
    if mask is not None:
        dead = (mask.view(B, 1, 1, Sk) == 0).expand(B, H, Sq, Sk)
        assert float(dS[..., :Sk][dead].float().abs().max()) == 0.0             # masked_fill passes no gradient
    if kind == "allmasked":  # a fully masked sample attends uniformly over every key
        assert float((P[0, ..., :Sk].float() - 1.0 / Sk).abs().max()) < 4e-3 / Sk + 1e-6
    # the rows of dS cancel against the P the dV product reads: sum_k dS = scale * (sum_k P dP - delta sum_k P) ~ 0
    rows = dS[..., :Sk].double().sum(-1).abs().max()
    assert float(rows) < 2e-2 * float(ds_ref.abs().max()) * math.sqrt(Sk) + 1e-9


def test_shapes_outside_the_kernel_are_refused(dev):
    from bmhrl_amd import ops
    assert not ops.attention_bwd_scores256_ok(256, 257, 256, 0)       # more than 256 keys
    assert not ops.attention_bwd_scores256_ok(256, 256, 128, 0)       # other head dimension
    assert not ops.attention_bwd_scores256_ok(256, 256, 256, 256)     # per-query mask
    t = torch.zeros(8, 2048, dtype=torch.bfloat16, device=dev)
    st = torch.ones(8, device=dev)
    with pytest.raises(RuntimeError):
        ops.attention_bwd_scores256(t, 1024, t, 1024, t, 1024, t, 1024, st, st, None, 0, t, t, 264, 1, 4, 2, 257, 1.0)


@pytest.mark.parametrize("Sq,Sk,packed", [(256, 256, True), (800, 256, False), (200, 250, False)])
def test_attention_backward_through_the_fused_scores_equals_the_gemm_path(dev, monkeypatch, Sq, Sk, packed):
    """dQ / dK / dV (and the bias column sums) of functional._attn_core_bwd with the one-launch score backward against the
    delta + PROB-GEMM + DSCORE-GEMM path on the same saved forward."""
    from bmhrl_amd import functional as F, ops
    B, H, dk = 2, 4, 256
    D = H * dk
    scale = 1 / math.sqrt(dk)
    (Q, K, V), _, lds, offs, dO, mask = _inputs(B, H, Sq, Sk, "pad", dev, packed)
    msb = Sk
    O = torch.empty(B * Sq, D, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, msb, 0, B, H, Sq, Sk, dk, scale, lds[0], lds[1], lds[2], D, q_off=offs[0],
                      k_off=offs[1], v_off=offs[2])
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(F, "FUSED_SCORES_BWD", fused)
        dQ = torch.zeros(B * Sq, D, dtype=torch.bfloat16, device=dev)
        dK = torch.zeros(B * Sk, D, dtype=torch.bfloat16, device=dev)
        dV = torch.zeros(B * Sk, D, dtype=torch.bfloat16, device=dev)
        dbs = [torch.zeros(D, device=dev) for _ in range(3)]
        F._attn_core_bwd(dO.view(B * Sq, D), O, ("flash", rmax, rsum), Q, offs[0], lds[0], K, offs[1], lds[1], V, offs[2], lds[2],
                         dQ, 0, D, dK, 0, D, dV, 0, D, mask, msb, 0, B, H, Sq, Sk, dk, 0.0, db_q=(dbs[0], 0), db_k=(dbs[1], 0),
                         db_v=(dbs[2], 0))
        torch.cuda.synchronize()
        out[fused] = (dQ, dK, dV, *dbs)
    ref_scale = {n: t for n, t in zip(("dQ", "dK", "dV", "db_q", "db_k", "db_v"), out[False])}
    for a, b, name in zip(out[True], out[False], ("dQ", "dK", "dV", "db_q", "db_k", "db_v")):
        # the two paths differ by the row term only: sum_k bf16(P) dP against sum_d dO O of the bf16 output.  The key bias
        # gradient is sum_q Q[q] sum_k dS[q, k] -- zero but for rounding (softmax is shift invariant): measured against db_q.
        denom = ref_scale["db_q"] if name == "db_k" else b
        e = float((a.double() - b.double()).norm() / denom.double().norm().clamp_min(1e-12))
        assert e < (5e-2 if name == "db_k" else 1.5e-2), (name, e)
