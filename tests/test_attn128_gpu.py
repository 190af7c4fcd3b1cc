"""bmhrl_attention_shared128_fwd: absorbed-projection attention (head dim 128, one key/value tile for all heads)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


# (8, 4, ...): batch % 8 == 0 takes the "all (head, q-tile) of a batch row on one XCD" workgroup map; Sk = 20: the second
# key half of the only tile is all padding; Sq = 70 / 200: ragged last q-tile (rows past Sq must not be stored)
@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 4, 64, 64), (2, 4, 200, 130), (3, 2, 256, 800), (1, 4, 800, 800), (8, 4, 70, 200),
                                       (16, 2, 96, 1024), (2, 4, 64, 20),
                                       (16, 4, 256, 800), (16, 4, 800, 800)])     # the bench workload's V<-A and audio self attention
def test_shared128_attention_matches_torch(B, H, Sq, Sk):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + Sq + Sk)
    Qp = (0.5 * torch.randn(B, Sq, H, 128, generator=g)).to(dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, 128, generator=g).to(dev).to(torch.bfloat16)
    mask = torch.ones(B, Sk, dtype=torch.bool, device=dev)
    mask[0, Sk - 5:] = False
    if B > 1:
        mask[B - 1, :] = False                     # fully masked sample: uniform attention
    scale = 1.0 / math.sqrt(256)
    ctx = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, mask, Sk, B, H, Sq, Sk, scale, H * 128, 128, H * 128)
    torch.cuda.synchronize()
    s = torch.einsum("bqhd,bkd->bhqk", Qp.float(), X.float()) * scale
    s = s.masked_fill(~mask[:, None, None, :], -1e9)
    p = torch.softmax(s, -1)
    ref = torch.einsum("bhqk,bkd->bqhd", p, X.float())
    assert float((ctx.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    # (row_max, row_sum) is a consistent pair, not necessarily the exact maximum (the kernel rescales lazily):
    # P = exp(score - row_max) / row_sum, i.e. row_max + log(row_sum) is the log-sum-exp of the row
    lse = torch.logsumexp(s, -1)
    got = rmax + torch.log(rsum)
    ok = lse > -1e8
    assert float((got[ok] - lse[ok]).abs().max()) < 2e-2
    if (~ok).any():                                # fully masked rows keep the exact fill value (backward relies on it)
        assert float((rmax[~ok] + 1e9).abs().max()) == 0.0 and float((rsum[~ok] - Sk).abs().max()) < 1e-3 * Sk


def test_shared128_attention_strided_rows_and_no_mask():
    """leading dimensions larger than the rows (the operands are column slices of wider buffers), mask = NULL; the
    columns next to the output slice must stay untouched"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    B, H, Sq, Sk = 8, 4, 100, 300
    g = torch.Generator().manual_seed(7)
    Qw = (0.5 * torch.randn(B, Sq, H * 128 + 64, generator=g)).to(dev).to(torch.bfloat16)
    Xw = torch.randn(B, Sk, 128 + 8, generator=g).to(dev).to(torch.bfloat16)
    ctxw = torch.full((B, Sq, H * 128 + 16), 7.0, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    scale = 1.0 / 16
    ops.attention_shared128_fwd(Qw, Xw, ctxw, rmax, rsum, None, 0, B, H, Sq, Sk, scale, H * 128 + 64, 128 + 8, H * 128 + 16)
    torch.cuda.synchronize()
    Qp = Qw[..., :H * 128].float().view(B, Sq, H, 128)
    X = Xw[..., :128].float()
    p = torch.softmax(torch.einsum("bqhd,bkd->bhqk", Qp, X) * scale, -1)
    ref = torch.einsum("bhqk,bkd->bqhd", p, X).reshape(B, Sq, H * 128)
    assert float((ctxw[..., :H * 128].float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    assert float((ctxw[..., H * 128:].float() - 7.0).abs().max()) == 0.0


def test_shared128_masked_keys_do_not_matter_and_key_order_is_free():
    """size-independent properties at the bench shape: rows of X at masked keys can hold anything (bit-identical output),
    and permuting the keys (with their mask) changes the result only by rounding"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    B, H, Sq, Sk = 16, 4, 256, 800
    g = torch.Generator().manual_seed(5)
    Qp = (0.5 * torch.randn(B, Sq, H, 128, generator=g)).to(dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, 128, generator=g).to(dev).to(torch.bfloat16)
    mask = (torch.rand(B, Sk, generator=g) > 0.2).to(dev)
    mask[:, 0] = True

    def run(Xi, mi):
        ctx = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
        rmax = torch.empty(B, H, Sq, device=dev); rsum = torch.empty(B, H, Sq, device=dev)
        ops.attention_shared128_fwd(Qp, Xi.contiguous(), ctx, rmax, rsum, mi.contiguous(), Sk, B, H, Sq, Sk, 1 / 16, H * 128, 128, H * 128)
        return ctx.float(), rmax + torch.log(rsum)

    base, lse = run(X, mask)
    X2 = X.clone()
    X2[~mask] = 37.0                                    # garbage (finite) at the masked keys
    other, lse2 = run(X2, mask)
    assert torch.equal(base, other) and torch.equal(lse, lse2)
    perm = torch.randperm(Sk, generator=g).to(dev)
    pout, plse = run(X[:, perm], mask[:, perm])
    assert float((pout - base).abs().max()) < 2e-2 * float(base.abs().max())
    assert float((plse - lse).abs().max()) < 2e-3


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 4, 64, 64), (3, 2, 200, 130), (8, 4, 70, 200), (2, 4, 256, 800), (16, 4, 256, 800),
                                       (4, 4, 800, 800), (2, 2, 96, 2100)])
def test_shared128_fused_backward_matches_torch_autograd(B, H, Sq, Sk):
    """bmhrl_attention_shared128_bwd (P / dS recomputed per tile, never in HBM) against torch autograd of the same attention
    on the bf16-rounded operands: dQp and d(mem) incl. a padded tail, a fully masked sample (uniform attention: its values
    get gradient, its scores none -- masked_fill has no gradient) and ragged query / key tiles."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 77 + Sq + 3 * Sk)
    Qp = (0.5 * torch.randn(B, Sq, H, 128, generator=g)).to(dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, 128, generator=g).to(dev).to(torch.bfloat16)
    dCx = torch.randn(B, Sq, H, 128, generator=g).to(dev).to(torch.bfloat16)
    mask = torch.ones(B, Sk, dtype=torch.bool, device=dev)
    mask[0, Sk - 37:] = False
    mask[0, 3] = False
    if B > 1:
        mask[B - 1, :] = False
    scale = 1.0 / 16
    ctx = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, mask, Sk, B, H, Sq, Sk, scale, H * 128, 128, H * 128)
    # reference: fp32 autograd
    q32 = Qp.float().requires_grad_(True)
    x32 = X.float().requires_grad_(True)
    s = torch.einsum("bqhd,bkd->bhqk", q32, x32) * scale
    s = s.masked_fill(~mask[:, None, None, :], -1e9)
    ref = torch.einsum("bhqk,bkd->bqhd", torch.softmax(s, -1), x32)
    ref.backward(dCx.float())
    delta = torch.empty(B, H, Sq, device=dev)
    # delta from the fp32 context (the kernel's own bf16 context differs by rounding only; the step uses the bf16 one)
    ops.attn_delta(dCx, H * 128, ctx, H * 128, delta, B, H, Sq, 128)
    dQp = torch.full((B, Sq, H, 128), 7.0, dtype=torch.bfloat16, device=dev)
    dX = torch.full((B, Sk, 128), 3.0, device=dev)
    ops.attention_shared128_bwd(Qp, X, dCx, rmax, rsum, delta, mask, Sk, dQp, dX, False, B, H, Sq, Sk, scale,
                                H * 128, 128, H * 128, H * 128)
    torch.cuda.synchronize()
    eq = float((dQp.float() - q32.grad).abs().max() / q32.grad.abs().max())
    ex = float((dX - x32.grad).abs().max() / x32.grad.abs().max())
    assert eq < 2e-2 and ex < 2e-2, (eq, ex)
    if B > 1:                                           # the fully masked sample: no score gradient at all
        assert float(dQp[B - 1].float().abs().max()) == 0.0
    # accumulate mode adds to what is there; dX = NULL skips the key-side kernel
    dX2 = dX.clone()
    ops.attention_shared128_bwd(Qp, X, dCx, rmax, rsum, delta, mask, Sk, dQp, dX2, True, B, H, Sq, Sk, scale,
                                H * 128, 128, H * 128, H * 128)
    assert float((dX2 - 2 * dX).abs().max()) <= 1e-5 * float(dX.abs().max())
    ops.attention_shared128_bwd(Qp, X, dCx, rmax, rsum, delta, None if False else mask, Sk, dQp, None, False, B, H, Sq, Sk, scale,
                                H * 128, 128, H * 128, H * 128)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,Sq,Sk", [(16, 4, 256, 800), (2, 4, 200, 130), (8, 4, 70, 200), (3, 2, 256, 800), (2, 4, 64, 20)])
def test_two_heads_per_wave_form_matches_the_default_kernel(B, H, Sq, Sk):
    """bmhrl_attention_config(128, 14): the r04 kernel in which one wave carries two heads of its query rows over four key
    splits (csrc/attention_pair.h; opt-in, BMHRL_ATTN_PAIR=1) against the shipped kernel on the same inputs -- contexts to bf16
    rounding, log-sum-exp of the statistics to 2e-3; shapes the form does not serve (odd head counts) fall back by themselves."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import _lib, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + Sq + Sk)
    Qp = (0.5 * torch.randn(B, Sq, H, 128, generator=g)).to(dev).to(torch.bfloat16)
    X = torch.randn(B, Sk, 128, generator=g).to(dev).to(torch.bfloat16)
    mask = torch.ones(B, Sk, dtype=torch.bool, device=dev)
    mask[0, Sk - 5:] = False
    mask[B - 1, :] = False                         # fully masked sample: uniform attention
    scale = 1.0 / math.sqrt(256)
    out = {}
    try:
        for code in (0, 14):
            _lib.check(_lib.load().bmhrl_attention_config(128, code), "bmhrl_attention_config")
            ctx = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
            rmax = torch.empty(B, H, Sq, device=dev)
            rsum = torch.empty(B, H, Sq, device=dev)
            ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, mask, Sk, B, H, Sq, Sk, scale, H * 128, 128, H * 128)
            torch.cuda.synchronize()
            out[code] = (ctx.float(), rmax + torch.log(rsum))
    finally:
        _lib.load().bmhrl_attention_config(128, 0)
    assert float((out[14][0] - out[0][0]).abs().max()) < 2e-2 * float(out[0][0].abs().max())
    live = out[0][1] > -1e8
    assert float((out[14][1][live] - out[0][1][live]).abs().max()) < 2e-3
