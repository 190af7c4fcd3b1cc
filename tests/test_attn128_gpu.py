"""bmhrl_attention_shared128_fwd: absorbed-projection attention (head dim 128, one key/value tile for all heads)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 4, 64, 64), (2, 4, 200, 130), (3, 2, 256, 800), (1, 4, 800, 800)])
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
