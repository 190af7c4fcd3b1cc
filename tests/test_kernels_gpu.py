"""Kernel-level parity tests (GPU): each C-ABI entry point against a plain torch fp32/fp64 computation of the same
op on bf16-rounded inputs, or against the CPU oracle for the losses.  Run with `pytest -m gpu` on an MI355X."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import _lib
    _lib.load()  # must exist: the product has no fallback
    return torch.device("cuda:0")


def rel_err(a, b, floor=1e-6):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def bf(t):
    return t.to(torch.bfloat16)


def padded(t, ld=None):
    """copy a 2-D (or batched) bf16 tensor into a buffer whose last dim is padded to a multiple of 8"""
    from bmhrl_amd.ops import pad8
    ld = ld or pad8(t.shape[-1])
    out = torch.zeros(*t.shape[:-1], ld, dtype=torch.bfloat16, device=t.device)
    out[..., :t.shape[-1]] = t
    return out


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (480, 300, 364), (4096, 1024, 1024), (100, 64, 300), (257, 513, 72), (1000, 10172 // 4, 364)])
@pytest.mark.parametrize("a_trans,b_trans", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_layouts(dev, M, N, K, a_trans, b_trans):
    from bmhrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    Bm = bf(torch.randn(K, N, generator=g)).to(dev)  # logical (K, N), asymmetric random
    ref = A.float() @ Bm.float()
    As = padded(A.t().contiguous() if a_trans else A)
    Bs = padded(Bm.contiguous() if b_trans else Bm.t().contiguous())
    out = torch.full((M, N), float("nan"), device=dev)
    outb = torch.zeros(M, ops.pad8(N), dtype=torch.bfloat16, device=dev)
    ops.gemm(As, Bs, M, N, K, lda=As.shape[1], ldb=Bs.shape[1], a_trans=a_trans, b_trans=b_trans, C_f32=out, ldc=N,
             C_bf16=outb, ldcb=outb.shape[1])
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 2e-5
    assert rel_err(outb[:, :N].float(), ref) < 1e-2
    assert float(outb[:, N:].float().abs().max()) == 0 if outb.shape[1] > N else True


def test_gemm_ordered_k_split(dev):
    """bmhrl_gemm_desc.split_ws: a K split sums its partial tiles in split order (second launch) instead of fp32 atomics --
    the d cat[x, goal] product of the vocabulary head (480 x 364 over K = 10 172) and a batched pair product: same values as the
    atomic form up to the order of fp32 additions, bit-identical from launch to launch, `accumulate` allowed, C not zeroed."""
    from bmhrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    for (M, N, K, batch) in ((480, 364, 10176, 1), (480, 300, 1024, 2)):
        splits = ops.gemm_splits(M, N, K, batch)
        assert splits > 1
        A = bf(torch.randn(batch, M, K, generator=g)).to(dev)
        Bm = bf(torch.randn(batch, K, N, generator=g)).to(dev)
        ref = torch.einsum("bmk,bkn->bmn", A.float(), Bm.float())
        As, Bs = padded(A), padded(Bm)
        kw = dict(lda=As.shape[-1], ldb=Bs.shape[-1], b_trans=True, batch=(1, batch), a_strides=(0, M * As.shape[-1]),
                  b_strides=(0, K * Bs.shape[-1]), ldc=N, c_strides=(0, M * N), allow_split_k=True)
        atom = torch.zeros(batch, M, N, device=dev)
        ops.gemm(As, Bs, M, N, K, C_f32=atom, **kw)
        ws = torch.full((splits * batch * M * N,), float("nan"), device=dev)
        outs = []
        for _ in range(3):
            out = torch.full((batch, M, N), float("nan"), device=dev)          # (never zeroed: every element is stored)
            ops.gemm(As, Bs, M, N, K, C_f32=out, split_ws=ws, **kw)
            outs.append(out)
        torch.cuda.synchronize()
        assert rel_err(outs[0], ref) < 2e-5 and rel_err(outs[0], atom) < 2e-6
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        acc = torch.ones(batch, M, N, device=dev)
        ops.gemm(As, Bs, M, N, K, C_f32=acc, split_ws=ws, accumulate=True, alpha=0.5, **kw)
        assert rel_err(acc, 1 + 0.5 * ref) < 2e-5
        # a workspace that is too small is refused for `accumulate` (atomics cannot accumulate into a live C) and ignored otherwise
        small = torch.zeros(8, device=dev)
        out = torch.zeros(batch, M, N, device=dev)
        ops.gemm(As, Bs, M, N, K, C_f32=out, split_ws=small, **kw)
        assert rel_err(out, ref) < 2e-5
        with pytest.raises(RuntimeError):
            ops.gemm(As, Bs, M, N, K, C_f32=acc, split_ws=small, accumulate=True, **kw)


def test_expand_goals_backward_is_ordered(dev):
    """bmhrl_scatter_add_rows along an expand_goals row map: every row of dx written, sums in row order (bit-identical runs)"""
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(2)
    B, L, D = 7, 30, 64
    seg = (torch.rand(B, L, generator=g) < 0.25).to(torch.int32).to(dev)
    seg[3] = 0
    src = torch.empty(B * L, dtype=torch.int32, device=dev)
    ops.expand_goals_index(seg, src, B, L)
    dout = torch.randn(B * L, D, generator=g).to(dev)
    want = torch.zeros(B * L, D, dtype=torch.float64)
    s_cpu, d_cpu = src.cpu(), dout.cpu().double()
    for r in range(B * L):
        if int(s_cpu[r]) >= 0:
            want[int(s_cpu[r])] += d_cpu[r]
    runs = []
    for _ in range(2):
        dx = torch.full((B * L, D), float("nan"), device=dev)
        ops.scatter_add_rows(dout, src, dx, B * L, D)
        runs.append(dx)
    torch.cuda.synchronize()
    assert torch.equal(runs[0], runs[1])
    assert float((runs[0].cpu().double() - want).abs().max()) < 1e-5


def test_gemm_linear_epilogue(dev):
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 480, 300, 1024
    x = bf(torch.randn(M, K, generator=g)).to(dev)
    w = bf(torch.randn(N, K, generator=g) / 32).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    ops.gemm(x, w, M, N, K, lda=K, ldb=K, C_f32=out, ldc=N, bias=bias, relu=True, residual=res, ldr=N, alpha=0.5)
    ref = torch.relu(0.5 * (x.float() @ w.float().t()) + bias) + res
    assert rel_err(out, ref) < 2e-5
    out2 = out.clone()
    ops.gemm(x, w, M, N, K, lda=K, ldb=K, C_f32=out2, ldc=N, accumulate=True)
    assert rel_err(out2, out + x.float() @ w.float().t()) < 2e-5
    # dropout: kept elements scaled by 1/(1-p), mask reproducible from the seed
    d1 = torch.empty(M, N, device=dev); d2 = torch.empty(M, N, device=dev)
    ops.gemm(x, w, M, N, K, lda=K, ldb=K, C_f32=d1, ldc=N, dropout_p=0.25, seed=77)
    ops.gemm(x, w, M, N, K, lda=K, ldb=K, C_f32=d2, ldc=N, dropout_p=0.25, seed=77)
    plain = x.float() @ w.float().t()
    assert torch.equal(d1, d2)
    keep = d1 != 0
    assert 0.70 < float(keep.float().mean()) < 0.80
    assert rel_err(d1[keep], plain[keep] / 0.75) < 2e-5


def test_gemm_batched_attention_epilogues(dev):
    """Q.K^T with PROB epilogue, dO.V^T with DSCORE epilogue, P^T.dO, dS.K  (the attention backward products)."""
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, Sq, Sk, dk = 2, 3, 37, 50, 64
    D = H * dk
    Q = bf(torch.randn(B, Sq, D, generator=g)).to(dev)
    Kt = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    V = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    dO = bf(torch.randn(B, Sq, D, generator=g)).to(dev)
    mask = torch.ones(B, Sk, dtype=torch.uint8)
    mask[0, 45:] = 0
    mask = mask.to(dev)
    scale = 1 / math.sqrt(dk)
    qh = Q.float().view(B, Sq, H, dk).transpose(1, 2)
    kh = Kt.float().view(B, Sk, H, dk).transpose(1, 2)
    vh = V.float().view(B, Sk, H, dk).transpose(1, 2)
    doh = dO.float().view(B, Sq, H, dk).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    s = s.masked_fill(mask.view(B, 1, 1, Sk) == 0, -1e9)
    m = s.max(-1).values
    l = torch.exp(s - m[..., None]).sum(-1)
    P_ref = torch.softmax(s, -1)
    Skp = ops.pad8(Sk)
    P = torch.zeros(B, H, Sq, Skp, dtype=torch.bfloat16, device=dev)
    ops.gemm(Q, Kt, Sq, Sk, dk, lda=D, ldb=D, batch=(B, H), a_strides=(Sq * D, dk), b_strides=(Sk * D, dk), C_bf16=P,
             ldcb=Skp, cb_strides=(H * Sq * Skp, Sq * Skp), epilogue=ops.EPI_PROB, alpha=scale, mask=mask, mask_sb1=Sk,
             mask_sm=0, rowvec=m.contiguous(), rowvec2=l.contiguous(), rv_strides=(H * Sq, Sq))
    assert rel_err(P[..., :Sk].float(), P_ref) < 1e-2
    Pb = P[..., :Sk].float()
    o = Pb @ vh
    delta = (doh * o).sum(-1).contiguous()
    dS = torch.zeros_like(P)
    ops.gemm(dO, V, Sq, Sk, dk, lda=D, ldb=D, batch=(B, H), a_strides=(Sq * D, dk), b_strides=(Sk * D, dk), C_bf16=dS,
             ldcb=Skp, cb_strides=(H * Sq * Skp, Sq * Skp), epilogue=ops.EPI_DSCORE, alpha=scale, rowvec=delta,
             rv_strides=(H * Sq, Sq), aux=P, ldaux=Skp, aux_strides=(H * Sq * Skp, Sq * Skp))
    dS_ref = Pb * (doh @ vh.transpose(-1, -2) - delta[..., None]) * scale
    assert rel_err(dS[..., :Sk].float(), dS_ref) < 1e-2
    # with the key mask: the -1e9 fill is a constant (masked_fill), so a masked key's score gets NO gradient -- also in a fully
    # masked row, where P is uniform instead of zero (torch autograd of the reference's op chain is the check)
    mask2 = mask.clone()
    mask2[1, :] = 0
    sq = ((qh @ kh.transpose(-1, -2)) * scale).requires_grad_(True)
    p2 = torch.softmax(sq.masked_fill(mask2.view(B, 1, 1, Sk) == 0, -1e9), -1)
    (p2 @ vh).backward(doh)
    P2 = torch.zeros_like(P)
    P2[..., :Sk] = bf(p2.detach())
    delta2 = (doh * (P2[..., :Sk].float() @ vh)).sum(-1).contiguous()
    for fast in (True, False):                           # 16-byte epilogue and the generic one (odd leading dimension of aux)
        ldp = Skp if fast else Skp + 4
        Pin = torch.zeros(B, H, Sq, ldp, dtype=torch.bfloat16, device=dev)
        Pin[..., :Sk] = P2[..., :Sk]
        dS2 = torch.zeros_like(P)
        ops.gemm(dO, V, Sq, Sk, dk, lda=D, ldb=D, batch=(B, H), a_strides=(Sq * D, dk), b_strides=(Sk * D, dk), C_bf16=dS2,
                 ldcb=Skp, cb_strides=(H * Sq * Skp, Sq * Skp), epilogue=ops.EPI_DSCORE, alpha=scale, rowvec=delta2,
                 rv_strides=(H * Sq, Sq), aux=Pin, ldaux=ldp, aux_strides=(H * Sq * ldp, Sq * ldp), mask=mask2, mask_sb1=Sk, mask_sm=0)
        assert float(dS2[1].abs().max()) == 0.0 and float(sq.grad[1].abs().max()) == 0.0
        assert rel_err(dS2[..., :Sk].float(), sq.grad * scale) < 1.5e-2       # (dS is taken w.r.t. the unscaled product)
    # dV = P^T dO  (both operands transposed: reduction runs over rows of P and dO)
    dV = torch.empty(B, Sk, D, device=dev)
    ops.gemm(P, dO, Sk, dk, Sq, lda=Skp, ldb=D, a_trans=True, b_trans=True, batch=(B, H),
             a_strides=(H * Sq * Skp, Sq * Skp), b_strides=(Sq * D, dk), C_f32=dV, ldc=D, c_strides=(Sk * D, dk))
    dV_ref = (Pb.transpose(-1, -2) @ doh).transpose(1, 2).reshape(B, Sk, D)
    assert rel_err(dV, dV_ref) < 2e-5
    # dQ = dS K  (B operand read transposed)
    dQ = torch.empty(B, Sq, D, device=dev)
    ops.gemm(dS, Kt, Sq, dk, Sk, lda=Skp, ldb=D, b_trans=True, batch=(B, H), a_strides=(H * Sq * Skp, Sq * Skp),
             b_strides=(Sk * D, dk), C_f32=dQ, ldc=D, c_strides=(Sq * D, dk))
    dQ_ref = (dS[..., :Sk].float() @ kh).transpose(1, 2).reshape(B, Sq, D)
    assert rel_err(dQ, dQ_ref) < 2e-5


def _attn_ref(Q, K, V, mask, H, scale):
    B, Sq, D = Q.shape
    Sk = K.shape[1]
    dk = D // H
    qh = Q.double().view(B, Sq, H, dk).transpose(1, 2)
    kh = K.double().view(B, Sk, H, dk).transpose(1, 2)
    vh = V.double().view(B, Sk, H, dk).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(mask.view(B, 1, -1, Sk) == 0, -1e9)
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, D), s


@pytest.mark.parametrize("B,H,Sq,Sk,kind", [(2, 4, 64, 64, "none"), (2, 4, 70, 100, "pad"), (3, 2, 30, 37, "causal"),
                                            (2, 4, 256, 800, "pad"), (2, 4, 800, 256, "pad"), (1, 4, 33, 130, "allmasked"),
                                            (2, 4, 64, 20, "pad"), (3, 3, 100, 1500, "none")])
def test_attention_fwd(dev, B, H, Sq, Sk, kind):
    from bmhrl_amd import ops
    dk = 256
    D = H * dk
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    Q = bf(torch.randn(B, Sq, D, generator=g)).to(dev)
    K = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    V = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    mask, sb, sq = None, 0, 0
    if kind == "pad":
        mask = torch.ones(B, 1, Sk, dtype=torch.uint8)
        mask[0, 0, Sk - Sk // 3:] = 0
        mask[-1, 0, 5:9] = 0
        sb, sq = Sk, 0
    elif kind == "causal":
        mask = torch.tril(torch.ones(Sq, Sk, dtype=torch.uint8)).repeat(B, 1, 1)
        mask[1, :, 20:] = 0
        sb, sq = Sq * Sk, Sk
    elif kind == "allmasked":
        mask = torch.zeros(B, 1, Sk, dtype=torch.uint8)
        sb, sq = Sk, 0
    if mask is not None:
        mask = mask.to(dev).contiguous()
    O = torch.zeros(B, Sq, D, dtype=torch.bfloat16, device=dev)
    rmax = torch.empty(B, H, Sq, device=dev)
    rsum = torch.empty(B, H, Sq, device=dev)
    scale = 1 / math.sqrt(dk)
    # spike one key so the running max jumps in a late tile (forces the rescale branch)
    K[0, Sk - 1] = K[0, Sk - 1] * 6
    ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, sb, sq, B, H, Sq, Sk, dk, scale, D, D, D, D)
    torch.cuda.synchronize()
    ref, s = _attn_ref(Q, K, V, mask, H, scale)
    assert rel_err(O.float(), ref) < 1.5e-2  # bf16 P and bf16 output
    # (row_max, row_sum) are any consistent pair with P = exp(s - row_max) / row_sum (the kernel keeps a running max
    # that may lag the true one by a bounded amount); compare the log-sum-exp and the reconstructed probabilities
    lse_ref = torch.logsumexp(s, -1)
    lse = rmax.double().cpu() + torch.log(rsum.double().cpu())
    assert float((lse - lse_ref.cpu()).abs().max()) < 2e-3 * max(1.0, float(lse_ref.abs().max()) * 1e-6 + 1.0)
    p_rec = torch.exp(s.cpu() - rmax.double().cpu()[..., None]) / rsum.double().cpu()[..., None]
    assert float((p_rec - torch.softmax(s, -1).cpu()).abs().max()) < 2e-3
    assert float((s.max(-1).values.cpu() - rmax.double().cpu()).max()) < 8 * 0.6932 + 1e-3   # lag bound (2^8)
    if kind == "allmasked":  # uniform attention over all Sk keys (reference fills -1e9, not -inf)
        assert rel_err(O.float(), V.float().view(B, Sk, D).mean(1, keepdim=True).expand(B, Sq, D)) < 1.5e-2


def test_attention_fwd_output_dropout(dev):
    """dropout on the attention output (model/multihead_attention.py:27-28): every element is either dropped or the
    undropped value / (1 - p); the pattern follows the seed"""
    from bmhrl_amd import ops
    B, H, Sq, Sk, dk = 2, 4, 96, 160, 256
    D = H * dk
    g = torch.Generator().manual_seed(11)
    Q = bf(torch.randn(B, Sq, D, generator=g)).to(dev)
    K = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    V = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    rmax = torch.empty(B, H, Sq, device=dev); rsum = torch.empty(B, H, Sq, device=dev)
    outs = []
    for p, seed in ((0.0, 0), (0.25, 5), (0.25, 5), (0.25, 6)):
        O = torch.zeros(B, Sq, D, dtype=torch.bfloat16, device=dev)
        ops.attention_fwd(Q, K, V, O, rmax, rsum, None, 0, 0, B, H, Sq, Sk, dk, 1 / 16, D, D, D, D, dropout_p=p, seed=seed)
        outs.append(O.float())
    base, d1, d1b, d2 = outs
    assert torch.equal(d1, d1b) and not torch.equal(d1, d2)
    kept = d1 != 0
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - 0.25) < 0.01
    assert float((d1[kept] - base[kept] / 0.75).abs().max()) < 2e-2 * float(base.abs().max()) / 0.75


def test_dropout_mask_statistics(dev):
    """the counter-based dropout bits: keep rate, no correlation between neighbouring elements, rows, or seeds"""
    from bmhrl_amd import ops
    rows, cols = 2048, 1024
    x = torch.ones(rows, cols, device=dev)
    masks = []
    for p, seed in ((0.1, 1), (0.1, 2), (0.5, 1)):
        y = ops.bf16_zeros(rows, cols, dev)
        ops.cast_bf16(x, cols, y, cols, rows, cols, dropout_p=p, seed=seed)
        m = (y.float() != 0)
        assert abs(m.float().mean().item() - (1 - p)) < 2e-3
        assert float((y.float()[m] - 1 / (1 - p)).abs().max()) < 1e-2
        masks.append(m.float())
    a, b, c = masks
    def corr(u, v):
        u = u - u.mean(); v = v - v.mean()
        return float((u * v).mean() / (u.std() * v.std()))
    assert abs(corr(a, b)) < 5e-3                                  # different seeds
    assert abs(corr(a[:, 1:], a[:, :-1])) < 5e-3                   # neighbouring columns
    assert abs(corr(a[1:], a[:-1])) < 5e-3                         # neighbouring rows
    assert abs(corr(a[:, ::2], a[:, 1::2])) < 5e-3
    assert float((a.mean(0) - 0.9).abs().max()) < 0.04 and float((a.mean(1) - 0.9).abs().max()) < 0.05   # no dead rows / columns


def test_softmax_rows_and_delta(dev):
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(3)
    S = torch.randn(90, 37, generator=g).to(dev) * 3
    S[5, 10:] = -1e9
    P = torch.zeros(90, 40, dtype=torch.bfloat16, device=dev)
    ops.softmax_rows(S, 37, P, 40, 90, 37)
    assert rel_err(P[:, :37].float(), torch.softmax(S, -1)) < 1e-2
    B, H, Sq, dk = 2, 2, 9, 64
    dO = bf(torch.randn(B, Sq, H * dk, generator=g)).to(dev)
    O = bf(torch.randn(B, Sq, H * dk, generator=g)).to(dev)
    delta = torch.empty(B, H, Sq, device=dev)
    ops.attn_delta(dO, H * dk, O, H * dk, delta, B, H, Sq, dk)
    ref = (dO.float() * O.float()).view(B, Sq, H, dk).sum(-1).transpose(1, 2)
    assert rel_err(delta, ref) < 1e-5


def test_softmax_bwd_rows(dev):
    """dS = scale * P * (dP - sum_k P dP) with the row term from the SAME rounded P, 0 at masked keys (masked_fill passes no
    gradient): rows (sample, query, head), per-sample and per-query masks."""
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(4)
    B, L, H, Sk = 3, 5, 2, 37
    Skp = 40
    P = torch.zeros(B, L, H, Skp, dtype=torch.bfloat16)
    P[..., :Sk] = bf(torch.softmax(torch.randn(B, L, H, Sk, generator=g) * 2, -1))
    dP = torch.randn(B, L, H, Skp, generator=g) + 5.0          # a large common component: what the consistent row term cancels
    for per_query in (False, True):
        mask = torch.rand(B, L if per_query else 1, Sk, generator=g) < 0.8
        mask[1] = False                                          # a fully masked sample: P is uniform there, dS must be 0
        dS = torch.zeros(B, L, H, Skp, dtype=torch.bfloat16, device=dev)
        m8 = mask.to(dev).contiguous()
        ops.softmax_bwd_rows(P.to(dev), Skp, dP.to(dev), Skp, dS, Skp, B * L * H, Sk, 0.25, m8, m8.shape[1] * Sk,
                             Sk if per_query else 0, H, L)
        Pf, dPf = P[..., :Sk].double(), dP[..., :Sk].double()
        ref = 0.25 * Pf * (dPf - (Pf * dPf).sum(-1, keepdim=True))
        ref = ref * mask.unsqueeze(2).expand(B, L, H, Sk) if per_query else ref * mask.view(B, 1, 1, Sk)
        assert rel_err(dS[..., :Sk].float(), ref) < 5e-3          # bf16 output rounding
        assert float(dS[1].abs().max()) == 0.0 and float(dS[..., Sk:].abs().max()) == 0.0
    dS = torch.zeros(B, L, H, Skp, dtype=torch.bfloat16, device=dev)
    ops.softmax_bwd_rows(P.to(dev), Skp, dP.to(dev), Skp, dS, Skp, B * L * H, Sk, 1.0)
    ref = Pf * (dPf - (Pf * dPf).sum(-1, keepdim=True))
    assert rel_err(dS[..., :Sk].float(), ref) < 5e-3
    # the rows of dS sum to ~0 although P is rounded (sum P != 1 is the only residue)
    assert float(dS[..., :Sk].float().sum(-1).abs().max()) < 2e-2 * float(dS.float().abs().max())


def test_split_operand_gemm(dev):
    """[x_hi | x_hi | x_lo] x [W_hi | W_lo | W_hi]^T (ops.cast_split3_bf16, one GEMM with K = 3 part) reproduces the fp32
    product to ~1e-5 where plain bf16 operands give ~3e-3 -- the vocabulary projection of WorkerHeadFn."""
    from bmhrl_amd import ops
    from bmhrl_amd.functional import ShadowCache
    g = torch.Generator().manual_seed(8)
    rows, K1, K2, N = 96, 300, 64, 1000
    K = K1 + K2
    x1, x2 = torch.randn(rows, K1, generator=g).to(dev), torch.randn(rows, K2, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    part = ShadowCache.split_part(K)
    assert part == 384
    xb = torch.zeros(rows, 3 * part, dtype=torch.bfloat16, device=dev)
    wb = torch.zeros(N, 3 * part, dtype=torch.bfloat16, device=dev)
    ops.cast_split3_bf16(x1, K1, xb, 3 * part, part, 2, rows, K1)
    ops.cast_split3_bf16(x2, K2, xb, 3 * part, part, 2, rows, K2, y_off=K1)
    ops.cast_split3_bf16(w, K, wb, 3 * part, part, 1, N, K)
    both = torch.zeros_like(xb)                               # the two sources in ONE launch == the two launches
    ops.cast_split3_bf16(x1, K1, both, 3 * part, part, 2, rows, K1, x2=x2, ldx2=K2, cols2=K2)
    assert torch.equal(both, xb)
    x = torch.cat([x1, x2], -1)
    hi = bf(x)
    assert torch.equal(xb[:, :K], hi) and torch.equal(xb[:, part:part + K], hi)
    assert torch.equal(xb[:, 2 * part:2 * part + K], bf(x - hi.float())) and float(xb[:, K:part].abs().max()) == 0.0
    assert torch.equal(wb[:, part:part + K], bf(w - bf(w).float())) and torch.equal(wb[:, 2 * part:2 * part + K], bf(w))
    y = torch.empty(rows, N, device=dev)
    ops.gemm(xb, wb, rows, N, 3 * part, lda=3 * part, ldb=3 * part, C_f32=y, ldc=N)
    ref = x.double() @ w.double().T
    plain = bf(x).double() @ bf(w).double().T
    assert rel_err(y, ref) < 3e-5 and rel_err(plain, ref) > 1e-3


@pytest.mark.parametrize("rows,D", [(4096, 1024), (12800, 128), (480, 300), (33, 20), (1001, 128), (7, 64), (130, 96)])
def test_layernorm(dev, rows, D):
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    dy = torch.randn(rows, D, generator=g).to(dev)
    yb = ops.bf16_zeros(rows, D, dev)
    yf = torch.empty(rows, D, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    ops.layernorm_fwd(x, gamma, beta, yb, yb.shape[1], yf, mean, rstd, rows, D)
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    ref.backward(dy.double())
    assert rel_err(yf, ref) < 1e-5
    assert rel_err(yb[:, :D].float(), ref) < 1e-2
    dx = torch.empty(rows, D, device=dev)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    ones = torch.ones(rows, D, device=dev)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, ones, dg, db, rows, D)
    assert rel_err(dx - 1, xr.grad) < 2e-5
    assert rel_err(dg, gr.grad) < 2e-5
    assert rel_err(db, br.grad) < 2e-5


def test_posenc_embed_cast_colsum(dev):
    from bmhrl_amd import ops
    from oracle.bmhrl_oracle import posenc_table
    g = torch.Generator().manual_seed(4)
    B, S, D = 3, 11, 20
    pe = posenc_table(64, D).float().to(dev)
    a = torch.rand(B, S, D, generator=g).to(dev); b = torch.rand(B, S, D, generator=g).to(dev)
    out = torch.empty(B, S, D, device=dev)
    ob = ops.bf16_zeros(B * S, D, dev)
    ops.add_posenc(a, b, pe, out, ob, ob.shape[1], B, S, D)
    ref = a + b + pe[:S].unsqueeze(0)
    assert rel_err(out, ref) < 1e-6
    assert rel_err(ob[:, :D].float().view(B, S, D), ref) < 1e-2
    V = 17
    table = torch.randn(V, D, generator=g).to(dev)
    tok = torch.randint(0, V, (B, S), generator=g).to(dev); tok2 = torch.randint(0, V, (B, S), generator=g).to(dev)
    emb = torch.empty(B, S, D, device=dev); out = torch.empty(B, S, D, device=dev)
    sc = math.sqrt(D)
    ops.embed_posenc(tok, tok2, 0.25, table, pe, emb, out, B, S, D, sc)
    e_ref = table[tok] * sc * 0.75 + table[tok2] * sc * 0.25
    assert rel_err(emb, e_ref) < 1e-6 and rel_err(out, e_ref + pe[:S]) < 1e-6
    dC = torch.randn(B, S, D, generator=g).to(dev)
    dt = torch.zeros(V, D, device=dev)
    ops.embed_bwd(tok, tok2, 0.25, dC, dt, B, S, D, sc)
    dref = torch.zeros(V, D, device=dev)
    dref.index_add_(0, tok.reshape(-1), dC.reshape(-1, D) * sc * 0.75)
    dref.index_add_(0, tok2.reshape(-1), dC.reshape(-1, D) * sc * 0.25)
    assert rel_err(dt, dref) < 1e-5
    x = torch.randn(500, 300, generator=g).to(dev)
    y = ops.bf16_zeros(500, 300, dev)
    ops.cast_bf16(x, 300, y, y.shape[1], 500, 300, 2.0)
    assert torch.equal(y[:, :300], (x * 2).to(torch.bfloat16)) and float(y[:, 300:].float().abs().max()) == 0
    db = torch.empty(300, device=dev)
    ops.colsum_bf16(y, y.shape[1], db, False, 500, 300)
    assert rel_err(db, y[:, :300].float().sum(0)) < 1e-5


def test_gate_expand_gather(dev, golden):
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, D = 96, 20
    cv = torch.randn(rows, D, generator=g).to(dev); ca = torch.randn(rows, D, generator=g).to(dev)
    for a0 in (0.3, -2.5):
        a = torch.tensor([a0], device=dev)
        out = torch.empty(rows, D, device=dev)
        ops.gate_fwd(cv, ca, a, out, None, 0, rows, D)
        ar = a.clone().requires_grad_(True); cvr = cv.clone().requires_grad_(True); car = ca.clone().requires_grad_(True)
        gt = torch.sigmoid(torch.clamp(ar, -2, 2))
        ref = gt * cvr + (1 - gt) * car
        assert rel_err(out, ref) < 1e-6
        dout = torch.randn(rows, D, generator=g).to(dev)
        ref.backward(dout)
        dcv = torch.empty_like(cv); dca = torch.empty_like(ca); da = torch.zeros(1, device=dev)
        ops.gate_bwd(dout, cv, ca, a, dcv, dca, da, rows, D)
        assert rel_err(dcv, cvr.grad) < 1e-6 and rel_err(dca, car.grad) < 1e-6
        assert abs(float(da) - float(ar.grad)) < 1e-4 * max(1.0, abs(float(ar.grad)))
    k = golden("kat")
    gin = torch.from_numpy(k["expand_in"]).to(dev); seg = torch.from_numpy(k["expand_seg"]).to(dev)
    B, L, Dg = gin.shape
    src = torch.empty(B * L, dtype=torch.int32, device=dev)
    ops.expand_goals_index(seg.contiguous(), src, B, L)
    out = torch.empty(B * L, Dg, device=dev)
    ops.gather_rows(gin.reshape(B * L, Dg).contiguous(), src, out, None, 0, B * L, Dg)
    assert np.array_equal(out.view(B, L, Dg).cpu().numpy(), k["expand_out"])
    dx = torch.zeros(B * L, Dg, device=dev)
    ops.scatter_add_rows(torch.ones(B * L, Dg, device=dev), src, dx, B * L, Dg)
    exp = torch.zeros(B * L); s = src.cpu()
    for i in range(B * L):
        if s[i] >= 0:
            exp[s[i]] += 1
    assert torch.equal(dx[:, 0].cpu(), exp)
    # random label patterns against the oracle's loop (incl. empty rows, row 0 empty, all empty)
    from oracle.bmhrl_oracle import expand_goals
    rg = np.random.default_rng(0)
    for trial in range(20):
        B, L = 6, 9
        seg = torch.from_numpy((rg.random((B, L)) < [0.0, 0.1, 0.3, 0.5][trial % 4]).astype(np.int32))
        if trial % 5 == 0:
            seg[0] = 0
        x = torch.randn(B, L, 3)
        src = torch.empty(B * L, dtype=torch.int32, device=dev)
        ops.expand_goals_index(seg.to(dev), src, B, L)
        out = torch.empty(B * L, 3, device=dev)
        ops.gather_rows(x.reshape(B * L, 3).to(dev), src, out, None, 0, B * L, 3)
        assert torch.equal(out.cpu().view(B, L, 3), expand_goals(x, seg))
        # the one-launch form: same row map, same rows, bf16 copy in a padded buffer
        src2 = torch.empty(B * L, dtype=torch.int32, device=dev)
        out2 = torch.empty(B * L, 3, device=dev)
        ob = torch.zeros(B * L, 8, dtype=torch.bfloat16, device=dev)
        ops.expand_goals(seg.to(dev), x.reshape(B * L, 3).to(dev), src2, out2, ob, 8, B, L, 3)
        assert torch.equal(src2, src) and torch.equal(out2, out)
        assert torch.equal(ob[:, :3].float(), out.to(torch.bfloat16).float()) and float(ob[:, 3:].abs().max()) == 0.0


def test_loss_kernels_vs_oracle(dev, golden):
    from bmhrl_amd import ops
    from oracle import bmhrl_oracle as O
    g = golden("losses")
    logits = torch.from_numpy(g["logits"]); trg = torch.from_numpy(g["trg"]); sampled = torch.from_numpy(g["sampled"])
    score = torch.from_numpy(g["score"]); baseline = torch.from_numpy(g["baseline"])
    B, S, V = logits.shape
    rows = B * S
    lp = logits.clone().to(dev).reshape(rows, V).contiguous()
    ops.log_softmax_(lp, V, rows, V)
    assert rel_err(lp, torch.log_softmax(logits, -1).reshape(rows, V)) < 1e-6
    t = trg.reshape(-1).to(dev)
    row_loss = torch.empty(rows, device=dev)
    ops.smooth_kl_fwd(lp, V, t, None, None, None, 0.7, 1, -1, row_loss, None, rows, V)
    assert rel_err(row_loss, torch.from_numpy(g["ls"]).sum(-1)) < 1e-5
    n_tok = float((trg != 1).sum())
    scale = torch.tensor([1.0 / n_tok], device=dev)
    gf = torch.empty(rows, V, device=dev)
    gb = ops.bf16_zeros(rows, V, dev)
    ops.smooth_kl_bwd(lp, V, t, None, None, None, 0.7, 1, -1, scale, gb, gb.shape[1], gf, rows, V)
    assert rel_err(gf, torch.from_numpy(g["ls_grad_logits"]).reshape(rows, V)) < 1e-5
    assert rel_err(gb[:, :V].float(), gf) < 1e-2
    # the same gradient in two steps: d/d log-probs, then the log-softmax backward kernel
    glp = torch.empty(rows, V, device=dev)
    ops.smooth_kl_bwd(lp, V, t, None, None, None, 0.7, 1, -1, scale, None, 0, glp, rows, V, wrt_logits=False)
    g2 = ops.bf16_zeros(rows, V, dev)
    ops.log_softmax_bwd(glp, lp, V, g2, g2.shape[1], rows, V)
    assert rel_err(g2[:, :V].float(), gf) < 1e-2
    a = sampled.reshape(-1).to(dev)
    mask = (trg != 1)
    n_row = mask.sum(-1, keepdim=True).expand(B, S).reshape(-1).float().to(dev)
    for stab, tag in ((False, "raw"), (True, "stab")):
        sc = ((score - baseline) * mask.float() if stab else score).reshape(-1).to(dev).contiguous()
        amp = torch.empty(rows, device=dev)
        ops.smooth_kl_fwd(lp, V, t, a, sc, n_row, 0.7, 1, -1, row_loss, amp, rows, V)
        assert rel_err(amp, torch.from_numpy(g[f"bkl_{tag}_amp"]).reshape(-1)) < 1e-5
        assert rel_err(row_loss, torch.from_numpy(g[f"bkl_{tag}"]).sum(-1)) < 1e-5
        scale = torch.tensor([1.0 / (n_tok * 0.2)], device=dev)
        ops.smooth_kl_bwd(lp, V, t, a, sc, n_row, 0.7, 1, -1, scale, None, 0, gf, rows, V)
        assert rel_err(gf, torch.from_numpy(g[f"bkl_{tag}_grad_logits"]).reshape(rows, V)) < 1e-5
    # the idx.sum()>0 guard (only padded flat index is 0)
    k = golden("kat")
    lp4 = torch.from_numpy(k["a4_lp"]).reshape(6, 6).to(dev).contiguous()
    rl = torch.empty(6, device=dev)
    ops.smooth_kl_fwd(lp4, 6, torch.tensor([1, 4, 2, 5, 3, 2], device=dev), None, None, None, 0.7, 1, -1, rl, None, 6, 6)
    assert rel_err(rl, torch.from_numpy(k["a4_ls_guard"]).sum(-1)) < 1e-5
    ops.smooth_kl_fwd(lp4, 6, torch.tensor([2, 4, 1, 5, 3, 2], device=dev), None, None, None, 0.7, 1, -1, rl, None, 6, 6)
    assert rel_err(rl, torch.from_numpy(k["a4_ls"]).sum(-1)) < 1e-5
    # reinforce terms
    act = torch.tensor([2, 0, 3, 1, 3, 4], device=dev)
    val = torch.tensor([.1, .2, .3, .4, .5, .6], device=dev); cv = torch.tensor([.3, .1, .0, .2, .2, .9], device=dev)
    rp = torch.empty(6, device=dev); rv = torch.empty(6, device=dev)
    ops.reinforce_fwd(lp4, 6, act, val, cv, rp, rv, 6, 6)
    assert abs(float(rp.mean() + rv.mean()) - float(k["a4_reinforce"])) < 1e-6
    # module form (probabilities in, as the reference) with gradients, against the reference's values
    from bmhrl_amd.loss.biased_kl import Reinforce
    x = torch.from_numpy(g["logits"]).to(dev).requires_grad_(True)
    val = score.to(dev).requires_grad_(True)
    r = Reinforce()(torch.softmax(x, -1), sampled.to(dev), val, baseline.to(dev))
    r.backward()
    assert abs(float(r) - float(g["reinforce"])) < 1e-5 * max(1.0, abs(float(g["reinforce"])))
    assert rel_err(x.grad, torch.from_numpy(g["reinforce_grad_logits"])) < 1e-5


@pytest.mark.parametrize("rows,V", [(480, 10172), (7, 10171), (33, 1024), (5, 4100), (3, 12288), (2, 12292), (9, 52)])
def test_log_softmax_fwd_bwd_paths(dev, rows, V):
    """register-resident rows (V % 4 == 0, V <= 12 288) and the strided fallback (odd or longer rows)"""
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + V)
    logits = (3 * torch.randn(rows, V, generator=g)).to(dev)
    lp = logits.clone()
    ops.log_softmax_(lp, V, rows, V)
    ref = torch.log_softmax(logits.double(), -1)
    assert float((lp.double() - ref).abs().max()) < 2e-5
    dlogp = torch.randn(rows, V, generator=g).to(dev)
    gb = ops.bf16_zeros(rows, V, dev)
    ops.log_softmax_bwd(dlogp, lp, V, gb, gb.shape[1], rows, V)
    want = dlogp.double() - ref.exp() * dlogp.double().sum(-1, keepdim=True)
    assert float((gb[:, :V].double() - want).abs().max()) < 1e-2 * float(want.abs().max())
    assert float(gb[:, V:].float().abs().sum()) == 0.0          # padding columns untouched


def test_sampling(dev):
    from bmhrl_amd import ops
    V, rows = 1000, 4096
    logits = torch.zeros(rows, V)
    logits[:, 7] = math.log(300.0)
    logits[:, 500] = math.log(700.0)
    logits[:, [i for i in range(V) if i not in (7, 500)]] = -30.0
    lp = torch.log_softmax(logits, -1).to(dev).contiguous()
    out = torch.empty(rows, dtype=torch.int64, device=dev); p = torch.empty(rows, device=dev)
    ops.sample_tokens(lp, V, out, p, rows, V, False, 123)
    f7 = float((out == 7).float().mean()); f500 = float((out == 500).float().mean())
    assert abs(f7 - 0.3) < 0.03 and abs(f500 - 0.7) < 0.03
    assert rel_err(p, torch.exp(lp[torch.arange(rows), out])) < 1e-6
    out2 = torch.empty_like(out)
    ops.sample_tokens(lp, V, out2, None, rows, V, False, 123)
    assert torch.equal(out, out2)
    lp2 = torch.log_softmax(torch.randn(64, 10172), -1).to(dev)
    o3 = torch.empty(64, dtype=torch.int64, device=dev)
    ops.sample_tokens(lp2, 10172, o3, None, 64, 10172, True, 0)
    assert torch.equal(o3, lp2.argmax(-1))


def test_adam_matches_torch(dev):
    from bmhrl_amd import ops
    g = torch.Generator().manual_seed(6)
    n = 100003
    p0 = torch.randn(n, generator=g).to(dev)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.01)
    p = p0.clone(); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g).to(dev)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    assert rel_err(p, ref.data) < 1e-6


def test_make_masks_kernel_equals_the_reference_functions(dev):
    """bmhrl_make_masks == model/masking.py make_masks (V_mask from rgb column 0, A_mask from audio column 0, C_mask = key
    padding & lower triangle), single and doubled"""
    from bmhrl_amd import ops, synthetic as syn
    from bmhrl_amd.model.masking import make_masks
    b = syn.synthetic_batch(5, 37, 90, 11, 60, seed=3, min_len=3)
    fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
    fs["rgb"][2, 5, 0] = 0.0                       # a zero first feature inside the clip masks that frame too
    trg = b["captions"][:, :-1].to(dev)
    ref = make_masks(fs, trg, "audio_video", 1)
    for copies in (1, 2):
        vm, am, cm = ops.make_masks(fs["rgb"], fs["audio"], trg, 1, copies=copies)
        for got, want in ((vm, ref["V_mask"]), (am, ref["A_mask"]), (cm, ref["C_mask"])):
            assert got.dtype == torch.bool and got.shape[0] == copies * 5
            assert torch.equal(got, torch.cat([want] * copies))
    assert not bool(ref["V_mask"].all()) and not bool(ref["C_mask"][:, -1].all())      # padding is present in the case


@pytest.mark.parametrize("rows,D", [(480, 300), (37, 128), (64, 1024), (2100, 300)])
def test_layernorm_groups_equal_one_launch_per_group(dev, rows, D):
    """bmhrl_layernorm_fwd_groups / _bwd_groups (one launch for the worker and the manager half of a paired block) == the
    single-group entry points on each half, bit for bit in the forward and in dx; column sums up to the order of the atomics"""
    from bmhrl_amd import ops
    torch.manual_seed(rows + D)
    G = 2
    x = torch.randn(G, rows, D, device=dev) * 2 + 0.3
    gam, bet = torch.randn(G, D, device=dev), torch.randn(G, D, device=dev)
    ld = (D + 7) & ~7
    yb, yf = torch.zeros(G * rows, ld, dtype=torch.bfloat16, device=dev), torch.empty(G, rows, D, device=dev)
    mean, rstd = torch.empty(G * rows, device=dev), torch.empty(G * rows, device=dev)
    ops.layernorm_fwd_groups(x, gam, bet, yb, ld, yf, mean, rstd, rows, D, G)
    yb1, yf1 = torch.zeros_like(yb), torch.empty_like(yf)
    mean1, rstd1 = torch.empty_like(mean), torch.empty_like(rstd)
    for g in range(G):
        ops.layernorm_fwd(x[g], gam[g], bet[g], yb1[g * rows:], ld, yf1[g], mean1[g * rows:], rstd1[g * rows:], rows, D)
    assert torch.equal(yb, yb1) and torch.equal(yf, yf1) and torch.equal(mean, mean1) and torch.equal(rstd, rstd1)
    assert rel_err(yf, torch.nn.functional.layer_norm(x, (D,)) * gam[:, None] + bet[:, None]) < 1e-5
    dy, add = torch.randn(G, rows, D, device=dev), torch.randn(G, rows, D, device=dev)
    dx, dg, db = torch.empty_like(x), torch.zeros(G, D, device=dev), torch.zeros(G, D, device=dev)
    ops.layernorm_bwd_groups(dy, x, gam, mean, rstd, dx, add, dg, db, rows, D, G)
    dx1, dg1, db1 = torch.empty_like(x), torch.zeros(G, D, device=dev), torch.zeros(G, D, device=dev)
    for g in range(G):
        ops.layernorm_bwd(dy[g], x[g], gam[g], mean[g * rows:], rstd[g * rows:], dx1[g], add[g], dg1[g], db1[g], rows, D)
    assert torch.equal(dx, dx1)
    assert rel_err(dg, dg1) < 1e-5 and rel_err(db, db1) < 1e-5


def test_gemm_group_equals_separate_launches(dev):
    """bmhrl_gemm_group: up to four weight-gradient style products (dY^T X, fp32 outputs, K no multiple of 64, different shapes,
    batches and K splits) as one launch == the same problems launched one by one; a mix that cannot share a kernel (a
    128-tile problem among them) takes the one-by-one path with the same results"""
    from bmhrl_amd import ops
    torch.manual_seed(21)
    R = 480
    shapes = [(300, 1024, 2), (256, 128, 8), (1024, 300, 2), (64, 72, 1), (3072, 300, 2)]
    def problems(big_one):
        out = []
        for (N, K, nb) in shapes[:5]:
            dy = (torch.randn(nb * R, ((N + 7) // 8) * 8, device=dev) * 0.3).bfloat16()
            x = (torch.randn(nb * R, ((K + 7) // 8) * 8, device=dev) * 0.3).bfloat16()
            out.append((dy, x, N, K, nb))
        if big_one:
            dy = (torch.randn(4096, 1024, device=dev) * 0.1).bfloat16()
            x = (torch.randn(4096, 1024, device=dev) * 0.1).bfloat16()
            out.append((dy, x, 1024, 1024, 1))
        return out
    for big_one in (False, True):
        probs = problems(big_one)
        got, ref, leaf = [], [], []
        for dy, x, N, K, nb in probs:
            rows = dy.shape[0] // nb
            kw = dict(lda=dy.shape[1], ldb=x.shape[1], a_trans=True, b_trans=True, batch=(1, nb), a_strides=(0, rows * dy.shape[1]),
                      b_strides=(0, rows * x.shape[1]), ldc=K, c_strides=(0, N * K), allow_split_k=True)
            c1, c2 = torch.zeros(nb * N, K, device=dev), torch.zeros(nb * N, K, device=dev)
            ops.gemm(dy, x, N, K, rows, C_f32=c1, defer=leaf, **kw)
            ops.gemm(dy, x, N, K, rows, C_f32=c2, **kw)
            got.append(c1); ref.append(c2)
        assert all(float(c.abs().max()) == 0.0 for c in got)          # nothing ran yet
        ops.gemm_flush(leaf)
        assert leaf == []
        for (dy, x, N, K, nb), a, b in zip(probs, got, ref):
            rows = dy.shape[0] // nb
            exact = torch.einsum("brn,brk->bnk", dy.float().view(nb, rows, -1)[..., :N],
                                 x.float().view(nb, rows, -1)[..., :K]).reshape(nb * N, K)
            assert rel_err(a, b) < 1e-5 and rel_err(a, exact) < 1e-3, (N, K, nb)


def test_grouped_colsum_and_copied_cast(dev):
    """bmhrl_colsum_bf16_groups == bmhrl_colsum_bf16 per group; bmhrl_cast_bf16_copies == bmhrl_cast_bf16 per copy"""
    from bmhrl_amd import ops
    torch.manual_seed(9)
    R, N, G = 480, 3072, 2
    dY = torch.randn(G * R, N, device=dev).bfloat16()
    db = torch.zeros(G * N + 8, device=dev)
    ops.colsum_bf16_groups(dY, N, db, R, N, G, N)
    ref = dY.float().view(G, R, N).sum(1).reshape(-1)
    assert rel_err(db[:G * N], ref) < 1e-5 and float(db[G * N:].abs().max()) == 0.0
    one = torch.zeros(N, device=dev)
    ops.colsum_bf16(dY, N, one, True, R, N, dy_off=R * N)
    assert rel_err(db[N:2 * N], one) < 1e-6
    x = torch.randn(200, 300, device=dev)
    ld = 304
    y = torch.zeros(2 * 200, ld, dtype=torch.bfloat16, device=dev)
    ops.cast_bf16_copies(x, 300, y, ld, 200, 300, 2, 200 * ld)
    y1 = torch.zeros(200, ld, dtype=torch.bfloat16, device=dev)
    ops.cast_bf16(x, 300, y1, ld, 200, 300)
    assert torch.equal(y[:200], y1) and torch.equal(y[200:], y1) and float(y[:, 300:].float().abs().max()) == 0.0


def test_batch_head_equals_shift_masks_and_counters(dev):
    """bmhrl_batch_head == captions[:, :-1] / captions[:, 1:] + make_masks of the shifted input (single and doubled) + the three
    device counters advanced by exactly one per launch"""
    from bmhrl_amd import ops, synthetic as syn
    from bmhrl_amd.model.masking import make_masks
    b = syn.synthetic_batch(5, 37, 90, 11, 60, seed=4, min_len=3)
    fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
    fs["audio"][1, 7, 0] = 0.0
    cap = b["captions"].to(dev)
    ref = make_masks(fs, cap[:, :-1], "audio_video", 1)
    seed = torch.tensor([41], dtype=torch.int64, device=dev)
    s1, s2 = torch.tensor([6], dtype=torch.int32, device=dev), torch.tensor([0], dtype=torch.int32, device=dev)
    for copies in (1, 2):
        vm, am, cm, trg_in, trg_y = ops.batch_head(fs["rgb"], fs["audio"], cap, 1, copies=copies, bump64=seed, bump32=[s1, s2])
        assert torch.equal(trg_in, cap[:, :-1]) and torch.equal(trg_y, cap[:, 1:]) and trg_in.is_contiguous() and trg_y.is_contiguous()
        for got, want in ((vm, ref["V_mask"]), (am, ref["A_mask"]), (cm, ref["C_mask"])):
            assert got.dtype == torch.bool and torch.equal(got, torch.cat([want] * copies))
    assert int(seed) == 43 and int(s1) == 8 and int(s2) == 2
    ops.batch_head(fs["rgb"], fs["audio"], cap, 1)                      # counters are optional
    assert int(seed) == 43
    with pytest.raises(RuntimeError):
        ops.batch_head(fs["rgb"], fs["audio"], cap.int(), 1)


def test_head_takes_dlogits_from_the_token_loss_node(dev):
    """TokenLossFn(sole_consumer=True) behind WorkerHeadFn: the head's backward uses the bf16 d logits the loss node's kernel
    wrote (log-softmax backward folded in) -- same gradients as the two-launch path, and the plain path when the log-probs
    did not come from a head or were not declared sole-consumer"""
    from bmhrl_amd import functional as F
    torch.manual_seed(5)
    B, L, d1, d2, V = 3, 7, 40, 24, 333
    x0, g0 = torch.randn(B, L, d1, device=dev), torch.randn(B, L, d2, device=dev)
    w0, b0 = torch.randn(V, d1 + d2, device=dev) * 0.1, torch.randn(V, device=dev) * 0.1
    trg = torch.randint(2, V, (B, L), device=dev)
    trg[0, 5:] = 1
    up = torch.tensor(2.5, device=dev)
    res = []
    for sole in (False, True):
        x, gc, w, b = (t.clone().requires_grad_(True) for t in (x0, g0, w0, b0))
        logp = F.WorkerHeadFn.apply(x, gc, w, b)
        loss = F.TokenLossFn.apply(logp, trg, None, None, None, 0.7, 1, 1.0, None, sole)
        loss.backward(gradient=up)
        assert not F._GRAD_TWIN
        res.append((loss.detach(), x.grad, gc.grad, w.grad, b.grad))
    assert torch.equal(res[0][0], res[1][0])
    for a, c in zip(res[0][1:], res[1][1:]):
        assert rel_err(c, a) < 4e-3            # (one bf16 rounding of d logits instead of two)
    # log-probs that no head produced: the flag changes nothing
    lp = torch.log_softmax(torch.randn(B, L, V, device=dev), -1)
    y = []
    for sole in (False, True):
        z = lp.clone().requires_grad_(True)
        F.TokenLossFn.apply(z, trg, None, None, None, 0.7, 1, 1.0, None, sole).backward()
        y.append(z.grad)
    assert torch.equal(y[0], y[1]) and not F._GRAD_TWIN


def test_sole_consumer_promise_is_checked_not_trusted(dev):
    """TokenLossFn(sole_consumer=True) hands bf16 d logits to the worker head behind autograd's back and returns a placeholder as
    the log-probs' gradient.  A broken promise -- a second consumer of the log-probs, a tensor hook that rewrites the gradient
    -- must raise in the head's backward; a retained gradient shows NaN, never uninitialised memory (r03 review)."""
    from bmhrl_amd import functional as F
    torch.manual_seed(6)
    B, L, d1, d2, V = 2, 5, 40, 24, 128
    x0, g0 = torch.randn(B, L, d1, device=dev), torch.randn(B, L, d2, device=dev)
    w0, b0 = torch.randn(V, d1 + d2, device=dev) * 0.1, torch.randn(V, device=dev) * 0.1
    trg = torch.randint(2, V, (B, L), device=dev)

    def head():
        x, gc, w, b = (t.clone().requires_grad_(True) for t in (x0, g0, w0, b0))
        return x, F.WorkerHeadFn.apply(x, gc, w, b)

    # a second consumer: autograd sums the two gradients, the placeholder does not arrive
    x, logp = head()
    loss = F.TokenLossFn.apply(logp, trg, None, None, None, 0.7, 1, 1.0, None, True) + 0.1 * logp.sum()
    with pytest.raises(RuntimeError, match="second consumer"):
        loss.backward()
    F._GRAD_TWIN.clear()
    # a hook that rewrites the gradient
    x, logp = head()
    logp.register_hook(lambda g: g * 2)
    with pytest.raises(RuntimeError, match="second consumer"):
        F.TokenLossFn.apply(logp, trg, None, None, None, 0.7, 1, 1.0, None, True).backward()
    F._GRAD_TWIN.clear()
    # a view of the log-probs is not the head's tensor: the plain path, correct gradients
    x, logp = head()
    F.TokenLossFn.apply(logp.view(B, L, V)[:, :, :], trg, None, None, None, 0.7, 1, 1.0, None, True).backward()
    gx_view = x.grad.clone()
    x, logp = head()
    F.TokenLossFn.apply(logp, trg, None, None, None, 0.7, 1, 1.0, None, False).backward()
    assert rel_err(gx_view, x.grad) < 1e-6 and not F._GRAD_TWIN
    # retain_grad: what is retained is the placeholder -- NaN, not garbage -- and the real gradients are unaffected
    x, logp = head()
    logp.retain_grad()
    F.TokenLossFn.apply(logp, trg, None, None, None, 0.7, 1, 1.0, None, True).backward()
    assert bool(torch.isnan(logp.grad).all()) and rel_err(x.grad, gx_view) < 4e-3 and not F._GRAD_TWIN


def test_token_loss_node_equals_sum_over_n_tokens(dev, golden):
    """functional.TokenLossFn (row sums -> one-block reduce -> scalar, gradient scaled inside the kernel) == the loops'
    torch.sum(criterion(pred, y)) / (n_tokens * factor) over SmoothKLFn, value and gradient; against the oracle too."""
    from bmhrl_amd.functional import SmoothKLFn, TokenLossFn
    from oracle import bmhrl_oracle as O
    g = golden("losses")
    logits, trg = torch.from_numpy(g["logits"]).to(dev), torch.from_numpy(g["trg"]).to(dev)
    sampled, score = torch.from_numpy(g["sampled"]).to(dev), torch.from_numpy(g["score"]).to(dev)
    n_row = (trg != 1).sum(-1, keepdim=True).float().expand_as(trg).contiguous()
    for bt, sc, nr, factor, w in ((None, None, None, 1.0, None), (sampled, score, n_row, 0.2, torch.tensor([1.25], device=dev))):
        x = logits.clone().requires_grad_(True)
        loss = TokenLossFn.apply(torch.log_softmax(x, -1), trg, bt, sc, nr, 0.7, 1, factor, w)
        (loss * 3.0).backward()
        y = logits.clone().requires_grad_(True)
        rows, _ = SmoothKLFn.apply(torch.log_softmax(y, -1), trg, bt, sc, nr, 0.7, 1)
        ref = rows.sum() / ((trg != 1).sum() * factor) * (1.0 if w is None else w[0])
        (ref * 3.0).backward()
        assert rel_err(loss, ref) < 1e-6 and rel_err(x.grad, y.grad) < 1e-5
    z = torch.from_numpy(g["logits"]).requires_grad_(True)
    ref = O.warmstart_loss(torch.log_softmax(z, -1), torch.from_numpy(g["trg"]), 0.7, 1)
    x = logits.clone().requires_grad_(True)
    loss = TokenLossFn.apply(torch.log_softmax(x, -1), trg, None, None, None, 0.7, 1, 1.0, None)
    assert rel_err(loss, ref.detach()) < 1e-5


def test_criteria_unreduced_form_equals_the_reference_outputs(dev, golden):
    """LabelSmoothing.unreduced / BiasedKL.unreduced return what the reference's forward returns -- the (B*S, V) divergence
    itself (loss/label_smoothing.py:32, loss/biased_kl.py:52) -- checked against the reference's own outputs (kat.npz: the
    SURVEY appendix A4 cases incl. the `idx.sum() > 0` guard; losses.npz: random case with padded rows)."""
    from bmhrl_amd.loss.biased_kl import BiasedKL
    from bmhrl_amd.loss.label_smoothing import LabelSmoothing
    T = lambda a: torch.from_numpy(a).to(dev)
    k = golden("kat")
    lp = T(k["a4_lp"])
    ls, bkl = LabelSmoothing(0.7, 1), BiasedKL(0.7, 1)
    assert rel_err(ls.unreduced(lp, torch.tensor([[2, 4, 1], [5, 3, 2]], device=dev)), T(k["a4_ls"])) < 1e-5
    assert rel_err(ls.unreduced(lp, torch.tensor([[1, 4, 2], [5, 3, 2]], device=dev)), T(k["a4_ls_guard"])) < 1e-5
    got = bkl.unreduced(lp, torch.tensor([[2, 4, 1], [5, 3, 2]], device=dev), torch.tensor([[2, 0, 3], [1, 3, 4]], device=dev),
                        torch.tensor([[.5, .25, 1.], [.8, 0., .1]], device=dev))
    assert rel_err(got, T(k["a4_bkl"])) < 1e-5
    g = golden("losses")
    lp = torch.log_softmax(T(g["logits"]), -1)
    assert rel_err(ls.unreduced(lp, T(g["trg"])), T(g["ls"])) < 1e-5
    assert rel_err(bkl.unreduced(lp, T(g["trg"]), T(g["sampled"]), T(g["bkl_raw_amp"])), T(g["bkl_raw"])) < 1e-5
    # and the row sums the training path uses are the sums of exactly these rows
    assert rel_err(ls(lp, T(g["trg"])).view(-1), T(g["ls"]).sum(-1)) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rows,V,pad_rows", [(480, 10172, (3, 7, 479)), (7, 300, (0, 2)), (5, 2048, ()), (16, 4096, (0,))])
def test_head_loss_one_launch_equals_the_four_kernels(rows, V, pad_rows):
    """ops.head_loss (log-softmax + label-smoothing rows + token-normalised loss + bf16 d logits, one launch) against
    log_softmax_ + smooth_kl_fwd + token_loss_reduce + smooth_kl_bwd(wrt_logits): log-probs, scale and the gradient bit for
    bit, the loss up to the order of a row's fp32 sum; twice in a row (the kernel re-arms its counter)"""
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + V)
    logits = (torch.randn(rows, V, generator=g) * 3).to(dev)
    trg = torch.randint(2, V, (rows,), generator=g).to(dev)
    pad = 1
    for r in pad_rows:
        trg[r] = pad
    weight = torch.tensor([0.7], device=dev)
    dl = torch.tensor(1.25, device=dev)
    ref = logits.clone()
    ops.log_softmax_(ref, V, rows, V)
    row_loss = torch.empty(rows, device=dev)
    ops.smooth_kl_fwd(ref, V, trg, None, None, None, 0.7, pad, -1, row_loss, None, rows, V)
    out = torch.empty(2, device=dev)
    ops.token_loss_reduce(row_loss, trg, rows, pad, weight, 1.0, out[0:1], out[1:2])
    ldg = (V + 7) // 8 * 8
    gb = torch.zeros(rows, ldg, dtype=torch.bfloat16, device=dev)
    ops.smooth_kl_bwd(ref, V, trg, None, None, None, 0.7, pad, -1, out[1:2], gb, ldg, None, rows, V, wrt_logits=True,
                      loss_scale2=dl.reshape(1))
    counter = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(2):
        x = logits.clone()
        rl2 = torch.empty(rows, device=dev)
        out2 = torch.full((2,), float("nan"), device=dev)
        gb2 = torch.zeros(rows, ldg, dtype=torch.bfloat16, device=dev)
        ops.head_loss(x, V, trg, 0.7, pad, weight, 1.0, dl, rl2, out2, gb2, ldg, counter, rows, V)
        torch.cuda.synchronize()
        assert int(counter.abs().sum()) == 0
        assert torch.equal(x, ref)
        assert torch.equal(out2[1], out[1])
        assert torch.equal(gb2, gb)
        assert float((rl2 - row_loss).abs().max()) <= 2e-6 * float(row_loss.abs().max())
        assert abs(float(out2[0]) - float(out[0])) <= 2e-6 * abs(float(out[0]))


@pytest.mark.parametrize("poison", ["inf", "-inf_all", "nan", "huge"])
def test_head_loss_reports_a_non_finite_row_like_the_unfused_tail(dev, poison):
    """A NaN / Inf logit (diverged training) must show in the loss of the one-launch tail exactly as in the four-kernel tail:
    the rows are summed in 2^-32 fixed point there, and a non-finite row has no fixed-point image (r03 ADVICE: it came out as a
    finite garbage number).  The step after it must be clean again (the flag word re-arms with the counter)."""
    from bmhrl_amd import ops
    rows, V, pad = 96, 1000, 1
    g = torch.Generator().manual_seed(7)
    logits = (torch.randn(rows, V, generator=g) * 3).to(dev)
    trg = torch.randint(2, V, (rows,), generator=g).to(dev)
    bad = logits.clone()
    if poison == "inf":
        bad[5, 17] = float("inf")
    elif poison == "-inf_all":
        bad[9, :] = float("-inf")
    elif poison == "nan":
        bad[40, 3] = float("nan")
    else:
        bad[3, int(trg[3])] = -3.0e38        # a finite logit whose row loss is beyond the fixed-point range
    weight = torch.tensor([1.0], device=dev)
    counter = torch.zeros(4, dtype=torch.int32, device=dev)

    def fused(x):
        rl = torch.empty(rows, device=dev)
        out = torch.full((2,), 123.0, device=dev)
        gb = torch.zeros(rows, V, dtype=torch.bfloat16, device=dev)
        ops.head_loss(x.clone(), V, trg, 0.7, pad, weight, 1.0, None, rl, out, gb, V, counter, rows, V)
        torch.cuda.synchronize()
        return out[0].item()

    def unfused(x):
        ref = x.clone()
        ops.log_softmax_(ref, V, rows, V)
        rl = torch.empty(rows, device=dev)
        ops.smooth_kl_fwd(ref, V, trg, None, None, None, 0.7, pad, -1, rl, None, rows, V)
        out = torch.empty(2, device=dev)
        ops.token_loss_reduce(rl, trg, rows, pad, weight, 1.0, out[0:1], out[1:2])
        torch.cuda.synchronize()
        return out[0].item()

    want, got = unfused(bad), fused(bad)
    if poison == "huge":        # a finite row beyond 2^20 saturates to +Inf (the float sum would be ~1e38: diverged either way)
        assert want > 1e30 and got == float("inf"), (want, got)
    else:
        assert not math.isfinite(want)
        assert (math.isnan(want) and math.isnan(got)) or want == got, (want, got)
    assert int(counter.abs().sum()) == 0
    clean = fused(logits)
    assert math.isfinite(clean) and abs(clean - unfused(logits)) <= 2e-6 * abs(clean)


@pytest.mark.parametrize("B,H,Sq,Sk,kind", [(2, 4, 800, 256, "pad"), (2, 4, 200, 250, "pad"), (1, 4, 33, 130, "allmasked"), (2, 4, 64, 20, "pad"),
                                            (3, 3, 100, 1, "none"), (2, 2, 129, 161, "none"), (16, 4, 800, 256, "pad")])
def test_attention_fwd_two_phase_form_for_short_memories(dev, B, H, Sq, Sk, kind):
    """bmhrl_attention_fwd in the exact-softmax two-phase form for at most 256 keys (csrc/attention_fwd_sk256.hip; taken by itself
    for grids that fill the chip, forced here with config code 256): context, statistics and the output-dropout mask against the
    float64 reference and against the generic online-softmax kernel on the same inputs."""
    from bmhrl_amd import _lib, ops
    dk = 256
    D = H * dk
    g = torch.Generator().manual_seed(Sq * 1000 + Sk)
    Q = bf(torch.randn(B, Sq, D, generator=g)).to(dev)
    K = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    V = bf(torch.randn(B, Sk, D, generator=g)).to(dev)
    mask, sb = None, 0
    if kind == "pad":
        mask = torch.ones(B, 1, Sk, dtype=torch.uint8)
        mask[0, 0, Sk - Sk // 3:] = 0
        mask[-1, 0, 5:9] = 0
        sb = Sk
    elif kind == "allmasked":
        mask = torch.zeros(B, 1, Sk, dtype=torch.uint8)
        sb = Sk
    if mask is not None:
        mask = mask.to(dev).contiguous()
    scale = 1 / math.sqrt(dk)
    out = {}
    try:
        for code in (256, 41):
            _lib.check(_lib.load().bmhrl_attention_config(256, code), "bmhrl_attention_config")
            for p_drop in (0.0, 0.25):
                O = torch.zeros(B, Sq, D, dtype=torch.bfloat16, device=dev)
                rmax = torch.empty(B, H, Sq, device=dev)
                rsum = torch.empty(B, H, Sq, device=dev)
                ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, sb, 0, B, H, Sq, Sk, dk, scale, D, D, D, D, dropout_p=p_drop, seed=77)
                torch.cuda.synchronize()
                out[code, p_drop] = (O.float(), rmax.double().cpu(), rsum.double().cpu())
    finally:
        _lib.load().bmhrl_attention_config(256, 0)
    ref, s = _attn_ref(Q, K, V, mask, H, scale)
    O, rmax, rsum = out[256, 0.0]
    assert rel_err(O, ref) < 1.5e-2
    lse_ref = torch.logsumexp(s, -1).cpu()
    assert float((rmax + torch.log(rsum) - lse_ref).abs().max()) < 2e-3 * max(1.0, float(lse_ref.abs().max()) * 1e-6 + 1.0)
    assert float((s.max(-1).values.cpu() - rmax).abs().max()) < 1e-3 * max(1.0, float(rmax.abs().max()) * 1e-6)     # the EXACT maximum
    if kind == "allmasked":
        assert float((rmax + 1e9).abs().max()) == 0.0 and float((rsum - Sk).abs().max()) < 1e-3 * Sk
        assert rel_err(O, V.float().view(B, Sk, D).mean(1, keepdim=True).expand(B, Sq, D)) < 1.5e-2
    assert rel_err(O, out[41, 0.0][0]) < 1.5e-2                       # the generic kernel on the same inputs
    # output dropout: the same elements are dropped for the same seed (same element numbering), kept ones scaled by 1 / (1 - p)
    Od, Og = out[256, 0.25][0], out[41, 0.25][0]
    big = O.abs() > 1e-2 * O.abs().max()
    assert torch.equal((Od == 0)[big], (Og == 0)[big])
    kept = big & (Od != 0)
    assert rel_err(Od[kept], (O / 0.75)[kept]) < 1.5e-2
