"""One-launch few-query attention over a memory (csrc/memory_attention.hip, the core of functional.PairMemAttnFn) against plain
fp32 torch and against the GEMM + row-kernel path it replaces."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,L,Sk,dm,masked", [(16, 4, 30, 256, 1024, True), (16, 4, 30, 800, 128, True), (2, 2, 7, 37, 64, True),
                                                 (3, 1, 32, 896, 32, False), (2, 3, 30, 20, 96, True)])
def test_memory_attention_kernel_against_torch(B, H, L, Sk, dm, masked):
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Sk + dm)
    B2, Skp = 2 * B, (Sk + 7) & ~7
    assert ops.memory_attention_ok(L, Sk, dm) and not ops.memory_attention_ok(33, Sk, dm) and not ops.memory_attention_ok(L, 900, dm)
    mem = torch.randn(B, Sk, dm, generator=g).to(dev)
    y = torch.empty(B * Sk, dm, dtype=torch.bfloat16, device=dev)
    ldt = (Sk + 15) & ~15
    yt = torch.full((B, dm, ldt), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.cast_memory(mem, y, yt, B, Sk, dm, ldt)
    assert torch.equal(y.view(B, Sk, dm), mem.bfloat16()) and torch.equal(yt[:, :, :Sk], mem.bfloat16().transpose(1, 2))
    assert float(yt[:, :, Sk:].float().abs().sum()) == 0.0
    scale = 1.0 / math.sqrt(dm / 4)
    ldq, ldp = 2 * H * dm, 2 * H * Skp
    QD = torch.zeros(B2 * L, ldq, dtype=torch.bfloat16, device=dev)
    Q = (torch.randn(B2, L, H, dm, generator=g) * 0.4).to(dev).bfloat16()
    dC = (torch.randn(B2, L, H, dm, generator=g) * 0.4).to(dev).bfloat16()
    QD.view(B2, L, 2, H, dm)[:, :, 1] = Q
    QD.view(B2, L, 2, H, dm)[:, :, 0] = dC
    mask = None
    if masked:
        mask = torch.rand(B2, 1, Sk, generator=g) > 0.25
        mask[:, :, 0] = True
        mask[1] = False                                  # a fully masked sample
        mask = mask.to(dev)
    PD = torch.zeros(B2 * L, ldp, dtype=torch.bfloat16, device=dev)
    Cx = torch.zeros(B2 * L, H * dm, dtype=torch.bfloat16, device=dev)
    ops.memory_attention(False, QD, H * dm, ldq, y, yt, ldt, PD, ldp, H * Skp, Cx, H * dm, mask, Sk, B, B2, H, L, Sk, dm, scale)
    memf = mem.bfloat16().float().repeat(2, 1, 1)                                   # sample b2 reads memory b2 % B
    s = torch.einsum("blhd,bkd->blhk", Q.float(), memf) * scale
    if mask is not None:
        s = s.masked_fill(~mask.view(B2, 1, 1, Sk), -1e9)
    p_ref = torch.softmax(s, -1)
    P = PD.view(B2, L, 2, H, Skp)[:, :, 0, :, :Sk]
    assert rel_err(P, p_ref) < 1e-2
    assert float(PD.view(B2, L, 2, H, Skp)[:, :, 0, :, Sk:].float().abs().sum()) == 0.0
    c_ref = torch.einsum("blhk,bkd->blhd", P.float(), memf)
    assert rel_err(Cx.view(B2, L, H, dm), c_ref) < 1e-2
    # backward: dS from the same (rounded) P, dQ' = dS mem
    dQ = torch.zeros(B2 * L, H * dm, dtype=torch.bfloat16, device=dev)
    ops.memory_attention(True, QD, 0, ldq, y, yt, ldt, PD, ldp, H * Skp, dQ, H * dm, mask, Sk, B, B2, H, L, Sk, dm, scale)
    dP = torch.einsum("blhd,bkd->blhk", dC.float(), memf)
    Pf = P.float()
    dS_ref = scale * Pf * (dP - (Pf * dP).sum(-1, keepdim=True))
    if mask is not None:
        dS_ref = dS_ref.masked_fill(~mask.view(B2, 1, 1, Sk), 0.0)
    dS = PD.view(B2, L, 2, H, Skp)[:, :, 1, :, :Sk]
    assert rel_err(dS, dS_ref) < 1.5e-2
    dq_ref = torch.einsum("blhk,bkd->blhd", dS.float(), memf)
    assert rel_err(dQ.view(B2, L, H, dm), dq_ref) < 1e-2
    if mask is not None:
        assert float(dS[1].float().abs().max()) == 0.0 and rel_err(P[1], torch.full_like(P[1].float(), 1.0 / Sk)) < 1e-2


def test_pair_memory_attention_fused_equals_gemm_path():
    """functional.PairMemAttnFn with the one-launch core == the same block on score GEMM + softmax + context GEMM (and
    dP GEMM + row kernel + dQ' GEMM): output and every gradient"""
    from bmhrl_amd import functional as F
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, L, dq, D, H = 4, 30, 300, 1024, 4
    for Sk, dm in ((256, 1024), (800, 128)):
        x0 = torch.randn(2, B, L, dq, device=dev)
        mem0 = torch.randn(B, Sk, dm, device=dev)
        mask = torch.ones(2 * B, 1, Sk, dtype=torch.bool, device=dev)
        mask[1, :, Sk // 2:] = False
        mask[5, :, 10:] = False
        def params():
            g = torch.Generator(device="cpu").manual_seed(7)
            mk = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(dev).requires_grad_(True)
            one = lambda: [torch.ones(dq, device=dev).requires_grad_(True), torch.zeros(dq, device=dev).requires_grad_(True),
                           mk(D, dq), mk(D), mk(D, dm), mk(D), mk(D, dm), mk(D), mk(dq, D), mk(dq)]
            return one() + one()
        res = []
        old = F.FUSED_MEMATTN
        try:
            for fused in (False, True):
                F.FUSED_MEMATTN = fused
                F.SHADOWS.invalidate()
                ps = params()
                x, mem = x0.clone().requires_grad_(True), mem0.clone().requires_grad_(True)
                y = F.PairMemAttnFn.apply(x, mem, mask, H, 0.0, *ps)
                (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
                res.append([y.detach(), x.grad, mem.grad] + [p.grad for p in ps])
        finally:
            F.FUSED_MEMATTN = old
        for i, (a, b) in enumerate(zip(*res)):
            if a is None or b is None:
                assert a is None and b is None
                continue
            assert rel_err(b, a) < 2e-2, (Sk, dm, i, rel_err(b, a))
