"""IEEE-half operand builds of the two attention forward kernels (BASELINE configs[4]: "fp16/bf16 MFMA cross-attention"):
same kernels, v_mfma_f32_32x32x16_f16, P and the output rounded to fp16.  Against fp64 torch attention on the fp16-rounded
inputs; tolerance 3e-3 of the output's maximum (fp16 P and output: 2^-11 per rounding; the bf16 build is held to 1.5e-2)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qh, kh, vh, mask, scale):
    s = (qh.double() @ kh.double().transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, -1e9)
    return torch.softmax(s, -1) @ vh.double(), s


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 4, 70, 100), (2, 4, 256, 800), (2, 4, 800, 256), (8, 4, 1024, 2048), (1, 4, 33, 130)])
def test_attention_fwd_f16(B, H, Sq, Sk):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    dk, D = 256, H * 256
    g = torch.Generator().manual_seed(Sq + 7 * Sk)
    Q, K, V = (torch.randn(B, S, D, generator=g).to(dev).to(torch.float16) for S in (Sq, Sk, Sk))
    mask = torch.ones(B, 1, Sk, dtype=torch.uint8, device=dev)
    mask[0, 0, Sk - Sk // 3:] = 0
    if Sq == 33:
        mask[:] = 0                                   # fully masked: uniform over all keys
    K[0, Sk - 1] *= 6                                 # late running-max jump: the rescale branch
    O = torch.zeros(B, Sq, D, dtype=torch.float16, device=dev)
    rmax, rsum = torch.empty(B, H, Sq, device=dev), torch.empty(B, H, Sq, device=dev)
    scale = 1 / math.sqrt(dk)
    ops.attention_fwd(Q, K, V, O, rmax, rsum, mask, Sk, 0, B, H, Sq, Sk, dk, scale, D, D, D, D)
    hv = lambda t, S: t.view(B, S, H, dk).transpose(1, 2)
    ref, s = _ref(hv(Q, Sq), hv(K, Sk), hv(V, Sk), mask.bool().view(B, 1, 1, Sk), scale)
    ref = ref.transpose(1, 2).reshape(B, Sq, D)
    err = float((O.double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-3, err
    lse = rmax.double() + torch.log(rsum.double())
    ok = torch.logsumexp(s, -1) > -1e8
    if ok.any():
        assert float((lse[ok] - torch.logsumexp(s, -1)[ok]).abs().max()) < 2e-3
    with pytest.raises(RuntimeError):                 # mixed operand types are refused
        ops.attention_fwd(Q, K.to(torch.bfloat16), V, O, rmax, rsum, mask, Sk, 0, B, H, Sq, Sk, dk, scale, D, D, D, D)


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 4, 200, 130), (16, 4, 256, 800), (8, 4, 1024, 2048), (8, 4, 70, 200)])
def test_shared128_attention_f16(B, H, Sq, Sk):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + Sq + Sk)
    Qp = (0.5 * torch.randn(B, Sq, H, 128, generator=g)).to(dev).to(torch.float16)
    X = torch.randn(B, Sk, 128, generator=g).to(dev).to(torch.float16)
    mask = torch.ones(B, Sk, dtype=torch.bool, device=dev)
    mask[0, Sk - 5:] = False
    mask[B - 1, :] = False
    scale = 1.0 / 16
    ctx = torch.empty(B, Sq, H, 128, dtype=torch.float16, device=dev)
    rmax, rsum = torch.empty(B, H, Sq, device=dev), torch.empty(B, H, Sq, device=dev)
    ops.attention_shared128_fwd(Qp, X, ctx, rmax, rsum, mask, Sk, B, H, Sq, Sk, scale, H * 128, 128, H * 128)
    s = torch.einsum("bqhd,bkd->bhqk", Qp.double(), X.double()) * scale
    s = s.masked_fill(~mask[:, None, None, :], -1e9)
    ref = torch.einsum("bhqk,bkd->bqhd", torch.softmax(s, -1), X.double())
    err = float((ctx.double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-3, err
    # the bf16 build on the same (fp16-representable) values is one order less accurate: the variants really differ
    ctx_b = torch.empty(B, Sq, H, 128, dtype=torch.bfloat16, device=dev)
    ops.attention_shared128_fwd(Qp.to(torch.bfloat16), X.to(torch.bfloat16), ctx_b, rmax, rsum, mask, Sk, B, H, Sq, Sk, scale,
                                H * 128, 128, H * 128)
    ref_b = torch.einsum("bhqk,bkd->bqhd", torch.softmax(
        (torch.einsum("bqhd,bkd->bhqk", Qp.to(torch.bfloat16).double(), X.to(torch.bfloat16).double()) * scale)
        .masked_fill(~mask[:, None, None, :], -1e9), -1), X.to(torch.bfloat16).double())
    err_b = float((ctx_b.double() - ref_b).abs().max() / ref_b.abs().max())
    assert err < err_b < 2e-2, (err, err_b)


def test_f16_and_bf16_builds_run_at_the_same_speed():
    """config-5 shape (B=8, Tv=1024, Ta=2048): the two operand types share the kernel, so their launch times agree"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    B, H, Sq, Sk, dk = 8, 4, 2048, 1024, 256
    D = H * dk
    times = {}
    for dt in (torch.bfloat16, torch.float16):
        Q, K, V = (torch.randn(B, S, D, device=dev).to(dt) for S in (Sq, Sk, Sk))
        O = torch.empty(B, Sq, D, dtype=dt, device=dev)
        rmax, rsum = torch.empty(B, H, Sq, device=dev), torch.empty(B, H, Sq, device=dev)
        run = lambda: ops.attention_fwd(Q, K, V, O, rmax, rsum, None, 0, 0, B, H, Sq, Sk, dk, 1 / 16, D, D, D, D)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        times[dt] = e0.elapsed_time(e1) / 20 * 1e3
    gf = 4.0 * B * H * Sq * Sk * dk / 1e9
    print("A<-V at config 5 (Sq 2048, Sk 1024): " + ", ".join(f"{str(k).split('.')[-1]} {v:.1f} us = {gf / v * 1e3 / 2500 * 100:.1f} % of peak"
                                                              for k, v in times.items()))
    a, b = times.values()
    assert abs(a - b) < 0.15 * max(a, b)
