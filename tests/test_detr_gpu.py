"""Post-norm encoder / decoder layers (reference model/encoder.py, model/decoder.py) on the HIP kernels against the
fixture produced by the reference and, for the causal branch the CPU reference cannot run, against the oracle."""
import pytest
import torch

from bmhrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    return float((a.detach().cpu().float() - b).abs().max() / b.abs().max())


def _setup(golden):
    from bmhrl_amd.model.blocks import PositionalEncoder
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = golden("detr")
    enc, dec, d = syn.detr_tiny_modules()
    pe = {k: PositionalEncoder(n, 0.0) for k, n in (("D", d["D"]), ("C", d["dC"]), ("G", d["dG"]))}
    t = {k: T(g[k]).to(dev) for k in ("src", "mask", "tgt", "qpos", "qmask", "objs", "goal")}
    return dev, g, enc.to(dev).eval(), dec.to(dev).eval(), d, pe, t


def test_encoder_decoder_match_reference_fixture(golden):
    dev, g, enc, dec, d, pe, t = _setup(golden)
    mem = enc(t["src"], t["mask"], pe["D"])
    assert rel(mem, T(g["enc_out"])) < 1e-2
    mem = T(g["enc_out"])[-1].to(dev)
    a = dec(t["tgt"], mem, t["mask"], pe["D"], t["qpos"], t["qmask"], None, None, None, True, t["objs"], None)
    assert rel(a, T(g["dec_a"])) < 1e-2
    b = dec(t["tgt"], mem, t["mask"], pe["D"], pe["C"], None, t["goal"], t["qmask"], pe["G"], False, None, None)
    assert rel(b, T(g["dec_b"])) < 1e-2


def test_causal_decoder_and_gradients_vs_oracle(golden):
    """add_pos=False with a query mask: the causal fill (GPU-only in the reference) + goal attention; forward and
    the gradients of inputs and a few weights against autograd through the oracle."""
    from oracle import bmhrl_oracle as O
    dev, g, enc, dec, d, pe, t = _setup(golden)
    H = d["H"]
    cpu = {k: v.cpu() for k, v in t.items()}
    dsd = {"dec." + k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and v.numel() > 0)
           for k, v in dec.state_dict().items()}
    mem_c = T(g["enc_out"])[-1].clone().requires_grad_(True)
    tgt_c = cpu["tgt"].clone().requires_grad_(True)
    pad = torch.ones(3, 1, 6, dtype=torch.bool)            # a padding-only query mask: the causal part must come from `causal`
    pad[1, 0, 4:] = False
    ref = O.detr_stack(dsd, "dec", 2, tgt_c, lambda p, x: O.detr_decoder_layer(
        dsd, p, x, mem_c, cpu["mask"], None, pad, cpu["goal"], cpu["qmask"], False, None, H), True)
    w = torch.linspace(-1, 1, ref.numel()).view_as(ref)
    (ref * w).sum().backward()

    mem = mem_c.detach().to(dev).requires_grad_(True)
    tgt = tgt_c.detach().to(dev).requires_grad_(True)
    out = dec(tgt, mem, t["mask"], pe["D"], pe["C"], pad.to(dev), t["goal"], t["qmask"], pe["G"], False, None, None)
    assert rel(out, ref.detach()) < 1e-2
    (out * w.to(dev)).sum().backward()

    def rel_l2(a, b):
        return float((a.detach().cpu() - b).norm() / b.norm())
    assert rel_l2(tgt.grad, tgt_c.grad) < 3e-2
    assert rel_l2(mem.grad, mem_c.grad) < 3e-2
    params = dict(dec.named_parameters())
    for name in ("layers.0.self_attn.linear_V2d.weight", "layers.0.multihead_attn.linear_K2d.weight",
                 "layers.1.goal_attention.linear_Q2d.weight", "layers.0.linear1.weight", "layers.1.norm3.weight",
                 "layers.0.norm1.bias", "norm.weight"):
        # linear1: at width 24 a few pre-activations round across zero in bf16 and flip whole ReLU rows (same effect
        # and bound as tests/test_agent_gpu.py's tiny-model gradient check)
        tol = 1e-1 if name.endswith("linear1.weight") else 3e-2
        assert rel_l2(params[name].grad, dsd["dec." + name].grad) < tol, name
    assert params["layers.0.detected_attention.linear_Q2d.weight"].grad is None     # branch not taken -> no gradient


def test_encoder_layer_full_width_flash_path():
    """d_model=1024, H=4, S=256: the general (q=k != v) entry runs the flash kernel; forward and d(src) vs the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd.model.blocks import PositionalEncoder
    from bmhrl_amd.model.encoder import TransformerEncoderLayer
    from oracle import bmhrl_oracle as O
    dev = torch.device("cuda:0")
    layer = TransformerEncoderLayer(1024, 4, 1024, 0.0)
    sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}, seed=4)
    layer.load_state_dict(sd)
    layer = layer.to(dev).eval()
    g = torch.Generator().manual_seed(8)
    src_c = torch.randn(2, 256, 1024, generator=g).requires_grad_(True)
    mask = torch.ones(2, 1, 256, dtype=torch.bool)
    mask[1, 0, 200:] = False
    osd = {"l." + k: v for k, v in sd.items()}
    ref = O.detr_encoder_layer(osd, "l", src_c, mask, 4)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    src = src_c.detach().to(dev).requires_grad_(True)
    out = layer(src, mask.to(dev), PositionalEncoder(1024, 0.0))
    assert rel(out, ref.detach()) < 1e-2
    (out * w.to(dev)).sum().backward()
    assert float((src.grad.cpu() - src_c.grad).norm() / src_c.grad.norm()) < 2e-2
