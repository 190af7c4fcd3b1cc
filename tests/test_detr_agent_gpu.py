"""DetrCaption (the reference's DETR-mode agent, model/det_bmhrl_agent.py) on the HIP kernels:
state-dict layout == the reference module's, the Conv1d('same') + GroupNorm input projection, the 100-query object detector and
the video encoder against the reference's own outputs (tests/golden/detr_agent.npz), the log-probs (whose causal decoder the CPU
reference cannot run) against the oracle, and the conv / group-norm gradients against torch autograd."""
import numpy as np
import pytest
import torch

from bmhrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def rel(a, b, floor=1e-6):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def _agent(dev, g):
    from types import SimpleNamespace
    from bmhrl_amd.model.det_bmhrl_agent import DetrCaption
    cfg = syn.tiny_cfg(d_model=64, d_model_video=64, d_vid=64, d_model_caps=20, rl_att_heads=4, rl_goal_d=8, dout_p=0.0)
    cfg.pre_goal_attention = False
    cfg.device = str(dev)
    agent = DetrCaption(cfg, SimpleNamespace(trg_voc_size=41, train_vocab=SimpleNamespace(vectors=None)))
    keys = [str(k) for k in g["keys"]]
    shapes = {k: tuple(int(d) for d in str(s).split(",") if d != "") for k, s in zip(keys, g["shapes"])}
    mine = {k: tuple(v.shape) for k, v in agent.state_dict().items()}
    assert mine == shapes                                   # the reference module's checkpoint layout, key for key
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=13)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(20, seed=1).items()})
    # worker_decoder.norm and manager_decoder.norm are ONE LayerNorm module in the reference (model/det_bmhrl_agent.py:52-53:
    # `worker_norm = manager_norm = ...`) and here: a real checkpoint holds the same values under both keys
    for leaf in ("weight", "bias"):
        sd[f"worker_decoder.norm.{leaf}"] = sd[f"manager_decoder.norm.{leaf}"]
    agent.load_state_dict(sd)
    assert agent.worker_decoder.norm is agent.manager_decoder.norm
    return agent.to(dev).eval(), cfg, sd


def test_detr_agent_against_the_reference_and_the_oracle(dev, golden):
    from oracle import bmhrl_oracle as O
    g = golden("detr_agent")
    agent, cfg, sd = _agent(dev, g)
    x, mask = torch.from_numpy(g["x_video"]).to(dev), torch.from_numpy(g["V_mask"]).to(dev)
    with torch.no_grad():
        vf = x
        for i, proj in enumerate(agent.input_proj):
            from bmhrl_amd.functional import Conv1dSameFn, GroupNormFn
            vf = GroupNormFn.apply(Conv1dSameFn.apply(vf, proj[0].weight, proj[0].bias), proj[1].weight, proj[1].bias, 32, proj[1].eps)
            assert rel(vf, torch.from_numpy(g[f"proj{i}"])) < 1e-2, i          # bf16 operands of the convolution GEMM
        cls, hs, ob_mask = agent.object_detector(vf, mask)
        assert rel(hs, torch.from_numpy(g["obj_hs"])) < 3e-2 and rel(cls, torch.from_numpy(g["obj_logits"])) < 3e-2
        assert float((ob_mask.cpu() != torch.from_numpy(g["obj_mask"])).float().mean()) < 0.02
        mem = agent.encoder(vf, mask, agent.pos_enc)
        assert rel(mem, torch.from_numpy(g["memory"])) < 3e-2
        # the whole forward against the oracle (pinned to the fixture above by tests/test_oracle_golden.py)
        gen = torch.Generator().manual_seed(3)
        B, L = x.shape[0], 7
        trg = torch.randint(4, 41, (B, L), generator=gen)
        trg[0, 5] = 3
        trg[0, 6:] = 1
        c_mask = ((trg != 1).unsqueeze(1) & torch.ones(L, L, dtype=torch.bool).tril().unsqueeze(0))
        masks_cpu = {"V_mask": torch.from_numpy(g["V_mask"]), "C_mask": c_mask}
        ref = O.detr_caption_forward(sd, cfg, torch.from_numpy(g["x_video"]), trg, masks_cpu)
        out = agent((x, None), trg.to(dev), {k: v.to(dev) for k, v in masks_cpu.items()})
        assert len(out) == 6 and out[3] is None and out[4] is None
        # (bf16 operands through 3 convolutions + 18 post-norm layers at head width 16)
        assert rel(out[0], ref[0]) < 3e-2 and rel(out[1], ref[1]) < 3e-2 and rel(out[2], ref[2]) < 3e-2 and rel(out[5], ref[3]) < 3e-2
        assert torch.allclose(out[0].exp().sum(-1).cpu(), torch.ones(B, L), atol=1e-4)
        pred, _, _ = agent.inference((x, None), trg.to(dev), {k: v.to(dev) for k, v in masks_cpu.items()}, None, None)
        assert torch.equal(pred, out[0])


@pytest.mark.parametrize("k", [3, 6, 9])
def test_conv1d_same_and_groupnorm_gradients(dev, k):
    """functional.Conv1dSameFn / GroupNormFn against torch's conv1d(padding='same') / group_norm on bf16-rounded operands"""
    from bmhrl_amd.functional import Conv1dSameFn, GroupNormFn
    g = torch.Generator().manual_seed(k)
    B, T, C, Co = 3, 17, 64, 64
    bf = lambda t: t.to(torch.bfloat16).float()
    x0, w0, b0 = bf(torch.randn(B, T, C, generator=g)), bf(torch.randn(Co, C, k, generator=g) * 0.1), torch.randn(Co, generator=g) * 0.1
    gam, bet = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    up = torch.randn(B, T, Co, generator=g)
    xr, wr, br, gr, ber = (t.clone().requires_grad_(True) for t in (x0, w0, b0, gam, bet))
    yr = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, padding="same")
    zr = torch.nn.functional.group_norm(yr, 32, gr, ber, 1e-5).transpose(1, 2)
    (zr * up).sum().backward()
    xd, wd, bd, gd, bed = (t.clone().to(dev).requires_grad_(True) for t in (x0, w0, b0, gam, bet))
    yd = Conv1dSameFn.apply(xd, wd, bd)
    assert rel(yd, yr.transpose(1, 2).detach()) < 2e-5
    zd = GroupNormFn.apply(yd, gd, bed, 32, 1e-5)
    assert rel(zd, zr.detach()) < 2e-5
    (zd * up.to(dev)).sum().backward()
    assert rel(gd.grad, gr.grad) < 1e-4 and rel(bed.grad, ber.grad) < 1e-4
    for a, b in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert rel(a, b) < 1.5e-2           # the incoming gradient is rounded to bf16 for the two GEMMs
