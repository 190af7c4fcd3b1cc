"""The HIP segment critic (fp32 MFMA input projections + per-step recurrence kernels) against the fixture produced by the
reference's SegmentCritic and against the CPU oracle at the real width (d=300, H=600, B=16, L=30)."""
import numpy as np
import pytest
import torch

from bmhrl_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _critic(cfg, dev):
    from bmhrl_amd.model.bm_hrl_agent import SegmentCritic
    c = SegmentCritic(cfg)
    c.load_state_dict(syn.synthetic_critic_state(cfg.d_model_caps, seed=1))
    return c.to(dev)


@pytest.mark.parametrize("wavefront", [True, False])
def test_critic_matches_reference_fixture(golden, wavefront):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = golden("critic")
    cfg = syn.tiny_cfg()
    c = _critic(cfg, dev)
    c.wavefront = wavefront
    emb = torch.from_numpy(g["emb"]).to(dev)
    score, labels = c.score_and_labels(emb, 0.25)
    ref = torch.from_numpy(g["out"])
    assert float((score.cpu() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(labels.cpu(), (torch.sigmoid(ref) > 0.25).squeeze(-1).int())
    assert torch.equal(c(emb).cpu(), score.cpu())        # module call == reference forward signature


@pytest.mark.parametrize("wavefront", [True, False])
def test_critic_full_width_vs_oracle(wavefront):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import bmhrl_oracle as O
    dev = torch.device("cuda:0")
    cfg = syn.default_cfg()
    c = _critic(cfg, dev)
    c.wavefront = wavefront
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(16, 30, 300, generator=g) * 17.3      # embeddings are scaled by sqrt(300) in the agent
    sd = {"critic." + k: v for k, v in syn.synthetic_critic_state(300, seed=1).items()}
    ref = O.segment_critic(sd, "critic", emb)
    score, labels = c.score_and_labels(emb.to(dev), 0.25)
    assert float((score.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    margin = (ref.squeeze(-1) - float(np.log(0.25 / 0.75))).abs()
    ok = margin > 1e-4                                      # labels must agree wherever the score is not on the threshold
    assert torch.equal(labels.cpu()[ok], O.segment_labels(ref, 0.25)[ok])


@pytest.mark.parametrize("chunk", [1, 2, 3, 5, 16])
@pytest.mark.parametrize("B,L", [(16, 30), (3, 7), (33, 2), (1, 1)])
def test_wavefront_equals_layer_by_layer(B, L, chunk):
    """the (layer, time) wavefront with the fused input projection against the GEMM + step form (summation order differs by
    fp32 rounding only); batch > 16 (two batch blocks), single step, single row"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    c = _critic(syn.default_cfg(), dev)
    emb = (torch.randn(B, L, 300, generator=torch.Generator().manual_seed(B * 100 + L)) * 17.3).to(dev)
    c.wavefront = True
    c.wave_chunk = chunk                                 # layers trail each other by `chunk` steps; W_ih once per chunk
    s1, l1 = c.score_and_labels(emb, 0.25)
    c.wavefront = False
    s0, l0 = c.score_and_labels(emb, 0.25)
    assert float((s1 - s0).abs().max()) <= 1e-5 * max(1.0, float(s0.abs().max()))
    far = (s0.squeeze(-1) - float(np.log(0.25 / 0.75))).abs() > 1e-4       # labels agree wherever the score is off the threshold
    assert torch.equal(l1[far], l0[far])
