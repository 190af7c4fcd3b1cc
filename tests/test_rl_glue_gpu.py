"""The vectorised RL glue on the device (no host loop, no nonzero() sync in the maths) against the oracle loops."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_glue_on_device_matches_loops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import rl_glue as G
    from oracle import bmhrl_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, L = 16, 30
    seg = (torch.rand(B, L, generator=g) < 0.2).int()
    seg[0] = 0
    p = torch.rand(B, L, generator=g) * 0.9 + 0.05
    es = torch.randn(B, L, generator=g)
    sp0, es0 = O.manager_segment_loop(p, es, seg)
    sp1, es1 = G.manager_segments(p.to(dev), es.to(dev), seg.to(dev))
    assert torch.allclose(sp1.cpu(), sp0, rtol=1e-5, atol=1e-7) and torch.allclose(es1.cpu(), es0, rtol=1e-5, atol=1e-6)
    r = torch.randn(B, L, generator=g)
    assert torch.allclose(G.segment_reward(r.to(dev), seg.to(dev))[0].cpu(), O.segment_reward_loop(r, seg)[0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(G.discontinue_reward(r.to(dev), 0.9, 5).cpu(), O.discontinue_reward_loop(r, 0.9, 5), rtol=1e-5, atol=1e-5)
    assert torch.allclose(G.discontinue_reward(r.to(dev), 0.9, 100, seg.to(dev)).cpu(), O.discontinue_reward_loop(r, 0.9, 100, seg),
                          rtol=1e-5, atol=1e-5)
