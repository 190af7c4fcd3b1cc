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


@pytest.mark.parametrize("stabilize", [False, True])
def test_manager_branch_of_biased_kl_matches_the_oracle(stabilize):
    """biased_kl(train_worker=False): arg-max tokens, score * segments, per-segment product / sum with the reference's
    row-transition quirks, amplitude attached to the prediction through EVERY probability of the segment product
    (reference epoch_loops/captioning_bmrl_loops.py:283-334) -- value and gradient against the oracle's loop restatement."""
    import torch
    from bmhrl_amd.epoch_loops.captioning_bmrl_loops import biased_kl
    from bmhrl_amd.loss.biased_kl import BiasedKL
    from oracle import bmhrl_oracle as O
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, L, V, pad = 5, 7, 23, 1
    logits = torch.randn(B, L, V, generator=g) * 2.0
    trg = torch.randint(2, V, (B, L), generator=g)
    trg[0, 5:] = pad
    trg[3, 3:] = pad
    # segment ends: row 0 none (zeroed by the `old_b = 0` quirk), rows 1 and 3 two segments, row 2 none, row 4 one (last row keeps its tail)
    seg = torch.zeros(B, L, dtype=torch.int32)
    seg[1, 1] = seg[1, 4] = seg[3, 0] = seg[3, 5] = seg[4, 2] = 1
    score = torch.rand(B, L, generator=g) * 4.0
    score[3, 0] = 50.0                                      # this segment's amplitude reaches the clamp (no gradient through it)
    baseline = torch.rand(B, L, generator=g) * 0.3
    mask = trg != pad

    ref_lp = torch.log_softmax(logits, -1).requires_grad_(True)
    div, ref_score, ref_tok, ref_amp = O.manager_biased_kl(ref_lp, trg, score, baseline, mask, seg, 0.7, pad, stabilize)
    ref_rows = div.sum(-1)
    w = torch.linspace(0.5, 1.5, B * L)                    # distinct upstream gradients per row
    (ref_rows * w).sum().backward()

    lp = torch.log_softmax(logits, -1).to(dev).requires_grad_(True)
    rows, sc, tok, amp = biased_kl(False, lp, None, baseline.to(dev), trg.to(dev), None, mask.to(dev), seg.to(dev), dev,
                                   BiasedKL(0.7, pad), stabilize, reward_fn=lambda a, c: score.to(dev))
    (rows.view(-1) * w.to(dev)).sum().backward()
    assert torch.equal(tok[0].cpu(), ref_tok)
    assert torch.allclose(sc[0].cpu(), ref_score, atol=1e-6)
    assert torch.allclose(amp[0].cpu(), ref_amp.detach(), atol=1e-5, rtol=1e-4)
    assert 0 < int(((ref_amp > 0) & (ref_amp < 1)).sum()) and int((ref_amp >= 1).sum()) > 0   # both clamp regimes are exercised
    assert torch.allclose(rows.view(-1).detach().cpu(), ref_rows.detach(), atol=1e-4, rtol=1e-4)
    gerr = (lp.grad.cpu() - ref_lp.grad).abs().max() / ref_lp.grad.abs().max()
    assert float(gerr) < 1e-4, float(gerr)


def test_biased_kl_against_the_reference_fixture(golden):
    """tests/golden/rl_loops.npz: outputs of the reference's own biased_kl (epoch_loops/captioning_bmrl_loops.py:271-334).
    Manager branch end to end (arg-max tokens, score, divergence rows, gradient w.r.t. the logits); worker branch through
    BiasedKL.biased_kl_from_score on the tokens the reference drew (its sample comes from torch's global generator)."""
    from bmhrl_amd.epoch_loops.captioning_bmrl_loops import biased_kl
    from bmhrl_amd.loss.biased_kl import BiasedKL
    dev = torch.device("cuda:0")
    g = golden("rl_loops")
    T = torch.from_numpy
    crit = BiasedKL(0.7, 1)
    for i in range(int(g["n"])):
        logits, trg, seg = T(g[f"logits{i}"]).to(dev), T(g[f"trg{i}"]).to(dev), T(g[f"seg{i}"]).to(dev)
        score, base, stab = T(g[f"score{i}"]).to(dev), T(g[f"base{i}"]).to(dev), bool(g[f"stab{i}"])
        mask = trg != 1
        x = logits.clone().requires_grad_(True)
        rows, sc, tok, _ = biased_kl(False, torch.log_softmax(x, -1), None, base.clone(), trg, None, mask, seg, dev, crit, stab,
                                     reward_fn=lambda a, c: score)
        rows.sum().backward()
        assert torch.equal(tok[0].cpu(), T(g[f"m_tok{i}"])), i
        assert torch.allclose(sc[0].cpu(), T(g[f"m_score{i}"]), atol=1e-6), i
        ref_rows = T(g[f"m_div{i}"]).sum(-1)
        assert torch.allclose(rows.view(-1).detach().cpu(), ref_rows, atol=2e-5, rtol=1e-4), i
        rg = T(g[f"m_grad{i}"])
        assert float((x.grad.cpu() - rg).abs().max()) <= 1e-4 * max(float(rg.abs().max()), 1e-3), i
        # worker branch, given tokens
        x = logits.clone().requires_grad_(True)
        w_tok = T(g[f"w_tok{i}"]).to(dev)
        s = (score - base) * mask.float() if stab else score
        n_row = mask.sum(-1, keepdim=True).float().expand_as(trg)
        rows, _ = crit.biased_kl_from_score(torch.log_softmax(x, -1), trg, w_tok, s, n_row)
        rows.sum().backward()
        assert torch.allclose(rows.view(-1).detach().cpu(), T(g[f"w_div{i}"]).sum(-1), atol=2e-5, rtol=1e-4), i
        rg = T(g[f"w_grad{i}"])
        assert float((x.grad.cpu() - rg).abs().max()) <= 1e-4 * max(float(rg.abs().max()), 1e-3), i


@pytest.mark.parametrize("tag", ["attached", "detached", "other"])
def test_biased_kl_forward_takes_the_amplitude_as_an_argument(golden, tag):
    """BiasedKL.forward(pred, trg, biased_trg, biased_offset) against the reference's own outputs
    (tests/golden/biased_kl_forward.npz from loss/biased_kl.py:22-53): the amplitude attached to the prediction the way the
    loops leave it, the same amplitude detached, and an amplitude that is another function of the prediction -- the
    gradient w.r.t. the logits differs in all three, and the node must not decide which one the caller meant."""
    from bmhrl_amd.loss.biased_kl import BiasedKL
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    g = golden("biased_kl_forward")
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    logits, trg, sampled, score, up = T("logits"), T("trg"), T("sampled"), T("score"), T("up")
    x = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(x, -1)
    p = torch.gather(torch.exp(lp), 2, sampled.unsqueeze(-1)).squeeze(-1)
    n = (trg != 1).sum(-1).reshape(-1, 1).float()
    amp = torch.clamp(score * torch.sqrt(p) * 0.9 + 0.05, 0, 1) if tag == "other" else torch.clamp(score * p * n, 0, 1)
    if tag == "detached":
        amp = amp.detach()
    assert torch.allclose(amp.detach(), T(f"{tag}_amp"), atol=1e-6)
    rows = BiasedKL(0.7, 1)(lp, trg, sampled, amp)
    assert rows.shape == (trg.numel(), 1)
    (rows.squeeze(-1) * up).sum().backward()
    ref_rows, ref_grad = T(f"{tag}_rows"), T(f"{tag}_grad_logits")
    assert torch.allclose(rows.view(-1).detach(), ref_rows, atol=2e-5, rtol=1e-4)
    assert float((x.grad - ref_grad).abs().max()) <= 2e-4 * float(ref_grad.abs().max())
    # the three gradients really are different things (the fixture is not degenerate)
    for t in ("attached", "detached", "other"):
        if t != tag:
            assert float((T(f"{t}_grad_logits") - ref_grad).abs().max()) > 2e-2 * float(ref_grad.abs().max())
