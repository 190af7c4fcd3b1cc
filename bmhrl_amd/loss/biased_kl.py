"""BiasedKL and Reinforce with the reference's signatures (loss/biased_kl.py:11-53, :61-81).

BiasedKL.forward(pred, trg, biased_trg, biased_offset) of the reference receives the amplitude as a tensor that is
still attached to `pred` (epoch_loops/captioning_bmrl_loops.py:285,321-322).  The fused kernel computes the
amplitude itself from (score, tokens per row), so the worker step calls `biased_kl_from_score`; the 4-argument
reference form is kept for an amplitude that is a plain (detached) tensor."""
import torch
import torch.nn as nn

from .. import ops
from ..functional import ManagerKLFn, SmoothKLFn


class BiasedKL(nn.Module):

    def __init__(self, label_smoothing, pad_idx):
        super().__init__()
        self.pad_idx = pad_idx
        self.ls = label_smoothing
        self.trg_factor = 1 - self.ls

    def biased_kl_from_score(self, pred, trg, biased_trg, score, n_row):
        """Row sums (B*S, 1) of the divergence with amp = clamp(score * p(a) * n_row, 0, 1) attached to pred;
        also returns the amplitude (B, S)."""
        rows, amp = SmoothKLFn.apply(pred, trg, biased_trg, score, n_row, float(self.ls), int(self.pad_idx))
        return rows.unsqueeze(-1), amp.view(trg.shape)

    def biased_kl_from_segments(self, pred, trg, biased_trg, score, n_seg, segments):
        """Manager form: amp = clamp(score * prod_{segment} p(a) * n_seg, 0, 1), attached to pred through EVERY token of the
        segment (reference epoch_loops/captioning_bmrl_loops.py:299-322).  Returns (row sums (B*S, 1), amplitude (B, S))."""
        rows, amp = ManagerKLFn.apply(pred, trg, biased_trg, score, n_seg, segments, float(self.ls), int(self.pad_idx))
        return rows.unsqueeze(-1), amp.view(trg.shape)

    def unreduced(self, pred, trg, biased_trg, biased_offset):
        """the (B*S, V) tensor the reference's forward returns (loss/biased_kl.py:52) for a GIVEN amplitude; no gradient."""
        B, S, V = pred.shape
        with torch.no_grad():
            p = torch.gather(torch.exp(pred), 2, biased_trg.unsqueeze(-1)).squeeze(-1)
            score = (biased_offset.detach().float() / p.clamp_min(1e-30)).contiguous().view(-1)
            out = torch.empty(B * S, V, device=pred.device)
            ops.smooth_kl_full(pred.detach().contiguous(), V, trg.contiguous().view(-1), biased_trg.contiguous().view(-1), score,
                               torch.ones_like(score), float(self.ls), int(self.pad_idx), -1, out, B * S, V)
        return out

    def forward(self, pred, trg, biased_trg, biased_offset, segments=None):
        # amp = clamp(offset * p(a) * n, 0, 1) with offset' = offset / p(a), n = 1 reproduces a given amplitude;
        # the gradient then treats the amplitude as constant only if it saturates -- use biased_kl_from_score
        # for the attached form.
        B, S, V = pred.shape
        with torch.no_grad():
            p = torch.gather(torch.exp(pred), 2, biased_trg.unsqueeze(-1)).squeeze(-1)
            score = biased_offset.detach().float() / p.clamp_min(1e-30)
        return self.biased_kl_from_score(pred, trg, biased_trg, score, torch.ones_like(score))[0]


class Reinforce(nn.Module):
    """-mean(adv.detach * log clamp(p(a))) + mean(adv^2) (reference :61-81); `pred` are probabilities."""

    def __init__(self):
        super().__init__()
        self.pad_idx = 0
        self.eps = 1e-5

    def forward(self, pred, action, value, critic_value):
        from ..functional import ReinforceFn
        return ReinforceFn.apply(pred, action.reshape(pred.shape[0], pred.shape[1]), value, critic_value)
