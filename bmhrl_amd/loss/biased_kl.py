"""BiasedKL and Reinforce with the reference's signatures (loss/biased_kl.py:11-53, :61-81).

BiasedKL.forward(pred, trg, biased_trg, biased_offset) takes the amplitude as a tensor ARGUMENT, exactly as the reference
does: if the caller computed it from `pred` (epoch_loops/captioning_bmrl_loops.py:285,321-322 leave it attached) the
gradient flows through it, if the caller detached it, it does not -- functional.GivenAmpKLFn differentiates w.r.t. both.
The trainer's own steps use the fused forms (`biased_kl_from_score`, `biased_kl_from_segments`), whose kernels form the
amplitude clamp(score * p(a) * n, 0, 1) themselves: same values and gradients as the attached 4-argument call, fewer passes.
forward() returns the ROW SUMS (B*S, 1) of the divergence (the callers only ever sum it); `unreduced()` gives the (B*S, V)
tensor itself."""
import torch
import torch.nn as nn

from .. import ops
from ..functional import GivenAmpKLFn, ManagerKLFn, SmoothKLFn


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
        """reference :22-53 (`segments` is ignored there too); amplitudes in [0, 1]"""
        rows = GivenAmpKLFn.apply(pred, trg, biased_trg, biased_offset, float(self.ls), int(self.pad_idx))
        return rows.unsqueeze(-1)


class Reinforce(nn.Module):
    """-mean(adv.detach * log clamp(p(a))) + mean(adv^2) (reference :61-81); `pred` are probabilities."""

    def __init__(self):
        super().__init__()
        self.pad_idx = 0
        self.eps = 1e-5

    def forward(self, pred, action, value, critic_value):
        from ..functional import ReinforceFn
        return ReinforceFn.apply(pred, action.reshape(pred.shape[0], pred.shape[1]), value, critic_value)
