"""Vectorised (loop-free, sync-free) versions of the reference's host-side RL glue: the manager's
segment product / sum (epoch_loops/captioning_bmrl_loops.py:301-316), segment_reward (metrics/batched_meteor.py:19-36)
and discontinue_reward (metrics/util.py:53-88).  The reference walks `torch.nonzero(segments)` in Python -- one device
sync plus a handful of tiny kernels per segment; these run as a fixed number of tensor ops on whatever device the inputs
live on.  Quirks of the loops are reproduced and named in the docstrings; tests/test_rl_glue_cpu.py checks every
function against the loop restatements in oracle/ on random inputs."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

Tensor = torch.Tensor


def _segment_layout(segments: Tensor):
    """seg (B, L) marks segment ends.  Returns (sid, n_seg, in_seg): the segment number of every position (number of ends
    strictly before it), the number of segments per row, and whether a position lies inside a segment (not in the tail)."""
    seg = (segments != 0).long()
    csum = torch.cumsum(seg, dim=1)
    sid = csum - seg
    n_seg = csum[:, -1:]
    return sid, n_seg, sid < n_seg


def _tail_zeroed(n_seg: Tensor) -> Tensor:
    """(B,1) bool: rows whose tail the reference's loops zero when they move on to a later row: every row with segments
    that is followed by another row with segments, plus row 0 when it has none but some other row has (`old_b = 0`)."""
    has = (n_seg > 0).squeeze(1)
    later = torch.flip(torch.cummax(torch.flip(has.long(), [0]), 0).values, [0])       # any(has[b:])
    later_strict = torch.cat([later[1:], later.new_zeros(1)]) > 0
    z = has & later_strict
    z0 = (~has[0]) & (has.any())
    z = z.clone()
    z[0] = z[0] | z0
    return z.unsqueeze(1)


def _segment_reduce(values: Tensor, sid: Tensor, in_seg: Tensor, op: str) -> Tensor:
    """reduce `values` over each segment and broadcast the result back over the segment's positions"""
    B, L = values.shape
    idx = torch.where(in_seg, sid, torch.full_like(sid, L))            # tail positions go to a spare bucket
    init = 1.0 if op == "prod" else 0.0
    buckets = torch.full((B, L + 1), init, dtype=values.dtype, device=values.device)
    buckets = buckets.scatter_reduce(1, idx, values, reduce=op, include_self=True)
    return torch.gather(buckets, 1, idx)


def manager_segments(sampled_probs: Tensor, expected_scores: Tensor, segments: Tensor) -> Tuple[Tensor, Tensor]:
    """(segment_prob, expected_scores') of the manager branch of biased_kl: product of the sampled probabilities and sum
    of the expected scores per segment, on every position of the segment; zero / unchanged tails as the reference."""
    sid, n_seg, in_seg = _segment_layout(segments)
    zt = _tail_zeroed(n_seg)
    prob = torch.where(in_seg, _segment_reduce(sampled_probs.float(), sid, in_seg, "prod"), torch.zeros_like(sampled_probs, dtype=torch.float32))
    es = expected_scores
    summed = _segment_reduce(es, sid, in_seg, "sum")
    es_out = torch.where(in_seg, summed, torch.where(zt, torch.zeros_like(es), es))
    return prob, es_out


def segment_reward(reward: Tensor, sections: Tensor) -> Tuple[Tensor, Tensor]:
    """(segment_reward, segment_indices) of metrics/batched_meteor.py: per-segment reward sums on the segment positions,
    zeros behind the last segment end."""
    sid, n_seg, in_seg = _segment_layout(sections)
    out = torch.where(in_seg, _segment_reduce(reward.float(), sid, in_seg, "sum"), torch.zeros_like(reward, dtype=torch.float32))
    return out, torch.nonzero(sections)


def discontinue_reward(cider_diff: Tensor, gamma: float, n_step: int = 100, segments: Optional[Tensor] = None) -> Tensor:
    """metrics/util.py discontinue_reward.  Without segments: out[b, t] = sum_{i < n_step} gamma^i x[b, t+i] as one
    (L, L) banded matrix product.  With segments: see the loop restatement for the reference's guard on the first two
    segment ends of the batch, which is kept."""
    x = cider_diff.float()
    B, L = x.shape
    dev = x.device
    if segments is None:
        d = torch.arange(L, device=dev)[None, :] - torch.arange(L, device=dev)[:, None]          # column - row
        w = torch.where((d >= 0) & (d < n_step), torch.as_tensor(float(gamma), device=dev) ** d.clamp(min=0).float(),
                        torch.zeros((), device=dev))
        return x @ w.t()
    sid, n_seg, in_seg = _segment_layout(segments)
    seg = segments != 0
    # rank of every segment end in the row-major list of all ends of the batch
    flat_rank = (torch.cumsum(seg.reshape(-1).long(), 0) - 1).reshape(B, L)
    # value written over segment k of row b: the reward at its end; ends with global rank < 2 add the discounted rewards
    # of the later ends of the same row (exponent = distance in the list, i.e. difference of the within-row numbers)
    end_val = torch.where(seg, x, torch.zeros_like(x))                                     # rewards at the ends
    k = sid                                                                                # segment number at an end
    # per row: table T[b, j] = reward at the j-th end (j < n_seg)
    T = torch.zeros(B, L + 1, device=dev).scatter_add(1, torch.where(seg, k, torch.full_like(k, L)), end_val)[:, :L]
    j = torch.arange(L, device=dev)
    disc = float(gamma) ** (j[None, :] - j[:, None]).clamp(min=0).float()                  # gamma^(j2 - j1) for j2 >= j1
    upper = (j[None, :] >= j[:, None]).float()
    valid = (j[None, :] < n_seg).float()                                                   # (B, L): j < n_seg
    full = torch.einsum("bj,ij->bi", T * valid, disc * upper)                              # sum_{j2 >= j1} gamma^(j2-j1) T[b, j2]
    # the guard: only ends whose global rank is 0 or 1 accumulate; the others keep their own reward
    first_rank = torch.where(seg, flat_rank, torch.full_like(flat_rank, 1 << 30)).min(dim=1, keepdim=True).values   # rank of end 0 of the row
    rank_of_seg = first_rank + j[None, :]                                                  # global rank of the row's j-th end
    per_seg = torch.where(rank_of_seg < 2, full, T)                                        # (B, L): value of segment j
    spread = torch.gather(per_seg, 1, sid.clamp(max=L - 1))
    zt = _tail_zeroed(n_seg)
    return torch.where(in_seg, spread, torch.where(zt, torch.zeros_like(x), x))
