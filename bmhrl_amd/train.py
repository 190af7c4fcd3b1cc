"""The per-batch training step of the hot path (epoch_loops/captioning_bmrl_loops.py:1148-1160 warmstart,
:837-877 worker RL step) over the HIP-backed agent, with

  * one flat fp32 bucket for the trainable parameters and one for their gradients (frozen / never-reached
    parameters excluded: critic, BMFusionLayer.feed_forward, manager_core -- SURVEY.md section 7),
  * RCCL all-reduce of the flat gradient bucket for data parallel runs (one process per GPU),
  * one fused Adam launch per bucket (torch.optim.Adam semantics, the reference's default betas / eps),
  * optional whole-step HIP graph capture (static shapes; dropout seeds advance through a device word).
"""
from __future__ import annotations

from types import SimpleNamespace
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops, synthetic as syn
from .functional import SCRATCH, SEEDS, SHADOWS, ScratchState, TokenLossFn, cancel_head_loss, request_head_loss
from .loss.biased_kl import BiasedKL
from .loss.label_smoothing import LabelSmoothing
from .model.bm_hrl_agent import BMHrlAgent, BMManagerValueFunction, BMWorkerValueFunction
from .model.masking import make_masks


class FlatAdam:
    """Adam over parameters re-homed into one flat fp32 buffer (views keep the nn.Parameter objects and the
    state-dict layout intact).  `params` that never receive a gradient keep a zero slot."""

    def __init__(self, params: List[torch.nn.Parameter], lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.params0 = list(self.params)          # construction order (in_param_order)
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets = []
        n = 0
        for s in self.sizes:
            self.offsets.append(n)
            n += (s + 3) & ~3          # keep every slice 16-byte aligned
        self.n = n
        self.flat = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        for p, o, s in zip(self.params, self.offsets, self.sizes):
            self.flat[o:o + s].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + s].view(p.shape)
        self.grad_views = [self.grad[o:o + s].view(p.shape) for p, o, s in zip(self.params, self.offsets, self.sizes)]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        # shadows (bf16 copies of weights, functional.ShadowCache) of these parameters: `generation` moves with every update
        # -- also with every replay of a captured step, which no host code sees -- and `maintained` names the shadow entries
        # the Adam pass writes itself; every other entry that involves one of these parameters is stale after an update
        self.generation = 0
        self.maintained = set()
        me = weakref.ref(self)
        for p in self.params:
            p._bmhrl_owner = me
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)   # advanced on the device (graph-safe)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    direct_grads = os.environ.get("BMHRL_DIRECT_GRADS", "1") == "1"   # one rank: Adam reads the gradients in place (no gather)

    fused_shadows = os.environ.get("BMHRL_FUSED_SHADOWS", "1") == "1"   # False: plain Adam kernel + a whole-cache shadow refresh at the start of the next step

    def shadow_generation(self, kind, key):
        return -1 if (kind, key) in self.maintained else self.generation

    def mark_uncovered_stale(self):
        """shadow groups this optimizer changes parameters of but cannot write (see _segment_plan) must be re-cast at the
        start of every step -- also inside a captured step, where the host-side bookkeeping of step() does not run"""
        cached = self.__dict__.get("_seg_plan")
        if cached is not None:
            for kind, key in cached[1][3]:
                SHADOWS.mark_stale(kind, key)

    def _segment_plan(self):
        """(table, n_segments, n_blocks, uncovered shadow entries) for ops.adam_segments, or None while it cannot be built
        (during a graph capture before a warm-up step made it).  One row per parameter in bucket order; a parameter that
        belongs to a live shadow group all of whose members this optimizer owns gets the address of its rows in the group's
        bf16 buffer (or of its slice of the concatenated bias)."""
        index = self.__dict__.setdefault("_index", {id(p): i for i, p in enumerate(self.params)})
        dev = self.flat.device
        groups = []
        for kind, store in ((1, SHADOWS.w), (0, SHADOWS.b)):
            for key, e in store.items():
                params = [r() for r in e[2]]
                if any(p is None for p in params) or e[1].device != dev:
                    continue
                mine = [id(p) in index and self.params[index[id(p)]] is p for p in params]
                if any(mine):
                    groups.append((kind, key, e[1], params, mine))
        sig = tuple((kind, key, buf.data_ptr()) for kind, key, buf, _, _ in groups)
        gptrs = self.__dict__.get("_direct_ptrs")
        cached = self.__dict__.get("_seg_plan")
        if gptrs is not None and self.__dict__.get("_grad_dirty", True) and not torch.cuda.is_current_stream_capturing() \
                and not self.__dict__.get("_homes", False):       # (homes: the bucket holds this step's gradients, zeroed at its start)
            # the direct path reads the bucket only for parameters without a gradient: they must find zeros there, also when
            # a cached plan is reused after a step that went through the gather path (which leaves gradients in the bucket)
            self.grad.zero_()
            self._grad_dirty = False
        if cached is not None and cached[0] == sig and cached[2] == gptrs:
            return cached[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        dst, uncovered = {}, []
        for kind, key, buf, params, mine in groups:
            # members that are not this optimizer's (frozen parameters: the manager side in the worker phase shares the
            # paired fusion stacks' groups with the worker side) keep the rows they have; a parameter of mine that already
            # has a destination in another group cannot be written twice: that group is re-cast by ShadowCache.refresh()
            if any(m and id(p) in dst for p, m in zip(params, mine)):
                uncovered.append((kind, key))
                continue
            off = 0
            for p, m in zip(params, mine):
                if kind:
                    n, kk, ld = p.shape[0], p.shape[1], buf.shape[1]
                    if m:
                        dst[id(p)] = (buf.data_ptr() + 2 * off * ld, n, kk, ld | (SHADOWS.split.get(key, 0) << 32))
                    off += n
                else:
                    if m:
                        dst[id(p)] = (buf.data_ptr() + 4 * off, 1, p.numel(), 0)
                    off += p.numel()
        rows, blk = [], 0
        for i, (p, o, sz) in enumerate(zip(self.params, self.offsets, self.sizes)):
            d = dst.get(id(p), (0, 1, sz, 0))
            rows.append([o, d[0], d[1], d[2], d[3], blk, gptrs[i] if gptrs is not None else 0])
            blk += (sz + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
        # one table per bucket as well (first-block column rebased): the update of a bucket can be launched as soon as its
        # gradients are complete (CaptionTrainer, phased Adam), while backward goes on below it
        parts = []
        bounds = self.__dict__.get("plan_bounds") or self.__dict__.get("bucket_bounds")   # (plan_bounds: a finer partition, see CaptionTrainer)
        if bounds is not None:
            for lo, hi in zip(bounds, bounds[1:]):
                sub = [r[:5] + [r[5] - rows[lo][5]] + r[6:] for r in rows[lo:hi]]
                nblk = (rows[hi][5] if hi < len(rows) else blk) - (rows[lo][5] if lo < len(rows) else blk)
                parts.append((torch.tensor(sub, dtype=torch.int64).to(dev) if sub else None, len(sub), nblk))
        self._part_plans = parts
        plan = (torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), blk, uncovered)
        self._seg_plan = (sig, plan, gptrs)
        self.maintained = {(kind, key) for kind, key, _, _, _ in groups if (kind, key) not in set(uncovered)}
        return plan

    def set_buckets(self, counts: List[int]):
        """consecutive runs of `counts[i]` parameters form bucket i, in the order backward completes their gradients"""
        assert sum(counts) == len(self.params) and all(c >= 0 for c in counts)
        self.bucket_bounds = [0]
        for c in counts:
            self.bucket_bounds.append(self.bucket_bounds[-1] + c)
        self.n_early = counts[0]
        self.split_off = self._elem_off(self.bucket_bounds[1])       # first element of bucket 1

    def set_split(self, n_early: int):
        """two buckets: the first n_early parameters (gradients complete early in backward) and the rest"""
        self.set_buckets([n_early, len(self.params) - n_early])

    def _elem_off(self, param_index: int) -> int:
        return self.offsets[param_index] if param_index < len(self.params) else self.n

    def _bucket_range(self, part):
        if part is None:
            return 0, len(self.params)
        return self.bucket_bounds[part], self.bucket_bounds[part + 1]

    def gather_grads(self, part=None):
        """copy the autograd gradients into the flat bucket; missing ones stay zero.  On the GPU this is ONE launch of the
        segmented copy kernel (bmhrl_cast_segments, fp32 mode): inside a trainer step the gradients live at fixed
        addresses (slices of the step scratch arena), so the segment table is built once and reused.
        part: None = all parameters, i = bucket i of set_buckets() / set_split()."""
        one_rank = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if part is not None and one_rank and self.direct_grads and self.fused_shadows and self.flat.is_cuda and \
                self.__dict__.get("phased_direct", False):
            # one rank, phased Adam: the bucket's gradients stay where autograd left them.  Before the plan exists (first
            # warm-up pass) there is nothing to do -- the full gather_grads() at the end of that pass records the addresses;
            # afterwards the addresses of this bucket must be the recorded ones (they are slices of the trainer's scratch arena)
            cached = self.__dict__.get("_seg_plan")
            if cached is not None and cached[2] is not None:
                lo, hi = self._bucket_range(part)
                ptrs = tuple(0 if p.grad is None else p.grad.data_ptr() for p in self.params[lo:hi])
                if ptrs != tuple(cached[2][lo:hi]):
                    raise RuntimeError("phased Adam: a gradient moved since the plan was built (bucket %d)" % part)
            return
        self._direct_ptrs = None
        if part is None and self.direct_grads and self.fused_shadows and self.flat.is_cuda and one_rank:
            # one process: nothing needs the gradients side by side -- the Adam pass reads each one where autograd left it
            ptrs = tuple(0 if p.grad is None else
                         (p.grad.data_ptr() if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else -1)
                         for p in self.params)
            if -1 not in ptrs and (not torch.cuda.is_current_stream_capturing() or
                                   self.__dict__.get("_seg_plan", (None, None, None))[2:] == (ptrs,)):
                self._direct_ptrs = ptrs
                return
        dst, src, missing = [], [], []
        lo, hi = self._bucket_range(part)
        for p, gv in zip(self.params[lo:hi], self.grad_views[lo:hi]):
            if p.grad is None:
                missing.append(gv)
            elif p.grad.data_ptr() != gv.data_ptr():          # (equal: produced in place, adopt_homes)
                dst.append(gv)
                src.append(p.grad)
        self._grad_dirty = True
        if missing:
            torch._foreach_zero_(missing)
        if not dst:
            return
        if not self.flat.is_cuda or any((not g.is_contiguous()) or g.dtype != torch.float32 for g in src):
            torch._foreach_copy_(dst, src)
            return
        sig = tuple((g.data_ptr(), d.data_ptr(), g.numel()) for g, d in zip(src, dst))
        plans = self.__dict__.setdefault("_gather_plans", {})
        if plans.get(part, (None,))[0] != sig:
            if torch.cuda.is_current_stream_capturing():      # (cannot upload a table now; capture() warms this up first)
                torch._foreach_copy_(dst, src)
                return
            rows, blk = [], 0
            for sp, dp, n in sig:
                rows.append([sp, dp, 1, n, 0, blk])
                blk += (n + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
            plans[part] = (sig, torch.tensor(rows, dtype=torch.int64).to(self.flat.device), len(rows), blk)
        _, table, n_seg, n_blk = plans[part]
        ops.cast_segments(table, n_seg, n_blk)

    def in_param_order(self, t: torch.Tensor) -> torch.Tensor:
        """a flat tensor of this optimiser (flat / grad / exp_avg ...) as the concatenation of its parameters' slices in
        CONSTRUCTION order: independent of the bucket layout (adopt_homes moves parameters inside their buckets)"""
        where = {id(p): (o, n) for p, o, n in zip(self.params, self.offsets, self.sizes)}
        return torch.cat([t[where[id(p)][0]:where[id(p)][0] + where[id(p)][1]] for p in self.params0])

    def _relayout(self, order: List[int]):
        """Re-home the parameters (and the Adam state) in the order `order`, a permutation of parameter indices that keeps
        every parameter inside its bucket"""
        old = {id(p): (o, n) for p, o, n in zip(self.params, self.offsets, self.sizes)}
        params = [self.params[i] for i in order]
        offsets, n = [], 0
        for i in order:
            offsets.append(n)
            n += (self.sizes[i] + 3) & ~3
        dev = self.flat.device
        new = {name: torch.zeros(n, device=dev) for name in ("flat", "exp_avg", "exp_avg_sq")}
        for name, buf in new.items():
            src = getattr(self, name)
            torch._foreach_copy_([buf[o:o + p.numel()] for p, o in zip(params, offsets)],
                                 [src[old[id(p)][0]:old[id(p)][0] + old[id(p)][1]] for p in params])
        self.params, self.offsets, self.sizes, self.n = params, offsets, [p.numel() for p in params], n
        self.flat, self.exp_avg, self.exp_avg_sq = new["flat"], new["exp_avg"], new["exp_avg_sq"]
        self.grad = torch.zeros(n, device=dev)
        for p, o, sz in zip(self.params, self.offsets, self.sizes):
            p.data = self.flat[o:o + sz].view(p.shape)
        self.grad_views = [self.grad[o:o + sz].view(p.shape) for p, o, sz in zip(self.params, self.offsets, self.sizes)]
        for k in ("_index", "_seg_plan", "_gather_plans", "_direct_ptrs", "_part_plans", "plan_bounds"):
            self.__dict__.pop(k, None)
        if self.__dict__.get("bucket_bounds") is not None:
            self.split_off = self._elem_off(self.bucket_bounds[1])
        SHADOWS.invalidate()

    def adopt_homes(self, state: ScratchState) -> int:
        """More than one rank: from now on the step's kernels write the leaf gradients straight into the flat bucket, which is
        what the all-reduce wants side by side -- gather_grads() then copies only what could not be placed.
        `state`: the trainer's ScratchState after a pass run with state.record set (p.grad still alive).  An allocation of that
        pass that the gradients of some parameters tile exactly (one weight; or the stacked [q; k; v] x two-stacks block of a
        paired layer, six parameters) becomes a contiguous run of the bucket: the parameters are moved next to each other
        inside their bucket, in the allocation's order (_relayout), and StepScratch.f32 hands out that run of self.grad instead
        of the arena slice.  A parameter with two consumers keeps working (autograd keeps the first producer's buffer and
        adds into it).  Returns the number of gradient elements that now skip the copy."""
        import bisect
        log = sorted(state.log, key=lambda r: r[3])
        starts = [r[3] for r in log]
        members = {}                                     # index into log -> [(byte offset in the allocation, param index)]
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.numel() != p.numel():
                continue
            j = bisect.bisect_right(starts, g.data_ptr()) - 1
            if j >= 0 and g.data_ptr() + 4 * g.numel() <= log[j][3] + 4 * log[j][2]:
                members.setdefault(j, []).append((g.data_ptr() - log[j][3], i))
        bucket_of = {}
        bounds = self.__dict__.get("bucket_bounds") or [0, len(self.params)]
        for b, (lo, hi) in enumerate(zip(bounds, bounds[1:])):
            for i in range(lo, hi):
                bucket_of[i] = b
        runs = []                                        # (log row, [param indices in allocation order])
        for j, mem in members.items():
            mem.sort()
            off, ok = 0, True
            for o, i in mem:
                ok = ok and o == off
                off += 4 * self.sizes[i]
            ok = ok and off == 4 * log[j][2] and len({bucket_of[i] for _, i in mem}) == 1
            ok = ok and (len(mem) == 1 or all(self.sizes[i] % 4 == 0 for _, i in mem))
            if ok:
                runs.append((log[j], [i for _, i in mem]))
        runs.sort(key=lambda r: (r[0][0], r[0][1]))      # allocation order
        in_run = {i: k for k, (_, idx) in enumerate(runs) for i in idx}
        order, done = [], set()
        for i in range(len(self.params)):                # bucket by bucket (the buckets are index ranges): runs first-come
            if i in done:
                continue
            group = runs[in_run[i]][1] if i in in_run else [i]
            order += group
            done.update(group)
        # (a run's members all sit in one bucket and `order` only moves a parameter to the position of the first member of
        # its run: the multiset of parameters per bucket range is unchanged only if runs do not straddle -- checked above)
        per_bucket = [sorted(order[lo:hi]) == list(range(lo, hi)) for lo, hi in zip(bounds, bounds[1:])]
        if not all(per_bucket):
            return 0
        self._relayout(order)
        pos_of_old = {old_i: k for k, old_i in enumerate(order)}     # (the members of a run are consecutive now)
        placed = 0
        for row, idx in runs:
            o0, n, off = self.offsets[pos_of_old[idx[0]]], row[2], 0
            ok = True
            for i in idx:
                ok = ok and self.offsets[pos_of_old[i]] == o0 + off
                off += self.sizes[pos_of_old[i]]
            if ok and off == n:
                state.homes[(row[0], row[1])] = self.grad[o0:o0 + n]
                placed += n
        if placed:
            state.home_buckets.append(self.grad)
            self._homes = True
        return placed

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return 1.0 / dist.get_world_size(group)
        return 1.0

    def all_reduce_part(self, part: Optional[int], group=None):
        """asynchronous all-reduce (sum) of one bucket (None: the whole flat gradient); returns the work handle (None without
        a group)"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            plo, phi = self._bucket_range(part)
            lo, hi = self._elem_off(plo), self._elem_off(phi)
            if hi > lo:
                return dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
        return None

    def step_part(self, part: int, grad_scale: float = 1.0, dev_step_advanced: bool = False, first: Optional[bool] = None,
                  last: Optional[bool] = None):
        """the update of bucket `part` alone (its gradients are complete; later buckets are still in backward).  Needs the
        plan of a previous full step(); the step counters advance with the first part of a step (default: part 0), shadows
        nobody maintains go stale with the last (default: the highest part)."""
        b1, b2 = self.betas
        if (part == 0) if first is None else first:
            self.step_count += 1
            self.generation += 1
            if not dev_step_advanced:
                self.step_dev.add_(1)
        table, n_seg, n_blk = self._part_plans[part]
        if n_seg:
            ops.adam_segments(table, n_seg, n_blk, self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, b1, b2, self.eps,
                              self.weight_decay, self.step_count, grad_scale, step_dev=self.step_dev)
        if (part == len(self._part_plans) - 1) if last is None else last:
            for kind, key in self._seg_plan[1][3]:
                SHADOWS.mark_stale(kind, key)

    def step(self, grad_scale: float = 1.0, dev_step_advanced: bool = False):
        """dev_step_advanced: the device step counter was already advanced for this step (ops.batch_head does it in the
        step's first launch)"""
        self.step_count += 1
        self.generation += 1
        b1, b2 = self.betas
        if self.flat.is_cuda:
            if not dev_step_advanced:
                self.step_dev.add_(1)
            plan = self._segment_plan() if self.fused_shadows else None
            if plan is None:
                ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.n, self.lr, b1, b2, self.eps,
                              self.weight_decay, self.step_count, grad_scale, step_dev=self.step_dev)
            else:
                # the update also writes the bf16 shadows (and concatenated-bias copies) of the weights it owns: they stay
                # current without the per-step refresh pass; shadows it could not place (a parameter in two groups) go stale
                table, n_seg, n_blk, uncovered = plan
                ops.adam_segments(table, n_seg, n_blk, self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, b1, b2,
                                  self.eps, self.weight_decay, self.step_count, grad_scale, step_dev=self.step_dev)
                for kind, key in uncovered:
                    SHADOWS.mark_stale(kind, key)
                return
        else:  # host-side reference used by the gloo tests of the data-parallel plumbing (no model arithmetic)
            g = self.grad * grad_scale
            if self.weight_decay:
                g = g + self.weight_decay * self.flat
            self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** self.step_count
            bc2 = 1 - b2 ** self.step_count
            self.flat.addcdiv_(self.exp_avg, (self.exp_avg_sq.sqrt() / bc2 ** 0.5).add_(self.eps), value=-self.lr / bc1)
        SHADOWS.invalidate()   # bf16 weight shadows are stale now


def token_weight(n_local: torch.Tensor, group=None) -> torch.Tensor:
    """Weight of this rank's loss in a data-parallel step: n_local * world / n_global.

    The reference normalises by the token count of the WHOLE batch (nn.DataParallel gathers the outputs of all devices and
    the loop divides the summed loss by the gathered `n_tokens`: scripts/train_rl_captioning_module.py:95-99,
    utilities/config_constructor.py:94, epoch_loops/captioning_bmrl_loops.py:846-847,1156-1158).  A rank here divides by
    its own count n_r and the gradients are averaged (sum / world); with the loss of rank r multiplied by
    n_r * world / sum(n) the average equals sum_r(rows_r) / sum_r(n_r) -- the reference's gradient.  One scalar all-reduce."""
    n = n_local.detach().to(torch.float32).reshape(1).clone()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return torch.ones((), device=n.device)
    total = n.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return (n * dist.get_world_size(group) / total.clamp_min(1.0)).reshape(())


def trainable_bucket(agent: BMHrlAgent) -> List[torch.nn.Parameter]:
    """Parameters the captioning loss can reach (the reference's optimiser holds all of them, but these are the only
    ones that ever receive a gradient): excludes the frozen critic, the never-applied fusion feed_forward and the
    unused manager_core (model/bm_hrl_agent.py:66,198-199,438)."""
    out = []
    for name, p in agent.named_parameters():
        if name.startswith("critic.") or ".feed_forward.fc" in name and "_fus." in name or name.startswith("manager_core.") \
                or name.startswith("manager.core."):
            continue
        if p.requires_grad:
            out.append(p)
    return out


class CaptionTrainer:
    """Owns agent + optimiser state for the captioning half of the reference's step on one GPU (one rank)."""

    def __init__(self, cfg, voc_size: int, device, lr: float = 1e-4, weight_decay: float = 0.0, seed: int = 0,
                 critic_state: Optional[Dict[str, torch.Tensor]] = None, pad_idx: int = 1, smoothing: float = 0.7,
                 phase: str = "warmstart", reward_fn=None, value_lr: float = 1e-4, exploration: Optional[bool] = None):
        """phase: "warmstart" (label-smoothing KL, reference :1132-1189), "worker" or "manager" (the RL step of
        train_bimodal_bl, reference :797-890: biased KL with sampled / arg-max tokens and their rewards + the value head's
        masked-MSE update; the phase decides which modules are trainable, teach_worker / teach_manager :572-589).
        reward_fn(sampled (B, L), captions) -> (B, L) rewards (BASELINE configs[2]: synthetic).
        exploration: None = what the reference's phase leaves it at -- ON in the warmstart and manager phases (the constructor
        sets it, teach_warmstart does not touch it: model/bm_hrl_agent.py:444-452,572-575), off in the worker phase;
        False switches the manager's Gaussian goal vector off (comparisons with the CPU oracle, which cannot draw it)."""
        assert phase in ("warmstart", "worker", "manager")
        self.phase = phase
        self.reward_fn = reward_fn
        cfg.device = str(device)
        ds = SimpleNamespace(trg_voc_size=voc_size, train_vocab=SimpleNamespace(vectors=None))
        self.cfg = cfg
        self.device = torch.device(device)
        self.agent = BMHrlAgent(cfg, ds)
        shapes = {k: tuple(v.shape) for k, v in self.agent.state_dict().items()}
        sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=seed, clone_layers=True)
        crit = critic_state if critic_state is not None else syn.synthetic_critic_state(cfg.d_model_caps, seed=1)
        sd.update({"critic." + k: v for k, v in crit.items()})
        self.agent.load_state_dict(sd)
        self.agent.to(self.device)
        if phase == "worker":
            self.agent.teach_worker()           # bm_enc + worker fusion + worker trainable, manager side frozen, no exploration
        elif phase == "manager":
            self.agent.teach_manager()          # manager fusion + manager trainable (exploration noise on, as the reference)
        else:
            self.agent.teach_warmstart()        # every module trainable; exploration stays as constructed (on), reference :572-575
        if exploration is not None:
            self.agent.manager.exploration = bool(exploration)
        self.value_net = None
        if phase != "warmstart":
            vcls = BMWorkerValueFunction if phase == "worker" else BMManagerValueFunction
            self.value_net = vcls(cfg)
            vshapes = {k: tuple(v.shape) for k, v in self.value_net.state_dict().items()}
            self.value_net.load_state_dict(syn.fill_state_dict(vshapes, seed=seed + 7))
            self.value_net.to(self.device)
            self.vopt = FlatAdam(list(self.value_net.parameters()), lr=value_lr)
        self.stabilize = bool(getattr(cfg, "rl_stabilize", False))
        self.pad_idx = pad_idx
        self.criterion = LabelSmoothing(smoothing, pad_idx)
        self.rl_criterion = BiasedKL(smoothing, pad_idx)
        # bucket order: everything but the encoder first.  Backward reaches the encoder last, so in data-parallel runs the
        # all-reduce of the first bucket overlaps the encoder's backward (split_backward).
        names = {id(p): n for n, p in self.agent.named_parameters()}
        bucket = trainable_bucket(self.agent)
        # Backward reaches the encoder layers last, the last layer first: phases = [everything downstream of the encoder,
        # encoder layer N-1, ..., encoder layer 0]; the all-reduce of a phase's bucket overlaps the next phase's backward.
        early = [p for p in bucket if not names[id(p)].startswith("bm_enc.")]
        enc_layers = list(self.agent.bm_enc.encoder.layers)
        n_enc = len(enc_layers)
        per_layer = [[p for p in bucket if names[id(p)].startswith(f"bm_enc.encoder.layers.{i}.")] for i in range(n_enc)]
        claimed = {id(p) for lp in per_layer for p in lp}
        per_layer[0] += [p for p in bucket if names[id(p)].startswith("bm_enc.") and id(p) not in claimed]
        # (the caption embedding last inside its bucket: its backward node is older than the encoder's, so one backward pass
        # reaches it AFTER the encoder -- the early Adam passes below leave it for the end of the step)
        late = [p for p in early if names[id(p)].startswith("emb_C.")]
        early = [p for p in early if not names[id(p)].startswith("emb_C.")] + late
        self.phase_params = [early] + [per_layer[i] for i in reversed(range(n_enc))]
        self.opt = FlatAdam([p for ph in self.phase_params for p in ph], lr=lr, weight_decay=weight_decay)
        self.opt.set_buckets([len(ph) for ph in self.phase_params])
        # parts of the optimizer pass: [head + fusion stacks | caption embedding | encoder layer N-1 | ... | encoder layer 0]
        pb = [0, len(early) - len(late), len(early)]
        for ph in self.phase_params[1:]:
            pb.append(pb[-1] + len(ph))
        self.opt.plan_bounds = pb
        self.opt.phased_direct = False      # set by capture() in the one-rank phased mode: gather_grads(bucket) leaves gradients in place
        self.early_params = early
        self.n_enc = n_enc
        self.split_backward = None          # None: split when a process group with more than one rank is active
        # one rank, optional (BMHRL_PHASED_ADAM=1): the backward runs in the same phases inside ONE graph and every bucket's
        # Adam pass starts on a side stream as soon as its phase is done, so the 221 MB update streams under the encoder's
        # backward.  Measured on MI355X at config 2: 5.88 ms against 5.73 ms for the plain step -- the phase boundaries join
        # every side stream of the backward three times and cost more than the 0.2 ms of Adam they hide; off by default.
        self.phased_adam = os.environ.get("BMHRL_PHASED_ADAM", "0") == "1"
        # one rank, optional (BMHRL_EARLY_ADAM=1): ONE backward pass, and the optimizer pass of a part starts on a side stream
        # the moment the backward has gone past it -- a tensor hook on the encoder-layer outputs fires when the layer below
        # starts its backward, i.e. after every node above it has been launched (the engine runs ready nodes newest first).
        # Only the Adam stream waits for the backward's streams; the backward itself joins nothing extra (what made the phased
        # form slower); the caption embedding and the first encoder layer are left for the end.  Bit-identical to the plain
        # order in deterministic mode (tests/test_split_backward_gpu.py).  Measured on MI355X at config 2, A/B/A/B on one box:
        # 4.953 / 4.916 ms with it, 4.897 / 4.926 ms without -- nothing: the chip is full during the encoder backward, so the
        # 0.29 ms of HBM-bound update moved beside it is added to it, not hidden.  Off by default.
        self.early_adam = os.environ.get("BMHRL_EARLY_ADAM", "0") == "1"
        self._early_armed = False
        self._early_pending = {}
        self._early_done = set()
        self._layer_out = {}
        self._keep_cuts = False
        for i, layer in enumerate(enc_layers):   # (V-stream, A-stream) after layer i: the cut between two backward phases
            # kept only while a phased step is being built (and dropped with its last phase): a layer output held past the end
            # of a step keeps that step's whole autograd graph alive -- memory, and objects of an eager pass that then die in
            # the middle of a later capture (DESIGN.md section 10 "r04")
            layer.register_forward_hook(lambda mod, inp, out, i=i: self._on_layer_out(i, out))
        self.modality = "audio_video"
        self.scratch = ScratchState()       # arena + operand pools of this trainer's steps: a captured step keeps their addresses
        self.graph = None
        self.static = None
        self.loss_weight = torch.ones((), device=self.device)   # token_weight() of the current batch (1 on one rank)
        # the device seed word of THIS trainer's steps (bumped by the step's first launch, read by every dropout / sampling
        # kernel): a captured step holds its address, so it lives with the trainer -- a process-wide word re-created by the next
        # trainer's constructor left the first trainer's graph bumping freed memory
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        SEEDS.dev = self.seed_dev

    # ------------------------------------------------------------------ one step, eager
    def _head(self, fs, captions):
        """The step's first launch (ops.batch_head): captions -> (trg_in, trg_y), every mask of the step with its doubled
        copy for the paired fusion stacks, the seed word and the optimisers' device step counters advanced.  CPU tensors:
        the plain form."""
        if not captions.is_cuda:
            SEEDS.dev.add_(1)
            trg_in, trg_y = captions[:, :-1].contiguous(), captions[:, 1:].contiguous()
            return trg_in, trg_y, make_masks(fs, trg_in, self.modality, self.pad_idx)
        B = captions.shape[0]
        counters = [self.opt.step_dev] + ([self.vopt.step_dev] if self.value_net is not None else [])
        vm2, am2, cm2, trg_in, trg_y = ops.batch_head(fs["rgb"], fs["audio"], captions, self.pad_idx, copies=2, bump64=SEEDS.dev,
                                                      bump32=counters)
        return trg_in, trg_y, {"V_mask": vm2[:B], "A_mask": am2[:B], "C_mask": cm2[:B], "_pair": (cm2, am2, vm2)}

    def _forward_loss(self, fs, trg_in, trg_y, rl=None, masks=None):
        if masks is None:
            if fs["rgb"].is_cuda:
                # every mask of the step, and their doubled copies for the paired fusion stacks, in one launch
                B = trg_in.shape[0]
                vm2, am2, cm2 = ops.make_masks(fs["rgb"], fs["audio"], trg_in, self.pad_idx, copies=2)
                masks = {"V_mask": vm2[:B], "A_mask": am2[:B], "C_mask": cm2[:B], "_pair": (cm2, am2, vm2)}
            else:
                masks = make_masks(fs, trg_in, self.modality, self.pad_idx)
        fused_tail = self.phase == "warmstart" and rl is None and trg_y.is_cuda and torch.is_grad_enabled()
        w = self.loss_weight if self._world_scale() != 1.0 else None
        if fused_tail:
            # the worker head runs log-softmax, loss and d logits as one launch (functional.request_head_loss); the gradient
            # this trainer starts every backward from is the constant of _unit_grad()
            request_head_loss(trg_y, float(self.criterion.smoothing), int(self.criterion.pad_idx), 1.0, w, self._unit_grad_dev(trg_y.device))
        try:
            pred, w_feat, m_feat, goals, seg = self.agent(((fs["rgb"], fs["flow"]), fs["audio"]), trg_in, masks)
        finally:
            cancel_head_loss()
        if self.phase == "warmstart" and rl is None and pred.is_cuda:
            # sum(LabelSmoothing) / n_tokens (x the rank's token weight) as one node: functional.TokenLossFn; the
            # prediction has no other differentiated consumer here, so the head's backward takes d logits from that node
            return TokenLossFn.apply(pred, trg_y, None, None, None, float(self.criterion.smoothing), int(self.criterion.pad_idx),
                                     1.0, w, True), pred
        loss_mask = trg_y != self.pad_idx
        n_tokens = loss_mask.sum()
        if self.phase != "warmstart":
            loss = self._rl_loss(pred, w_feat, m_feat, goals, seg, trg_y, loss_mask, n_tokens)
        elif rl is None:
            loss = torch.sum(self.criterion(pred, trg_y)) / n_tokens
        else:
            sampled, score = rl
            n_row = loss_mask.sum(-1, keepdim=True).expand_as(trg_y).float()
            rows, _ = self.rl_criterion.biased_kl_from_score(pred, trg_y, sampled, score, n_row)
            loss = torch.sum(rows) / (n_tokens * (4.0 / 20.0))
        if self._world_scale() != 1.0 and self.phase == "warmstart":
            loss = loss * self.loss_weight       # global n_tokens normalisation (token_weight); a device scalar: graph-safe
        return loss, pred

    def _rl_loss(self, pred, w_feat, m_feat, goals, seg, trg_y, loss_mask, n_tokens):
        """worker / manager RL step of the reference (:846-877): value head on detached features, biased KL of the sampled
        (worker) or arg-max (manager) tokens with their rewards, captioning loss / (n_tokens * 4/20) and the value head's
        masked MSE against the rewards.  The two losses reach disjoint parameters (the features are detached), so one
        backward pass serves both optimisers; sampling draws from a device seed word (fresh samples at every graph replay)."""
        from .epoch_loops.captioning_bmrl_loops import biased_kl
        worker = self.phase == "worker"
        expected = (self.value_net((w_feat.detach(), goals.detach())) if worker else self.value_net(m_feat.detach())).squeeze(-1)
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        rows, scores, _, _ = biased_kl(worker, pred, None, expected.detach(), trg_y, None, loss_mask, seg, pred.device,
                                       self.rl_criterion, self.stabilize, reward_fn=self.reward_fn, seed=12345, seed_dev=SEEDS.dev,
                                       row_offset=rank * trg_y.numel())       # ranks draw independent samples
        cap_loss = torch.sum(rows) / (n_tokens * (4.0 / 20.0))
        if self._world_scale() != 1.0:
            # more than one rank: only the captioning term is normalised by a token count (token_weight makes the averaged
            # gradient that of the gathered batch); the value loss is a plain mean over B x L (reference :871-873) -- ranks
            # hold equal B, so the average of the ranks' means IS the gathered mean and it must not be re-weighted
            cap_loss = cap_loss * self.loss_weight
        vmask = loss_mask.float() if worker else seg.detach().float()
        value_loss = (((expected - scores[0].float()) ** 2) * vmask).mean()
        self.last_value_loss = value_loss.detach()
        return cap_loss + value_loss

    def _sync_token_weight(self, captions):
        """before a step: this rank's share of the global token count -> self.loss_weight (in place: the captured step
        reads the same device scalar)"""
        if self._world_scale() != 1.0:
            self.loss_weight.copy_(token_weight((captions[:, 1:] != self.pad_idx).sum()))

    def step(self, fs, captions, rl=None):
        """zero_grad -> forward -> loss -> backward -> (all-reduce) -> Adam.  Returns the loss (device scalar).
        Called on the device's default stream, the step runs on this process's warm-up stream instead (forked from and joined
        into the default stream): an autograd graph built on the legacy default stream that is still referenced when a step
        is captured later makes hipStreamEndCapture of this runtime fault (tests/probes/step_then_capture.py, DESIGN.md
        section 10 "r04"); on any other stream it does not."""
        if captions.is_cuda and torch.cuda.current_stream(self.device) == torch.cuda.default_stream(self.device):
            cur, s = torch.cuda.current_stream(self.device), self._warm_stream()
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                loss = self._step(fs, captions, rl)
            cur.wait_stream(s)
            loss.record_stream(cur)
            return loss
        return self._step(fs, captions, rl)

    def _warm_stream(self):
        """one warm-up / eager-step stream per device and process (torch's pool of 32 streams wraps)"""
        s = CaptionTrainer._warm_streams.get(self.device)
        if s is None:
            s = CaptionTrainer._warm_streams[self.device] = torch.cuda.Stream(device=self.device)
        return s

    def _step(self, fs, captions, rl=None):
        self._early_armed = False
        self._sync_token_weight(captions)
        SEEDS.dev = self.seed_dev
        self._keep_cuts = False
        self._layer_out.clear()
        self.opt.zero_grad()
        if self.value_net is not None:
            self.vopt.zero_grad()
        SCRATCH.begin_step(self.device, self.scratch)
        trg_in, trg_y, masks = self._head(fs, captions)
        SHADOWS.refresh()
        loss, _ = self._forward_loss(fs, trg_in, trg_y, rl, masks)
        self._backward(loss)
        self.opt.gather_grads()
        if self.value_net is not None:
            self.vopt.gather_grads()
        SCRATCH.end_step()
        scale = self.opt.all_reduce()
        adv = captions.is_cuda
        self.opt.step(scale, dev_step_advanced=adv)
        self._plan_clean = self.scratch.last_spill == 0
        if self.value_net is not None:
            self.vopt.all_reduce()
            self.vopt.step(scale, dev_step_advanced=adv)
        return loss.detach()

    def _step_bucket(self, b):
        """the optimizer pass of gradient bucket b (phased Adam): bucket 0 is two parts of the table when the caption embedding
        has its own (FlatAdam.plan_bounds)"""
        n = len(self.opt._part_plans)
        parts = [b] if n == self.n_enc + 1 else ([0, 1] if b == 0 else [b + 1])
        for p in parts:
            self.opt.step_part(p, 1.0, dev_step_advanced=True, first=(p == 0), last=(p == n - 1))

    def _on_layer_out(self, i, out):
        if self._keep_cuts:
            self._layer_out[i] = out
        if self._early_armed:
            ts = [t for t in out if torch.is_tensor(t) and t.requires_grad]
            part = 0 if i == self.n_enc - 1 else self.n_enc - i      # the part whose gradients are complete once layer i starts
            self._early_pending[part] = len(ts)
            for t in ts:
                t.register_hook(lambda g, part=part: self._early_fire(part))
        return None

    def _early_ok(self) -> bool:
        o = self.opt
        plan = o.__dict__.get("_seg_plan")
        return (self.early_adam and self.device.type == "cuda" and self._world_scale() == 1.0 and not self._split()
                and o.direct_grads and o.fused_shadows and not o.__dict__.get("_homes", False)
                and plan is not None and plan[2] is not None and len(o.__dict__.get("_part_plans", ())) == self.n_enc + 2
                and self.__dict__.get("_plan_clean", False) and self.scratch.last_spill == 0
                and torch.is_grad_enabled())

    def _adam_stream(self):
        side = CaptionTrainer._adam_streams.get(self.device)
        if side is None:
            side = CaptionTrainer._adam_streams[self.device] = torch.cuda.Stream(device=self.device)
        return side

    def _early_fire(self, part):
        """(autograd engine thread, before the first backward node of the encoder layer below `part` runs)"""
        self._early_pending[part] -= 1
        if self._early_pending[part] > 0 or part in self._early_done:
            return None
        self._early_done.add(part)
        from .model.bm_hrl_agent import BMEncoderLayer, BMFusionLayer, BMHrlAgent
        side = self._adam_stream()
        cur = torch.cuda.current_stream(self.device)
        capturing = torch.cuda.is_current_stream_capturing()
        side.wait_stream(cur)
        for s in (self._main_stream, BMEncoderLayer._side, BMFusionLayer._side, BMHrlAgent._critic_streams.get(self.device)):
            if s is None or s == cur:
                continue
            if capturing:                    # only streams that are part of this capture may be waited for
                with torch.cuda.stream(s):
                    if not torch.cuda.is_current_stream_capturing():
                        continue
            side.wait_stream(s)
        with torch.cuda.stream(side):
            self.opt.step_part(part, 1.0, dev_step_advanced=True, first=(part == 0), last=False)
        return None

    def _backward(self, loss):
        """d loss / d every parameter of the trainer's optimisers, left in p.grad.  torch.autograd.grad rather than
        loss.backward(): backward() delivers through the parameters' AccumulateGrad nodes, and such a node keeps the stream it
        was first used on -- a parameter touched by an eager step on the default stream pulls that stream into every later
        capture (autograd syncs the producing side stream with it: a nested fork / join, the shape hipStreamEndCapture of this
        runtime does not survive).  grad() returns the gradients on the streams that produced them."""
        leaves = [p for p in self.opt.params if p.requires_grad]
        if self.value_net is not None:
            leaves += [p for p in self.vopt.params if p.requires_grad]
        outs = torch.autograd.grad(loss, leaves, grad_outputs=self._unit_grad(loss), allow_unused=True)
        for p, g in zip(leaves, outs):
            p.grad = g

    def _unit_grad(self, loss):
        """d loss / d loss as a constant kept for the trainer's lifetime (autograd otherwise fills a fresh one every step)"""
        if loss.dtype == torch.float32:
            return self._unit_grad_dev(loss.device)
        return torch.ones((), device=loss.device, dtype=loss.dtype)

    def _unit_grad_dev(self, device):
        one = self.__dict__.get("_one")
        if one is None or one.device != device:
            one = self._one = torch.ones((), device=device, dtype=torch.float32)
        return one

    # ------------------------------------------------------------------ whole-step HIP graph
    _warm_streams = {}
    _adam_streams = {}
    _zero_streams = {}

    def _zero_stream(self):
        """side stream of the scratch arena's fill (StepScratch.begin_step); one per device and process"""
        if not os.environ.get("BMHRL_ZERO_STREAM", "0") == "1":
            return None
        s = CaptionTrainer._zero_streams.get(self.device)
        if s is None:
            s = CaptionTrainer._zero_streams[self.device] = torch.cuda.Stream(device=self.device)
        return s

    def capture(self, fs, captions, warmup: int = 3):
        """Capture step() for static shapes; afterwards replay(fs, captions) copies the inputs into the captured
        buffers and launches the graph.  The all-reduce stays outside the graph (between two captured halves) when
        a process group is active, so RCCL sees ordinary stream launches."""
        self.static = {k: v.clone() for k, v in fs.items()}
        self.static["captions"] = captions.clone()
        self.static_loss = torch.zeros((), device=self.device)
        self.opt.phased_direct = self._phased_one_rank()
        s = self._warm_stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            homes = self._grad_homes_wanted()
            if self.scratch.arena is None or homes:
                # size this trainer's scratch arena first: a pass without the optimizer (weights untouched).  The warm-up
                # steps below then already work on the arena slices the captured step will use -- the Adam pass reads the
                # gradients where autograd leaves them, and its table holds those addresses.
                # Data parallel: the same pass records which allocation every leaf gradient comes from; from the next pass on
                # they are produced in the flat bucket (FlatAdam.adopt_homes moves the parameters of a stacked allocation
                # next to each other inside their bucket first -- before any captured address exists)
                self._sync_token_weight(self.static["captions"])
                self.scratch.record = homes
                self._graph_body_a()
                if self._split():
                    for j in range(1, self.n_enc + 1):
                        self._graph_body_phase(j)
                if homes:
                    self.scratch.record = False
                    self.grad_elems_in_place = self.opt.adopt_homes(self.scratch)
                    if self.value_net is not None:
                        self.vopt.adopt_homes(self.scratch)
                self.opt.zero_grad()
                if self.value_net is not None:
                    self.vopt.zero_grad()
                self.seed_dev.sub_(1)         # (the pass does not count as a step: dropout masks / samples continue as if it had not run)
                self.opt.step_dev.sub_(1)     # (nor do the optimisers' device counters, which the step's first launch advanced)
                if self.value_net is not None:
                    self.vopt.step_dev.sub_(1)
            for _ in range(max(1, warmup)):   # at least one eager pass: lazily built tables / shadows must exist
                self._sync_token_weight(self.static["captions"])
                self._graph_body_a()
                if self._split():
                    for j in range(1, self.n_enc + 1):
                        self._graph_body_phase(j)
                if self._phased_one_rank():
                    self.opt.gather_grads()                 # records where every gradient lives: the plan of the phased passes
                scale = self.opt.all_reduce()               # warm-up steps are real steps: replicas stay identical
                if self.value_net is not None:
                    self.vopt.all_reduce()
                self._graph_body_b(scale)
            SHADOWS.refresh()                 # builds the segment table of the one-launch shadow refresh (a host -> device
                                              # copy, not allowed while capturing); the captured body reuses it
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        phased = self._phased_one_rank()
        alone = self._world_scale() == 1.0 and (phased or not self._split())
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a):
            self._graph_body_a()
            if phased:
                # ONE graph: backward phase j on the capture stream, the Adam pass of bucket j - 1 beside it on a side
                # stream (forked from and joined into the capture stream only: DESIGN.md section 10)
                main = torch.cuda.current_stream()
                side = CaptionTrainer._adam_streams.get(self.device)
                if side is None:
                    side = CaptionTrainer._adam_streams[self.device] = torch.cuda.Stream(device=self.device)
                for j in range(1, self.n_enc + 1):
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        self._step_bucket(j - 1)
                    self._graph_body_phase(j)
                main.wait_stream(side)
                self._step_bucket(self.n_enc)
                if self.value_net is not None:
                    self.vopt.step(1.0, dev_step_advanced=True)
            elif alone:                       # no all-reduce to leave room for: the optimizer joins the same graph
                self._graph_body_b(1.0)
        if alone:
            self.graph_b = None
            self.graph = True
            return
        if self._split():
            self.graph_a2 = []                # one graph per encoder layer, in backward order
            for j in range(1, self.n_enc + 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.graph_a.pool()):
                    self._graph_body_phase(j)
                self.graph_a2.append(g)
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
            self._graph_body_b(self._world_scale())
        self.graph = True

    @staticmethod
    def _world_scale():
        return 1.0 / dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1.0

    def _grad_homes_wanted(self) -> bool:
        """leaf gradients written straight into the flat bucket (FlatAdam.adopt_homes): with more than one rank, where the
        bucket is what the all-reduce reads; BMHRL_GRAD_HOMES=1 / 0 forces it on (one-rank rehearsal) / off (A/B).  Adopted
        once per trainer."""
        if self.scratch.homes or self.device.type != "cuda":
            return False
        env = os.environ.get("BMHRL_GRAD_HOMES")
        if env in ("0", "1"):
            return env == "1"
        return self._world_scale() != 1.0

    def _phased_one_rank(self) -> bool:
        return self.phased_adam and self.split_backward is None and self._world_scale() == 1.0 and self.device.type == "cuda" \
            and self.opt.direct_grads and self.opt.fused_shadows and os.environ.get("BMHRL_SPLIT_BACKWARD") is None

    def _split(self) -> bool:
        if self._phased_one_rank():
            return True
        if self.split_backward is None:
            import os
            if os.environ.get("BMHRL_SPLIT_BACKWARD") in ("0", "1"):      # tuning / rehearsal override
                return os.environ["BMHRL_SPLIT_BACKWARD"] == "1"
            return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        return bool(self.split_backward)

    def _graph_body_a(self):
        st = self.static
        cap = st["captions"]
        SEEDS.dev = self.seed_dev
        self.opt.zero_grad()
        if self.value_net is not None:
            self.vopt.zero_grad()
        SCRATCH.begin_step(self.device, self.scratch, zero_stream=self._zero_stream())
        self._keep_cuts = self._split()
        self._layer_out.clear()
        self._early_armed = self._early_ok()
        self._early_pending, self._early_done = {}, set()
        self._main_stream = torch.cuda.current_stream(self.device)
        trg_in, trg_y, masks = self._head(st, cap)
        if not self.opt.fused_shadows:
            SHADOWS.invalidate()
        for o in (self.opt, getattr(self, "vopt", None)):
            if o is not None:
                o.mark_uncovered_stale()
        SHADOWS.refresh()                     # (fused_shadows: the Adam pass keeps the shadows current -- usually nothing to do)
        loss, _ = self._forward_loss(st, trg_in, trg_y, None, masks)
        SCRATCH.join_zero()
        if self._split():
            # phase 0 of the backward: everything downstream of the encoder output (head, both fusion stacks, embedding)
            # (RL phases: the value head hangs off detached features -- its parameters are leaves of this first phase too; in
            # the manager phase the encoder is frozen and the cut carries no gradient: the later phases then have nothing to do)
            cut = [t for t in self._layer_out[self.n_enc - 1] if t.requires_grad]
            vparams = list(self.vopt.params) if self.value_net is not None else []
            leaves = self.early_params + vparams
            outs = torch.autograd.grad(loss, leaves + cut, grad_outputs=self._unit_grad(loss), retain_graph=bool(cut),
                                       allow_unused=True)
            for p, g in zip(leaves, outs):
                p.grad = g
            self._cut = (cut, list(outs[len(leaves):]))
            self.opt.gather_grads(0)
            if self.value_net is not None:
                self.vopt.gather_grads()
        else:
            self._backward(loss)
            self.opt.gather_grads()
            if self.value_net is not None:
                self.vopt.gather_grads()
            SCRATCH.end_step()
        # the loss of the captured step lives in the graph's own pool: the tensor IS the static output (no copy launch)
        self.static_loss = loss.detach()

    def _graph_body_phase(self, j: int):
        """phase j = 1..n_enc of the backward (split mode): encoder layer n_enc - j, from the gradients of its outputs
        to its parameters and to the outputs of the layer below"""
        idx = self.n_enc - j
        cut, gcut = self._cut
        pairs = [(t, g) for t, g in zip(cut, gcut) if g is not None]
        params = self.phase_params[j]
        below = list(self._layer_out[idx - 1]) if idx > 0 else []
        if pairs and params:
            outs = torch.autograd.grad([t for t, _ in pairs], params + below, grad_outputs=[g for _, g in pairs],
                                       retain_graph=idx > 0, allow_unused=True)
        else:                                     # frozen encoder (manager phase): nothing flows below the cut
            outs = [None] * (len(params) + len(below))
        for p, g in zip(params, outs):
            p.grad = g
        self._cut = (below, list(outs[len(params):])) if idx > 0 else None
        self.opt.gather_grads(j)
        if idx == 0:
            self._layer_out.clear()
            self._keep_cuts = False
            SCRATCH.end_step()

    def _graph_body_b(self, scale):
        if self._early_armed:
            # some parts of the update are already running beside the backward (_early_fire): check that every gradient sat
            # where their tables read it, then the rest -- the caption embedding, the first encoder layer, anything whose hook
            # did not fire -- on the step's own stream
            self._early_armed = False
            plan = self.opt.__dict__.get("_seg_plan")
            if self.opt.__dict__.get("_direct_ptrs") is None or plan is None or plan[2] != self.opt._direct_ptrs:
                raise RuntimeError("early Adam: a gradient is not where the optimizer's table reads it (the weights of the parts "
                                   "already updated are wrong now); run with BMHRL_EARLY_ADAM=0")
            torch.cuda.current_stream(self.device).wait_stream(self._adam_stream())
            n_parts = len(self.opt._part_plans)
            rest = [p for p in range(n_parts) if p not in self._early_done]
            for k, p in enumerate(rest):
                self.opt.step_part(p, 1.0, dev_step_advanced=True, first=(p == 0 and 0 not in self._early_done),
                                   last=(k == len(rest) - 1))
            if 0 not in rest and 0 not in self._early_done:
                raise RuntimeError("early Adam: part 0 never ran")
            if self.value_net is not None:
                self.vopt.step(scale, dev_step_advanced=True)
            return
        self.opt.step(scale, dev_step_advanced=True)              # (the counters moved in the step's first launch: _head)
        # the table this pass (re)built reads the gradients where THIS pass left them: valid for the next pass when all of them
        # came out of the scratch arenas (a deterministic bump allocator) -- what _early_ok() asks for
        self._plan_clean = self.scratch.last_spill == 0
        if self.value_net is not None:
            self.vopt.step(scale, dev_step_advanced=True)

    def replay(self, fs=None, captions=None):
        if fs is not None:
            for k in ("rgb", "flow", "audio"):
                self.static[k].copy_(fs[k])
            self.static["captions"].copy_(captions)
        self._sync_token_weight(self.static["captions"])
        for o in (self.opt, getattr(self, "vopt", None)):     # the captured Adam pass changes the weights behind the host's back
            if o is not None:
                o.generation += 1
        self.graph_a.replay()
        if self.graph_b is None:                  # single process: forward, backward and Adam are one graph
            return self.static_loss
        if self._split():
            works = [self.opt.all_reduce_part(0)]             # overlaps the encoder backward below
            if self.value_net is not None:
                works.append(self.vopt.all_reduce_part(None))
            for j, g in enumerate(self.graph_a2, start=1):
                g.replay()
                works.append(self.opt.all_reduce_part(j))     # layer by layer: only the first layer's bucket is exposed
            for w in works:
                if w is not None:
                    w.wait()
        else:
            self.opt.all_reduce()
            if self.value_net is not None:
                self.vopt.all_reduce()
        self.graph_b.replay()
        return self.static_loss
