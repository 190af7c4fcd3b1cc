"""Autograd building blocks of the bimodal transformer, each a coarse torch.autograd.Function whose forward and
backward are sequences of C-ABI kernel launches (bmhrl_amd.ops).  Only fp32 "stream" tensors and parameters cross
the autograd boundary; bf16 operands, softmax statistics and dropout seeds stay inside a block.

Precision plan (north_star: logits within 1e-3 relative): residual streams, LayerNorm / softmax statistics, the
vocabulary logits and all reductions are fp32; GEMM and attention operands are bf16 with fp32 MFMA accumulation.
"""
from __future__ import annotations

import math
import itertools
import os
import weakref
from typing import Optional

import torch

from . import ops
from .ops import pad8

_BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------------ parameter shadows
class ShadowCache:
    """bf16 copies (row stride padded to 8) of fp32 weights and concatenated fp32 biases, refreshed when a
    parameter's version counter moves (optimizer step, load_state_dict) or the cache is invalidated."""

    def __init__(self):
        self.w = {}
        self.b = {}
        self.split = {}          # key -> block width of the entries of self.w that are [hi | lo | hi] split shadows
        self.epoch = 0

    def invalidate(self):
        self.epoch += 1

    @staticmethod
    def _alive(ent, params):
        return ent is not None and all(r() is p for r, p in zip(ent[2], params))

    def _version(self, params, kind=None, key=None):
        """what a shadow of these parameters was made from.  autograd's version counters do not see the optimizer kernel
        (it writes through raw pointers), so a parameter owned by a FlatAdam also contributes that optimizer's generation
        -- it moves with every step / graph replay -- unless the optimizer's own pass maintains this very entry."""
        gens = []
        for p in params:
            o = getattr(p, "_bmhrl_owner", None)
            o = o() if o is not None else None
            gens.append(0 if o is None else o.shadow_generation(kind, key))
        return (self.epoch,) + tuple(p._version for p in params) + tuple(gens) + (params[0].device,)

    def weight(self, *params):
        key = tuple(id(p) for p in params)
        ver = self._version(params, 1, key)
        ent = self.w.get(key)
        if not self._alive(ent, params):   # id() of a freed parameter can be reused by a new one
            ent = None
        if ent is not None and ent[0] == ver:
            return ent[1]
        K = params[0].shape[1]
        N = sum(p.shape[0] for p in params)
        buf = ent[1] if ent is not None and ent[1].device == params[0].device else ops.bf16_zeros(N, K, params[0].device)
        ld = buf.shape[1]
        off = 0
        for p in params:
            ops.cast_bf16(p.detach(), K, buf, ld, p.shape[0], K, y_off=off * ld)
            off += p.shape[0]
        self.w[key] = (ver, buf, tuple(weakref.ref(p) for p in params))
        return buf

    @staticmethod
    def split_part(K: int) -> int:
        """column-block width of a split shadow: K rounded up to the GEMM's 64-wide k-step"""
        return (K + 63) & ~63

    def weight_split3(self, p):
        """[hi | lo | hi] bf16 shadow of ONE fp32 weight (N, K) -> (N, 3 * split_part(K)): hi = bf16(w), lo = bf16(w - hi)
        (ops.cast_split3_bf16, lo_slot 1).  Lives in the same table as the plain shadows under the key (id(p), -3); the
        refresh / Adam passes find the split layout in `self.split` and write all three blocks."""
        key = (id(p), -3)
        ver = self._version((p,), 1, key)
        ent = self.w.get(key)
        if not self._alive(ent, (p,)):
            ent = None
        if ent is not None and ent[0] == ver:
            return ent[1]
        N, K = p.shape
        part = self.split_part(K)
        buf = ent[1] if ent is not None and ent[1].device == p.device else torch.zeros(N, 3 * part, dtype=_BF16, device=p.device)
        ops.cast_split3_bf16(p.detach(), K, buf, 3 * part, part, 1, N, K)
        self.w[key] = (ver, buf, (weakref.ref(p),))
        self.split[key] = part
        return buf

    def bias(self, *params):
        if len(params) == 1:
            return params[0].detach()
        key = tuple(id(p) for p in params)
        ver = self._version(params, 0, key)
        ent = self.b.get(key)
        if self._alive(ent, params) and ent[0] == ver:
            return ent[1]
        if self._alive(ent, params) and ent[1].device == params[0].device:
            buf = ent[1]                       # re-made in place: captured graphs (decoder, trainer) keep reading this address
            torch.cat([p.detach() for p in params], out=buf)
        else:
            buf = torch.cat([p.detach() for p in params])
        self.b[key] = (ver, buf, tuple(weakref.ref(p) for p in params))
        return buf


    # ---- refresh of every stale entry in one launch
    def _is_current(self, kind, key, e):
        params = tuple(r() for r in e[2])
        return all(p is not None for p in params) and e[0] == self._version(params, kind, key)

    def mark_stale(self, kind, key):
        """the parameters of this entry were written behind autograd's back (optimizer kernel) without their shadow"""
        store = self.w if kind else self.b
        e = store.get(key)
        if e is not None:
            store[key] = (None, e[1], e[2])

    def refresh(self):
        """Re-cast every STALE cached weight shadow and concatenated bias with ONE kernel launch and mark the entries
        current, so the lookups of the following forward / backward are hits.  (The optimizer keeps the shadows of the
        parameters it owns current itself -- bmhrl_adam_segments writes them in the Adam pass -- so in a training step this
        usually finds nothing to do.)  Segment tables are cached per set of stale entries."""
        live = []
        for kind, store in ((1, self.w), (0, self.b)):
            for k, e in store.items():
                params = tuple(r() for r in e[2])          # strong references from here on (a collection may run any time)
                if all(p is not None for p in params) and e[1].is_cuda and e[0] != self._version(params, kind, k):
                    live.append((k, e, kind, params))
        if not live:
            return
        sig = tuple((k, kind, e[1].data_ptr()) + tuple(p.data_ptr() for p in params) for k, e, kind, params in live)
        plans = self.__dict__.setdefault("_plans", {})
        if sig not in plans:
            if len(plans) > 16:
                plans.clear()
            rows_, blk = [], 0
            for k, e, kind, params in live:
                buf, off = e[1], 0
                for p in params:
                    if kind:      # weight: (N_p, K) fp32 -> rows [off, off + N_p) of the (N, ld) bf16 shadow
                        n, kk, ld = p.shape[0], p.shape[1], buf.shape[1]
                        rows_.append([p.data_ptr(), buf.data_ptr() + 2 * off * ld, n, kk, ld | (self.split.get(k, 0) << 32), blk])
                        off += n
                        blk += (n * kk + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
                    else:         # bias: fp32 copy into the concatenated vector
                        n = p.numel()
                        rows_.append([p.data_ptr(), buf.data_ptr() + 4 * off, 1, n, 0, blk])
                        off += n
                        blk += (n + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
            dev = live[0][1][1].device
            plans[sig] = (torch.tensor(rows_, dtype=torch.int64).to(dev), len(rows_), blk)
        table, n_seg, n_blk = plans[sig]
        ops.cast_segments(table, n_seg, n_blk)
        for k, e, kind, params in live:
            (self.w if kind else self.b)[k] = (self._version(params, kind, k), e[1], e[2])


SHADOWS = ShadowCache()


class DropoutSeeds:
    """Per-site dropout seeds.  Every site draws `base + counter`; under graph replay the device word `dev`
    (incremented inside the captured step) is added by the kernels so masks differ between replays."""

    def __init__(self, base: int = 0x5EED):
        self.base = base
        self.counter = 0
        self.dev: Optional[torch.Tensor] = None

    def next(self) -> int:
        self.counter += 1
        return (self.base * 0x9E3779B97F4A7C15 + self.counter * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF


SEEDS = DropoutSeeds()


# ------------------------------------------------------------------------------------------------ step-scoped scratch
_ARENA_POISON = os.environ.get("BMHRL_ARENA_POISON", "0") == "1"


class ScratchState:
    """What a StepScratch hands out, owned by whoever runs the steps (a CaptionTrainer): the fp32 arena, the bf16 pools and
    their sizes.  A captured step bakes the ADDRESSES of these buffers into its graph, so they must live exactly as long as
    their owner -- with one process-wide state, a second trainer that needed a larger arena freed the one a still-live graph
    of the first trainer was reading (r02: module-global arena)."""

    def __init__(self):
        self.arena: Optional[torch.Tensor] = None
        self.need = 0
        self.raw: Optional[torch.Tensor] = None       # never zeroed: outputs their producer writes whole (f32(zero=False))
        self.need_raw = 0
        self.pool = {}
        self.sync = None                               # the four sync words of ops.head_loss (re-armed by the kernel itself)
        # data-parallel steps: leaf gradients are produced IN their slices of the optimiser's flat bucket instead of the arena
        # (train.FlatAdam.adopt_homes).  homes: (arena kind, element offset) of an allocation of the step -> its bucket view;
        # home_buckets: the buckets, zeroed with the arena at the start of a step; log: allocations of the current step
        # [(kind, offset, elements, data_ptr, storage)] while record is set (the storage is kept so that no address repeats; not the
        # tensor: autograd only adopts a gradient nobody else holds) (the warm-up pass the homes are derived from)
        self.last_spill = -1                           # elements the last step could not take from the arenas (-1: no step yet)
        self.homes = {}
        self.home_buckets = []
        self.record = False
        self.log = []


class StepScratch:
    """Zero-initialised scratch handed out inside ONE training step (between begin_step() and end_step(), as
    bmhrl_amd.train.CaptionTrainer does), so that a step issues one fill instead of ~250:

      * f32(): slices of one fp32 arena that begin_step() zeroes with a single launch -- weight / bias / LayerNorm
        gradient accumulators (split-K GEMMs and column sums add into them with atomics);
      * bf16(): (rows, pad8(cols)) bf16 operand buffers whose padding columns must be zero.  Kernels never write the
        padding, so a buffer zeroed when it was created stays valid: the pool hands the same buffers out again, in
        the same order, every step.

    Outside a step (unit tests, inference) both fall back to freshly zeroed tensors.  The first step sizes the arena,
    so everything is allocated before a HIP graph capture of the step (capture runs warm-up steps first).  The buffers
    themselves belong to the ScratchState passed to begin_step() (one per trainer); without one, a process-wide default."""

    def __init__(self):
        self.armed = False
        self.state = self.default_state = ScratchState()
        self.off = 0
        self.spill = 0
        self.off_raw = 0
        self.spill_raw = 0
        self.cursor = {}
        self.memo = {}
        self.memo_hits = 0           # memo_bf16 calls served without a cast (a producer's offer or an earlier call)
        self._zeroing = None
        self._zero_origin = None

    # (the buffers live in the bound state)
    arena = property(lambda self: self.state.arena, lambda self, v: setattr(self.state, "arena", v))
    need = property(lambda self: self.state.need, lambda self, v: setattr(self.state, "need", v))
    pool = property(lambda self: self.state.pool)

    def begin_step(self, device, state: Optional[ScratchState] = None, zero_stream=None):
        """zero_stream: run the arena's fill there (forked from the current stream) instead of in line -- only the backward
        pass touches the arena, so the fill (200 MB at the reference widths) overlaps the forward; the caller joins it with
        join_zero() before backward (f32() joins by itself should something ask for arena memory earlier)."""
        device = torch.device(device)
        if device.type != "cuda":
            return
        self.state = state if state is not None else self.default_state
        self._zeroing = None
        self._zero_origin = torch.cuda.current_stream() if zero_stream is not None else None
        if self.arena is None or self.arena.device != device or self.arena.numel() < self.need:
            self.arena = torch.zeros(self.need, device=device) if self.need else None
        elif self.off_of_last_step():
            if zero_stream is not None:
                zero_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(zero_stream):
                    self.arena[:self.off_of_last_step()].zero_()
                self._zeroing = zero_stream
            else:
                self.arena[:self.off_of_last_step()].zero_()
        st = self.state
        for bucket in st.home_buckets:                  # (gradients accumulate into / skip parts of their bucket slices too)
            if zero_stream is not None:
                if self._zeroing is None:
                    zero_stream.wait_stream(torch.cuda.current_stream())
                    self._zeroing = zero_stream
                with torch.cuda.stream(zero_stream):
                    bucket.zero_()
            else:
                bucket.zero_()
        st.log = []
        if st.need_raw and (st.raw is None or st.raw.device != device or st.raw.numel() < st.need_raw):
            st.raw = torch.empty(st.need_raw, device=device)
        if st.raw is not None and _ARENA_POISON:
            st.raw.fill_(float("nan"))            # (test aid: a consumer that accumulates into "overwritten" memory shows up)
        self.off_raw = self.spill_raw = 0
        self.off = self.spill = 0
        self.cursor = {}
        self.memo = {}
        self.armed = True

    def sync_words(self, device) -> torch.Tensor:
        """four zeroed int32 words for ops.head_loss (64-bit row sum, arrival count, non-finite flags).  Inside a step they belong
        to the bound ScratchState -- two trainers (or two captured graphs replaying on different streams) never share the
        accumulator --; outside a step one set per device."""
        if self.armed:
            st = self.state
            if st.sync is None or st.sync.device != device:
                st.sync = torch.zeros(4, dtype=torch.int32, device=device)
            return st.sync
        c = _HEAD_LOSS_COUNTER.get(device)
        if c is None:
            c = _HEAD_LOSS_COUNTER[device] = torch.zeros(4, dtype=torch.int32, device=device)
        return c

    def off_of_last_step(self) -> int:
        """elements of the bound state's arena a step may have written (its own high-water mark)"""
        return min(self.state.need, self.arena.numel()) if self.arena is not None else 0

    def join_zero(self):
        """the arena's fill (begin_step(zero_stream=...)) is joined into the stream it was forked FROM -- the step's own stream,
        which every side stream of the backward is forked from later -- and, when somebody asks for arena memory from another
        stream first, into that stream as well (a join into the asking stream alone would leave the step's stream unordered
        against the fill)"""
        if self._zeroing is not None:
            cur = torch.cuda.current_stream()
            origin = self._zero_origin if self._zero_origin is not None else cur
            origin.wait_stream(self._zeroing)
            if cur != origin:
                cur.wait_stream(self._zeroing)
            self._zeroing = None

    def end_step(self):
        self.join_zero()
        self.need = max(self.need, self.off + self.spill)
        self.state.need_raw = max(self.state.need_raw, self.off_raw + self.spill_raw)
        self.state.last_spill = self.spill + self.spill_raw      # 0: every request of this step came from the arenas
        self.armed = False

    def f32(self, *shape, device, zero: bool = True) -> torch.Tensor:
        """zero=False: the caller's kernel stores every element (e.g. a weight-gradient GEMM that ops.gemm_overwrites()
        says runs without a K split) -- the slice comes from a second arena that is never filled"""
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) & ~3
        st = self.state
        if not zero:
            r = st.raw
            if not self.armed or r is None or r.device != device or self.off_raw + n4 > r.numel():
                t = torch.empty(*shape, device=device)
                if self.armed:
                    if st.record:     # (the offset this allocation gets once the arena has been sized: every request so far fits)
                        st.log.append((1, self.off_raw + self.spill_raw, n, t.data_ptr(), t.untyped_storage()))
                    self.spill_raw += n4
                return t
            off = self.off_raw
            self.off_raw += n4
            home = st.homes.get((1, off + self.spill_raw))
            if home is not None and home.numel() == n:
                return home.view(*shape)
            t = r[off:off + n].view(*shape)
            if st.record:
                st.log.append((1, off + self.spill_raw, n, t.data_ptr(), t.untyped_storage()))
            return t
        a = self.arena
        if self._zeroing is not None:
            self.join_zero()
        if not self.armed or a is None or a.device != device or self.off + n4 > a.numel():
            t = torch.zeros(*shape, device=device)
            if self.armed:
                if st.record:
                    st.log.append((0, self.off + self.spill, n, t.data_ptr(), t.untyped_storage()))
                self.spill += n4
            return t
        off = self.off
        self.off += n4
        home = st.homes.get((0, off + self.spill))
        if home is not None and home.numel() == n:
            return home.view(*shape)
        t = a[off:off + n].view(*shape)
        if st.record:
            st.log.append((0, off + self.spill, n, t.data_ptr(), t.untyped_storage()))
        return t

    def memo_bf16(self, t: torch.Tensor, rows: int, cols: int, copies: int = 1) -> torch.Tensor:
        """bf16 (copies * rows, pad8(cols)) copy of the fp32 tensor t (laid out `copies` times back to back: the paired
        fusion stacks address the memory as 2 B samples); inside a step the copy is made once per distinct tensor
        (the encoder memory is read by every fusion layer of both stacks)."""
        key = (t.data_ptr(), t._version, rows, cols, copies)
        if self.armed:
            hit = self.memo.get(key)
            if hit is not None and hit[0]() is not None:       # (same address + version as a tensor that is still alive)
                self.memo_hits += 1
                return hit[1]
        buf = self.bf16(copies * rows, cols, t.device)
        ops.cast_bf16_copies(t.contiguous(), cols, buf, buf.shape[1], rows, cols, copies, rows * buf.shape[1])
        if self.armed:
            self.memo[key] = (weakref.ref(t), buf)
        return buf

    def offer_bf16(self, t: torch.Tensor, rows: int, cols: int, buf: torch.Tensor):
        """the producer of the fp32 tensor t wrote its bf16 copy `buf` ((rows, pad8(cols)), from the same epilogue): later
        memo_bf16(t, rows, cols) calls of this step take it instead of casting (the self-attention outputs of an encoder layer
        are the other modality's memory)"""
        if self.armed:
            self.memo[(t.data_ptr(), t._version, rows, cols, 1)] = (weakref.ref(t), buf)

    def memo_mem(self, t: torch.Tensor, B: int, Sk: int, dm: int):
        """(bf16 copy (B * Sk, dm), per-sample transposed bf16 copy (B, dm, ldt), ldt) of the fp32 memory t (B, Sk, dm) for
        ops.memory_attention -- one launch, made once per distinct tensor inside a step (both fusion layers of both stacks
        read the same encoder output)"""
        key = (t.data_ptr(), t._version, B, Sk, dm, "T")
        if self.armed:
            hit = self.memo.get(key)
            if hit is not None and hit[0]() is t:
                return hit[1]
        ldt = (Sk + 15) & ~15
        y = torch.empty(B * Sk, dm, dtype=_BF16, device=t.device)
        yt = torch.empty(B, dm, ldt, dtype=_BF16, device=t.device)
        ops.cast_memory(t.contiguous(), y, yt, B, Sk, dm, ldt)
        if self.armed:
            self.memo[key] = (weakref.ref(t), (y, yt, ldt))
        return y, yt, ldt

    def zeroed_bf16(self, rows: int, cols: int, device) -> torch.Tensor:
        """(rows, cols) bf16 buffer that was zero when created and whose users only ever write the same body columns
        (pooled inside a step like bf16(); a fresh zero tensor outside)"""
        if not self.armed:
            return torch.zeros(rows, cols, dtype=_BF16, device=device)
        key = (rows, cols, device, "z")
        lst = self.pool.setdefault(key, [])
        i = self.cursor.get(key, 0)
        if i == len(lst):
            lst.append(torch.zeros(rows, cols, dtype=_BF16, device=device))
        self.cursor[key] = i + 1
        return lst[i]

    def bf16(self, rows: int, cols: int, device) -> torch.Tensor:
        if cols % 8 == 0 or not self.armed:
            return ops.bf16_zeros(rows, cols, device)
        key = (rows, cols, device)
        lst = self.pool.setdefault(key, [])
        i = self.cursor.get(key, 0)
        if i == len(lst):
            lst.append(ops.bf16_zeros(rows, cols, device))
        self.cursor[key] = i + 1
        return lst[i]


SCRATCH = StepScratch()


def _mask_u8(mask: Optional[torch.Tensor]):
    """(B,1,Sk) or (B,Sq,Sk) bool/byte mask -> contiguous byte tensor + (batch stride, row stride)."""
    if mask is None:
        return None, 0, 0
    m = mask
    if m.dtype != torch.bool and m.dtype != torch.uint8:
        m = m != 0
    m = m.contiguous()
    B, R, Sk = m.shape
    return m, R * Sk, (Sk if R > 1 else 0)


# ------------------------------------------------------------------------------------------------ attention core
def _use_flash(dk: int, Sq: int, Sk: int) -> bool:
    """fused kernel: head dimension 256, enough queries to fill it, and a memory within the kernels' key limit (longer
    ones take the materialised GEMM path)"""
    return dk == 256 and Sq >= 128 and Sk <= ops.attention_max_keys()


# one-launch few-query memory attention core (csrc/memory_attention.hip) of the paired fusion stacks.  Alone it is slower than
# the three launches it replaces on the video side (44 vs 35 us; audio side 28 vs 33) -- it streams the memory from L2 straight
# into the A operand --, but in the captured step nothing changes (r04 A/B/A/B on one box: 4.907 / 4.913 ms with it, 4.906 /
# 4.918 without: the caption side is not where the step's time goes) and 16 launches leave the step (384 -> 368 kernels), so it
# is ON by default since r04.  BMHRL_FUSED_MEMATTN=0: the GEMM path; =2: the audio side (dm = 128) only.
FUSED_MEMATTN = os.environ.get("BMHRL_FUSED_MEMATTN", "1") in ("1", "2")
FUSED_MEMATTN_MAXD = 128 if os.environ.get("BMHRL_FUSED_MEMATTN", "1") == "2" else 1 << 30    # "2": the audio side only (dm = 128)
SMALL_ATTN = os.environ.get("BMHRL_SMALL_ATTN", "1") == "1"      # one-launch attention core for Sq, Sk <= 32 (A/B switch)
# P / delta / dS of the head-dimension-256 attentions with <= 256 keys in one launch (csrc/attention_bwd256.hip; A/B switch)
FUSED_SCORES_BWD = os.environ.get("BMHRL_FUSED_SCORES_BWD", "1") == "1"


def _use_small(dk, Sq, Sk, *lds_and_offs) -> bool:
    """the one-launch core of short sequences (csrc/small_attention.hip): caption self attention, goal attention"""
    return SMALL_ATTN and all(v % 8 == 0 for v in lds_and_offs) and ops.small_attention_ok(Sq, Sk, dk)


def _attn_core_fwd(Qb, q_off, ldq, Kb, k_off, ldk, Vb, v_off, ldv, mask, msb, msq, B, H, Sq, Sk, dk, p_drop, seed):
    """Returns (O bf16 [B*Sq, H*dk] with the reference's output dropout applied, stats) where stats is
    ('flash', row_max, row_sum) or ('mat', P bf16 [B,H,Sq,Skp])."""
    dev = Qb.device
    D = H * dk
    scale = 1.0 / math.sqrt(dk)
    O = torch.empty(B * Sq, D, dtype=_BF16, device=dev)
    if _use_small(dk, Sq, Sk, ldq, ldk, ldv, q_off, k_off, v_off):
        Skp = pad8(Sk)
        P = torch.empty(B, H, Sq, Skp, dtype=_BF16, device=dev)           # (the kernel writes the padding columns as zero)
        ops.small_attention_fwd(Qb, Kb, Vb, O, P, Skp, mask, msb, msq, B, H, Sq, Sk, dk, scale, ldq, ldk, ldv, D, q_off=q_off,
                                k_off=k_off, v_off=v_off, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
        return O, ("mat", P)
    if _use_flash(dk, Sq, Sk):
        rmax = torch.empty(B, H, Sq, device=dev)
        rsum = torch.empty(B, H, Sq, device=dev)
        ops.attention_fwd(Qb, Kb, Vb, O, rmax, rsum, mask, msb, msq, B, H, Sq, Sk, dk, scale, ldq, ldk, ldv, D,
                          q_off=q_off, k_off=k_off, v_off=v_off, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
        return O, ("flash", rmax, rsum)
    Skp = pad8(Sk)
    S = torch.empty(B, H, Sq, Sk, device=dev)
    ops.gemm(Qb, Kb, Sq, Sk, dk, lda=ldq, ldb=ldk, a_off=q_off, b_off=k_off, batch=(B, H), a_strides=(Sq * ldq, dk),
             b_strides=(Sk * ldk, dk), C_f32=S, ldc=Sk, c_strides=(H * Sq * Sk, Sq * Sk), alpha=scale, mask=mask,
             mask_sb1=msb, mask_sm=msq)
    P = _padded_bf16(B * H * Sq, Sk, dev).view(B, H, Sq, Skp)
    ops.softmax_rows(S, Sk, P, Skp, B * H * Sq, Sk)
    ops.gemm(P, Vb, Sq, dk, Sk, lda=Skp, ldb=ldv, b_off=v_off, b_trans=True, batch=(B, H),
             a_strides=(H * Sq * Skp, Sq * Skp), b_strides=(Sk * ldv, dk), C_bf16=O, ldcb=D, cb_strides=(Sq * D, dk),
             dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev, drop_strides=(Sq * D, dk, D))
    return O, ("mat", P)


def _padded_bf16(rows, cols, dev):
    """(rows, pad8(cols)) bf16 operand whose padding columns are zero and whose body the caller overwrites: no padding ->
    uninitialised memory; padded -> a pooled buffer of the step (zeroed once when created, kernels never write the padding)"""
    if cols % 8 == 0:
        return torch.empty(rows, cols, dtype=_BF16, device=dev)
    return SCRATCH.bf16(rows, cols, dev)


def _attn_core_bwd(dOb, Ob, stats, Qb, q_off, ldq, Kb, k_off, ldk, Vb, v_off, ldv, dQb, dq_off, lddq, dKb, dk_off, lddk,
                   dVb, dv_off, lddv, mask, msb, msq, B, H, Sq, Sk, dk, p_drop, db_q=None, db_k=None, db_v=None):
    """dOb: gradient w.r.t. the PRE-dropout attention output (bf16 [B*Sq, D]); Ob: saved post-dropout output.
    Writes bf16 dQ/dK/dV into column slices of the given buffers.  db_* = (zeroed fp32 tensor, offset): the column sums
    of dQ / dK / dV (bias gradients of the three projections) are accumulated there by the producing GEMM's epilogue."""
    if stats[0] == "mat" and _use_small(dk, Sq, Sk, ldq, ldk, ldv, lddq, lddk, lddv, q_off, k_off, v_off, dq_off, dk_off, dv_off):
        t = lambda d: (None, 0) if d is None else d
        (bq, oq), (bk, ok), (bv, ov) = t(db_q), t(db_k), t(db_v)
        if ops.deterministic() and (bq is not None or bk is not None or bv is not None):
            # the kernel's bias sums are one atomic per (sample, head) block and column: ordered column sums afterwards instead
            ops.small_attention_bwd(dOb, H * dk, stats[1], pad8(Sk), Qb, Kb, Vb, dQb, dKb, dVb, mask, msb, msq, B, H, Sq, Sk, dk,
                                    1.0 / math.sqrt(dk), ldq, ldk, ldv, lddq, lddk, lddv, q_off=q_off, k_off=k_off, v_off=v_off,
                                    dq_off=dq_off, dk_off=dk_off, dv_off=dv_off)
            for db, off, buf, ld, boff, rows in ((bq, oq, dQb, lddq, dq_off, B * Sq), (bk, ok, dKb, lddk, dk_off, B * Sk),
                                                  (bv, ov, dVb, lddv, dv_off, B * Sk)):
                if db is not None:
                    ops.colsum_bf16(buf, ld, db, True, rows, H * dk, dy_off=boff, db_off=off)
            return
        ops.small_attention_bwd(dOb, H * dk, stats[1], pad8(Sk), Qb, Kb, Vb, dQb, dKb, dVb, mask, msb, msq, B, H, Sq, Sk, dk,
                                1.0 / math.sqrt(dk), ldq, ldk, ldv, lddq, lddk, lddv, q_off=q_off, k_off=k_off, v_off=v_off,
                                dq_off=dq_off, dk_off=dk_off, dv_off=dv_off, dbq=bq, dbk=bk, dbv=bv, dbq_off=oq, dbk_off=ok,
                                dbv_off=ov)
        return
    csq = dict(colsum=db_q[0], colsum_off=db_q[1], colsum_sb2=dk) if db_q is not None else {}
    csk = dict(colsum=db_k[0], colsum_off=db_k[1], colsum_sb2=dk) if db_k is not None else {}
    csv = dict(colsum=db_v[0], colsum_off=db_v[1], colsum_sb2=dk) if db_v is not None else {}
    dev = dOb.device
    D = H * dk
    scale = 1.0 / math.sqrt(dk)
    Skp = pad8(Sk)
    pstr = (H * Sq * Skp, Sq * Skp)
    fused = (FUSED_SCORES_BWD and stats[0] == "flash" and ops.attention_bwd_scores256_ok(Sq, Sk, dk, msq)
             and all(v % 8 == 0 for v in (ldq, ldk, ldv, q_off, k_off, v_off)))
    if fused:
        # at most 256 keys: P, the row term and dS in one launch, the whole score row of a query on one lane
        P = torch.empty(B, H, Sq, Skp, dtype=_BF16, device=dev)
        dS = torch.empty(B, H, Sq, Skp, dtype=_BF16, device=dev)
        ops.attention_bwd_scores256(Qb, ldq, Kb, ldk, Vb, ldv, dOb, D, stats[1], stats[2], mask, msb, P, dS, Skp, B, H, Sq, Sk,
                                    scale, q_off=q_off, k_off=k_off, v_off=v_off)
        _attn_grad_products(P, dS, pstr, dOb, Qb, q_off, ldq, Kb, k_off, ldk, dQb, dq_off, lddq, dKb, dk_off, lddk, dVb, dv_off,
                            lddv, B, H, Sq, Sk, dk, Skp, csq, csk, csv)
        return
    delta = torch.empty(B, H, Sq, device=dev)
    # sum_d dO_pre * O_pre == (1-p) * sum_d dO_pre * O_post   (both carry the same keep mask / scale)
    ops.attn_delta(dOb, D, Ob, D, delta, B, H, Sq, dk, scale=1.0 - p_drop)
    if stats[0] == "flash":
        P = _padded_bf16(B * H * Sq, Sk, dev).view(B, H, Sq, Skp)
        ops.gemm(Qb, Kb, Sq, Sk, dk, lda=ldq, ldb=ldk, a_off=q_off, b_off=k_off, batch=(B, H), a_strides=(Sq * ldq, dk),
                 b_strides=(Sk * ldk, dk), C_bf16=P, ldcb=Skp, cb_strides=pstr, epilogue=ops.EPI_PROB, alpha=scale,
                 mask=mask, mask_sb1=msb, mask_sm=msq, rowvec=stats[1], rowvec2=stats[2], rv_strides=(H * Sq, Sq))
    else:
        P = stats[1]
    dS = _padded_bf16(B * H * Sq, Sk, dev).view(B, H, Sq, Skp)
    ops.gemm(dOb, Vb, Sq, Sk, dk, lda=D, ldb=ldv, b_off=v_off, batch=(B, H), a_strides=(Sq * D, dk),
             b_strides=(Sk * ldv, dk), C_bf16=dS, ldcb=Skp, cb_strides=pstr, epilogue=ops.EPI_DSCORE, alpha=scale,
             rowvec=delta, rv_strides=(H * Sq, Sq), aux=P, ldaux=Skp, aux_strides=pstr, mask=mask, mask_sb1=msb, mask_sm=msq)
    _attn_grad_products(P, dS, pstr, dOb, Qb, q_off, ldq, Kb, k_off, ldk, dQb, dq_off, lddq, dKb, dk_off, lddk, dVb, dv_off, lddv,
                        B, H, Sq, Sk, dk, Skp, csq, csk, csv)


def _attn_grad_products(P, dS, pstr, dOb, Qb, q_off, ldq, Kb, k_off, ldk, dQb, dq_off, lddq, dKb, dk_off, lddk, dVb, dv_off, lddv,
                        B, H, Sq, Sk, dk, Skp, csq, csk, csv):
    """dV = P^T dO ; dK = dS^T Q ; dQ = dS K (bf16 into column slices; cs*: fused column sums for the projection biases)"""
    D = H * dk
    ops.gemm(P, dOb, Sk, dk, Sq, lda=Skp, ldb=D, a_trans=True, b_trans=True, batch=(B, H), a_strides=pstr,
             b_strides=(Sq * D, dk), C_bf16=dVb, ldcb=lddv, cb_off=dv_off, cb_strides=(Sk * lddv, dk), **csv)
    ops.gemm(dS, Qb, Sk, dk, Sq, lda=Skp, ldb=ldq, b_off=q_off, a_trans=True, b_trans=True, batch=(B, H), a_strides=pstr,
             b_strides=(Sq * ldq, dk), C_bf16=dKb, ldcb=lddk, cb_off=dk_off, cb_strides=(Sk * lddk, dk), **csk)
    ops.gemm(dS, Kb, Sq, dk, Sk, lda=Skp, ldb=ldk, b_off=k_off, b_trans=True, batch=(B, H), a_strides=pstr,
             b_strides=(Sk * ldk, dk), C_bf16=dQb, ldcb=lddq, cb_off=dq_off, cb_strides=(Sq * lddq, dk), **csq)


# ------------------------------------------------------------------------------------------------ linear helpers
# K-split products on the ACTIVATION path (d x of the paired caption projections, d cat[x, goal] of the vocabulary head: few
# output tiles over a long reduction) sum their splits through a workspace in split order instead of fp32 atomics: the
# activation gradients -- and with them everything upstream -- no longer depend on the order the workgroups arrive in.  Leaf
# weight gradients keep the atomics (nothing reads them but the optimizer).  BMHRL_ORDERED_DX=0: atomics there too (A/B).
ORDERED_DX = os.environ.get("BMHRL_ORDERED_DX", "1") == "1"


def _split_ws(M, N, K, batch, device):
    """workspace for an ordered K split of an (M, N) fp32 product over K, or None when the product does not split"""
    if not ORDERED_DX:
        return None
    n = ops.gemm_splits(M, N, K, batch)
    return SCRATCH.f32(n * batch * M * N, device=device, zero=False) if n > 1 else None


def _linear_bwd(dyb, ldy, rows, N, xb, ldx, K, wb, *, need_dw, need_db, need_dx, dx_f32=None, dx_bf16=None, lddxb=0,
                dx_epilogue=ops.EPI_LINEAR, dx_alpha=1.0, dx_aux=None, ldaux=0, dx_drop=0.0, dx_seed=0,
                dy_off=0, x_off=0, w_off=0, dx_accumulate=False, dx_colsum=None, dx_split_k=False, leaf=None):
    """Gradients of y = x W^T + b given dy (bf16 [rows, N] at dy_off, leading dim ldy).
    dW (fp32 [N, K]) = dy^T x ; db = column sums of dy ; dx = dy W (fp32 and/or bf16, optional fused epilogue).
    leaf: a list -- the dW product is deferred into it (the caller launches the block's weight gradients together with
    ops.gemm_flush; see bmhrl_gemm_group)."""
    dev = dyb.device
    dw = db = None
    if need_dw:
        # zeroed when the long row reduction runs split-K with fp32 atomics; the large projections do not split
        dw = SCRATCH.f32(N, K, device=dev, zero=not ops.gemm_overwrites(N, K, rows))
        ops.gemm(dyb, xb, N, K, rows, lda=ldy, ldb=ldx, a_off=dy_off, b_off=x_off, a_trans=True, b_trans=True, C_f32=dw, ldc=K,
                 allow_split_k=True, defer=leaf)
    if need_db:
        db = SCRATCH.f32(N, device=dev)
        ops.colsum_bf16(dyb, ldy, db, True, rows, N, dy_off=dy_off)
    if need_dx:
        ws = _split_ws(rows, K, N, 1, dev) if dx_split_k else None
        ops.gemm(dyb, wb, rows, K, N, lda=ldy, ldb=wb.shape[1], a_off=dy_off, b_off=w_off, b_trans=True, C_f32=dx_f32,
                 ldc=K, C_bf16=dx_bf16, ldcb=lddxb, epilogue=dx_epilogue, alpha=dx_alpha, aux=dx_aux, ldaux=ldaux,
                 dropout_p=dx_drop, seed=dx_seed, seed_dev=SEEDS.dev, accumulate=dx_accumulate, colsum=dx_colsum,
                 allow_split_k=dx_split_k, split_ws=ws)   # (dx_split_k: few output tiles over a long reduction; dx_f32 is
                                                           #  zeroed for the case that the split runs on atomics)
    return dw, db


def _row_stride(t, cols):
    """leading dimension of `t` read as (rows, cols) fp32 rows, or None when it is not such a matrix (a column slice of a
    wider contiguous tensor -- d cat[x, goal] of the worker head -- is: no copy is needed to read it)"""
    if t.dim() < 2 or t.shape[-1] != cols or t.stride(-1) != 1:
        return None
    for i in range(t.dim() - 2):
        if t.shape[i] != 1 and t.stride(i) != t.stride(i + 1) * t.shape[i + 1]:
            return None
    return t.stride(-2) if t.stride(-2) >= cols else None


def _cast_dy(dy, rows, N, p_drop, seed, want_db, ld=None):
    """bf16 copy of an incoming fp32 gradient (through the layer's output dropout) and, when the bias needs one, its
    column sums from the same pass.  Returns (dyb, db or None).  ld: row stride of dy (default N)."""
    dev = dy.device
    ld = N if ld is None else ld
    dyb = SCRATCH.bf16(rows, N, dev)
    if want_db:
        db = SCRATCH.f32(N, device=dev)
        ops.cast_colsum_bf16(dy, ld, dyb, dyb.shape[1], rows, N, db, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
        return dyb, db
    ops.cast_bf16(dy, ld, dyb, dyb.shape[1], rows, N, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
    return dyb, None


class MHAFn(torch.autograd.Function):
    """[x +] dropout( d2Q( attention( Q2d(LN?(x)), K2d(kv), V2d(kv), mask ) ) )

    model/multihead_attention.py:60-92 wrapped (optionally) by model/blocks.py:135-144.  Self attention
    (kv_in is None): keys/values come from the same (normalised) tensor and Q/K/V run as one [3D, dq] GEMM.
    Cross attention: kv_in is the OTHER stream, un-normalised (model/bm_hrl_agent.py:362-364, 91-94).
    """

    @staticmethod
    def forward(ctx, x, kv_in, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, mask, H, p_drop, residual, kv_cache=None,
                emit_bf16=False):
        """emit_bf16: the output's bf16 copy leaves the last GEMM's epilogue too and is offered to the step's memo
        (StepScratch.offer_bf16): the consumer that takes this output as its attention memory finds it there"""
        dev = x.device
        B, Sq, dq = x.shape
        D = wq.shape[0]
        dk = D // H
        rows_q = B * Sq
        x = x.contiguous()
        self_att = kv_in is None
        has_ln = ln_w is not None
        ldx = pad8(dq)
        mean = rstd = None
        if has_ln:
            xb = SCRATCH.bf16(rows_q, dq, dev)
            mean = torch.empty(rows_q, device=dev)
            rstd = torch.empty(rows_q, device=dev)
            ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), xb, ldx, None, mean, rstd, rows_q, dq)
        else:
            xb = SCRATCH.memo_bf16(x, rows_q, dq)        # (its producer may have offered it: StepScratch.offer_bf16)
        m8, msb, msq = _mask_u8(mask)
        s_attn, s_res = SEEDS.next(), SEEDS.next()
        if self_att:
            Sk, rows_k, dkv = Sq, rows_q, dq
            w_qkv = SHADOWS.weight(wq, wk, wv)
            b_qkv = SHADOWS.bias(bq, bk, bv)
            QKV = torch.empty(rows_q, 3 * D, dtype=_BF16, device=dev)
            ops.gemm(xb, w_qkv, rows_q, 3 * D, dq, lda=ldx, ldb=w_qkv.shape[1], C_bf16=QKV, ldcb=3 * D, bias=b_qkv)
            Qb = Kb = Vb = QKV
            q_off, k_off, v_off, ldq, ldk = 0, D, 2 * D, 3 * D, 3 * D
            kvb = None
        else:
            _, Sk, dkv = kv_in.shape
            rows_k = B * Sk
            w_q = SHADOWS.weight(wq)
            Qb = torch.empty(rows_q, D, dtype=_BF16, device=dev)
            ops.gemm(xb, w_q, rows_q, D, dq, lda=ldx, ldb=w_q.shape[1], C_bf16=Qb, ldcb=D, bias=bq.detach())
            # decoding: the memory (encoder output) and these weights do not change between the tokens of a clip, so its
            # K|V projection is computed once per clip (`kv_cache`: a dict owned by the decoder, no-grad only)
            KV = kv_cache.get((id(wk), id(wv))) if kv_cache is not None else None
            kvb = None
            if KV is None:
                kv_in = kv_in.contiguous()
                kvb = SCRATCH.memo_bf16(kv_in, rows_k, dkv)      # (the producer may have offered it: offer_bf16)
                w_kv = SHADOWS.weight(wk, wv)
                KV = torch.empty(rows_k, 2 * D, dtype=_BF16, device=dev)
                ops.gemm(kvb, w_kv, rows_k, 2 * D, dkv, lda=kvb.shape[1], ldb=w_kv.shape[1], C_bf16=KV, ldcb=2 * D,
                         bias=SHADOWS.bias(bk, bv))
                if kv_cache is not None:
                    kv_cache[(id(wk), id(wv))] = KV
            Kb = Vb = KV
            q_off, k_off, v_off, ldq, ldk = 0, 0, D, D, 2 * D
        Ob, stats = _attn_core_fwd(Qb, q_off, ldq, Kb, k_off, ldk, Vb, v_off, ldk, m8, msb, msq, B, H, Sq, Sk, dk, p_drop, s_attn)
        w_o = SHADOWS.weight(wo)
        y = torch.empty(B, Sq, dq, device=dev)
        yb = SCRATCH.bf16(rows_q, dq, dev) if (emit_bf16 and SCRATCH.armed) else None
        ops.gemm(Ob, w_o, rows_q, dq, D, lda=D, ldb=w_o.shape[1], C_f32=y, ldc=dq, bias=bo.detach(),
                 residual=x if residual else None, ldr=dq, dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev,
                 C_bf16=yb, ldcb=yb.shape[1] if yb is not None else 0)
        if yb is not None:
            SCRATCH.offer_bf16(y, rows_q, dq, yb)
        ctx.save_for_backward(x, ln_w, mean, rstd, xb, kvb, Qb, Kb, Ob, m8, *stats[1:], wq, wk, wv, wo)
        ctx.cfg = (B, H, Sq, Sk, dq, dkv, D, dk, p_drop, residual, self_att, has_ln, stats[0], msb, msq, s_attn, s_res,
                   q_off, k_off, v_off, ldq, ldk)
        return y

    @staticmethod
    def backward(ctx, dy):
        (B, H, Sq, Sk, dq, dkv, D, dk, p_drop, residual, self_att, has_ln, kind, msb, msq, s_attn, s_res, q_off, k_off, v_off,
         ldq, ldk) = ctx.cfg
        saved = ctx.saved_tensors
        x, ln_w, mean, rstd, xb, kvb, Qb, Kb, Ob, m8 = saved[:10]
        n_stats = 2 if kind == "flash" else 1
        stats = (kind,) + tuple(saved[10:10 + n_stats])
        wq, wk, wv, wo = saved[10 + n_stats:]
        Vb = Kb
        dev = dy.device
        rows_q, rows_k = B * Sq, B * Sk
        ldx = pad8(dq)
        need = ctx.needs_input_grad
        ld_dy = None if dy.is_contiguous() or (residual and has_ln) else _row_stride(dy, dq)
        if ld_dy is None:
            dy = dy.contiguous()             # (a strided row view is read where it lies: the cast below takes its row stride)
        keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        # d(out) through the residual-branch dropout -> bf16
        dyb, dbo = _cast_dy(dy, rows_q, dq, p_drop, s_res, need[11], ld=ld_dy)
        # linear_d2Q backward; its dx is d(attention output), taken back through the output dropout in the epilogue
        dOb = torch.empty(rows_q, D, dtype=_BF16, device=dev)
        w_o = SHADOWS.weight(wo)
        # the weight gradients of a short block (caption side: 480 rows) leave as one grouped launch at the end; long ones
        # (the encoder's: 4096 / 12 800 rows, a kernel each anyway) stay next to the dY they read
        leaf = [] if max(rows_q, rows_k) < 2048 else None
        dwo, _ = _linear_bwd(dyb, ldx, rows_q, dq, Ob, D, D, w_o, need_dw=need[10], need_db=False, need_dx=True,
                             dx_bf16=dOb, lddxb=D, dx_drop=p_drop, dx_seed=s_attn, leaf=leaf)
        del keep
        if self_att:
            dQKV = torch.empty(rows_q, 3 * D, dtype=_BF16, device=dev)
            dQb = dKb = dVb = dQKV
            lddq = lddk = 3 * D
        else:
            dQb = torch.empty(rows_q, D, dtype=_BF16, device=dev)
            dKV = torch.empty(rows_k, 2 * D, dtype=_BF16, device=dev)
            dKb = dVb = dKV
            lddq, lddk = D, 2 * D
        # bias gradients of the Q / K / V projections = column sums of dQ / dK / dV: taken in the epilogues of the GEMMs
        # that produce them (layout [q | k | v] for self attention, [q], [k | v] for cross attention)
        need_bq, need_bkv = need[5], need[7] or need[9]
        if self_att:
            db_all = SCRATCH.f32(3 * D, device=dev) if (need_bq or need_bkv) else None
            tq = tk = tv = None
            if db_all is not None:
                tq, tk, tv = (db_all, 0), (db_all, D), (db_all, 2 * D)
        else:
            db_qq = SCRATCH.f32(D, device=dev) if need_bq else None
            db_kv = SCRATCH.f32(2 * D, device=dev) if need_bkv else None
            tq = (db_qq, 0) if need_bq else None
            tk = (db_kv, 0) if need_bkv else None
            tv = (db_kv, D) if need_bkv else None
        _attn_core_bwd(dOb, Ob, stats, Qb, q_off, ldq, Kb, k_off, ldk, Vb, v_off, ldk, dQb, q_off, lddq, dKb, k_off, lddk,
                       dVb, v_off, lddk, m8, msb, msq, B, H, Sq, Sk, dk, p_drop, db_q=tq, db_k=tk, db_v=tv)
        dxn = torch.empty(rows_q, dq, device=dev) if (need[0] or (has_ln and (need[2] or need[3]))) else None
        dwq = dbq = dwk = dbk = dwv = dbv = dkv_in = None
        if self_att:
            w_qkv = SHADOWS.weight(wq, wk, wv)
            need_w = need[4] or need[6] or need[8]
            dw, _ = _linear_bwd(dQKV, 3 * D, rows_q, 3 * D, xb, ldx, dq, w_qkv, need_dw=need_w, need_db=False,
                                need_dx=dxn is not None, dx_f32=dxn, leaf=leaf)
            db = db_all
            if dw is not None:
                dwq, dwk, dwv = dw[:D], dw[D:2 * D], dw[2 * D:]
            if db is not None:
                dbq, dbk, dbv = db[:D], db[D:2 * D], db[2 * D:]
        else:
            w_q = SHADOWS.weight(wq)
            w_kv = SHADOWS.weight(wk, wv)
            dwq, _ = _linear_bwd(dQb, D, rows_q, D, xb, ldx, dq, w_q, need_dw=need[4], need_db=False,
                                 need_dx=dxn is not None, dx_f32=dxn, leaf=leaf)
            dbq = db_qq
            if need[1]:
                dkv_in = torch.empty(B, Sk, dkv, device=dev)
            dw, _ = _linear_bwd(dKV, 2 * D, rows_k, 2 * D, kvb, kvb.shape[1], dkv, w_kv, need_dw=need[6] or need[8],
                                need_db=False, need_dx=need[1], dx_f32=dkv_in, leaf=leaf)
            db = db_kv
            if dw is not None:
                dwk, dwv = dw[:D], dw[D:]
            if db is not None:
                dbk, dbv = db[:D], db[D:]
        if leaf:
            ops.gemm_flush(leaf)
        dx = dlnw = dlnb = None
        if has_ln:
            if dxn is not None:
                dx = torch.empty(B, Sq, dq, device=dev)
                dlnw = SCRATCH.f32(dq, device=dev) if need[2] else None
                dlnb = SCRATCH.f32(dq, device=dev) if need[3] else None
                ops.layernorm_bwd(dxn, x, ln_w, mean, rstd, dx, dy if residual else None, dlnw, dlnb, rows_q, dq)
        elif need[0]:
            dx = dxn.view(B, Sq, dq)
            if residual:
                dx = dx + dy
        return (dx, dkv_in, dlnw, dlnb, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, None, None, None, None, None, None)


# fused (flash-style) backward of the head-dimension-128 attention; BMHRL_FUSED_ATTN_BWD=0 keeps the materialised P / dS GEMMs
# of r01 (A/B and parity tests of one form against the other)
FUSED_ATTN_BWD = os.environ.get("BMHRL_FUSED_ATTN_BWD", "1") != "0"


class MemAttnFn(torch.autograd.Function):
    """x + dropout( d2Q( attention( Q2d(LN(x)), K2d(mem), V2d(mem), mask ) ) ) for FEW queries against a LONG memory
    (the caption -> encoder-memory attentions of BMFusionLayer: 30 caption positions against 256 video / 800 audio
    positions; model/bm_hrl_agent.py:91-94 around model/multihead_attention.py:60-92).

    Same function as MHAFn's cross-attention branch, evaluated in an order that never projects the memory:
        scores_h = Q_h K_h^T = Q_h (mem Wk_h^T + bk_h)^T = (Q_h Wk_h) mem^T + (Q_h . bk_h) 1^T
    the last term is constant along the keys and cancels in the softmax (exactly; the -1e9 fill is unaffected), and
        P_h V_h = P_h (mem Wv_h^T + 1 bv_h^T) = (P_h mem) Wv_h^T + bv_h            (rows of P sum to 1).
    So the keys / values of the memory are the memory itself, and the two d_model-wide projections move to the query
    side: per layer 2 x 480 rows instead of 2 x 12 800 (audio) / 4 096 (video) rows -- the K|V projection GEMMs, their
    weight- and input-gradient GEMMs and the memory casts (about a fifth of the step's FLOPs at config 2) disappear.
    bk gets an exactly-zero gradient (the reference's is fp32 noise around zero).  All tensors of the attention proper
    are laid out (B, L, H, .) so that sums over heads are plain GEMMs with K = L*H."""

    @staticmethod
    def forward(ctx, x, mem, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, mask, H, p_drop, emit_bf16=False):
        # mem is None: self attention -- the memory is LN(x) itself (keys / values of the reference's self attention
        # are projections of the normalised input), and its gradient joins the one of the query path
        dev = x.device
        B, L, dq = x.shape
        self_att = mem is None
        Sk, dm = (L, dq) if self_att else mem.shape[1:]
        D = wq.shape[0]
        dk = D // H
        rows = B * L
        x = x.contiguous()
        ldx, dmp, Skp = pad8(dq), pad8(dm), pad8(Sk)
        scale = 1.0 / math.sqrt(dk)
        xb = SCRATCH.bf16(rows, dq, dev)
        mean = torch.empty(rows, device=dev)
        rstd = torch.empty(rows, device=dev)
        ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), xb, ldx, None, mean, rstd, rows, dq)
        s_attn, s_res = SEEDS.next(), SEEDS.next()
        w_q, w_k, w_v, w_o = SHADOWS.weight(wq), SHADOWS.weight(wk), SHADOWS.weight(wv), SHADOWS.weight(wo)
        Qb = torch.empty(rows, D, dtype=_BF16, device=dev)
        ops.gemm(xb, w_q, rows, D, dq, lda=ldx, ldb=w_q.shape[1], C_bf16=Qb, ldcb=D, bias=bq.detach())
        # bf16 memory rows: LN(x) for self attention, else a copy of the memory shared by every layer of a step
        memb = xb if self_att else SCRATCH.memo_bf16(mem, B * Sk, dm)
        zeros = torch.zeros if dmp != dm else torch.empty           # per-head padding columns must be zero (operands)
        # Q'_h = Q_h Wk_h : (rows, dk) x (dk, dm)
        Qp = zeros(rows, H * dmp, dtype=_BF16, device=dev)
        ops.gemm(Qb, w_k, rows, dm, dk, lda=D, ldb=w_k.shape[1], b_trans=True, batch=(1, H), a_strides=(0, dk),
                 b_strides=(0, dk * w_k.shape[1]), C_bf16=Qp, ldcb=H * dmp, cb_strides=(0, dmp))
        m8, msb, msq = _mask_u8(mask)
        Cx = zeros(rows, H * dmp, dtype=_BF16, device=dev)
        flash = dm == 128 and L >= 128 and msq == 0 and Sk <= ops.attention_max_keys()
        if flash:
            # many queries against the 128-wide (audio) rows: fused kernel, one key / value tile for all heads; the
            # probabilities are recomputed in backward from the softmax statistics
            rmax = torch.empty(B, H, L, device=dev)
            rsum = torch.empty(B, H, L, device=dev)
            ops.attention_shared128_fwd(Qp, memb, Cx, rmax, rsum, m8, msb, B, H, L, Sk, scale, H * dmp, dmp, H * dmp)
            stats = (rmax, rsum)
        else:
            # scores (B, L, H, Sk) = scale * Q'_h mem^T, masked
            S = torch.empty(B, L, H, Skp, device=dev)
            ops.gemm(Qp, memb, L, Sk, dm, lda=H * dmp, ldb=dmp, batch=(B, H), a_strides=(L * H * dmp, dmp),
                     b_strides=(Sk * dmp, 0), C_f32=S, ldc=H * Skp, c_strides=(L * H * Skp, Skp), alpha=scale, mask=m8,
                     mask_sb1=msb, mask_sm=msq)
            P = _padded_bf16(B * L * H, Sk, dev).view(B, L, H, Skp)
            ops.softmax_rows(S, Skp, P, Skp, B * L * H, Sk)
            # context in memory space (B, L, H, dm) = P_h mem
            ops.gemm(P, memb, L, dm, Sk, lda=H * Skp, ldb=dmp, b_trans=True, batch=(B, H), a_strides=(L * H * Skp, Skp),
                     b_strides=(Sk * dmp, 0), C_bf16=Cx, ldcb=H * dmp, cb_strides=(L * H * dmp, dmp))
            stats = (P,)
        # O_h = dropout(Cx_h Wv_h^T + bv_h)
        Ob = torch.empty(rows, D, dtype=_BF16, device=dev)
        ops.gemm(Cx, w_v, rows, dk, dm, lda=H * dmp, ldb=w_v.shape[1], batch=(1, H), a_strides=(0, dmp),
                 b_strides=(0, dk * w_v.shape[1]), C_bf16=Ob, ldcb=D, cb_strides=(0, dk), bias=bv.detach(), bias_sb2=dk,
                 dropout_p=p_drop, seed=s_attn, seed_dev=SEEDS.dev, drop_strides=(0, dk, D))
        y = torch.empty(B, L, dq, device=dev)
        yb = SCRATCH.bf16(rows, dq, dev) if (emit_bf16 and SCRATCH.armed) else None      # (see MHAFn.forward)
        ops.gemm(Ob, w_o, rows, dq, D, lda=D, ldb=w_o.shape[1], C_f32=y, ldc=dq, bias=bo.detach(), residual=x, ldr=dq,
                 dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev, C_bf16=yb, ldcb=yb.shape[1] if yb is not None else 0)
        if yb is not None:
            SCRATCH.offer_bf16(y, rows, dq, yb)
        ctx.save_for_backward(x, ln_w, mean, rstd, xb, memb, Qb, Qp, Cx, Ob, wq, wk, wv, wo, m8, *stats)
        ctx.cfg = (B, L, Sk, dq, dm, D, H, dk, p_drop, s_attn, s_res, self_att, flash, msb, msq)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, L, Sk, dq, dm, D, H, dk, p_drop, s_attn, s_res, self_att, flash, msb, msq = ctx.cfg
        x, ln_w, mean, rstd, xb, memb, Qb, Qp, Cx, Ob, wq, wk, wv, wo, m8 = ctx.saved_tensors[:15]
        stats = ctx.saved_tensors[15:]
        dev = dy.device
        rows = B * L
        ldx, dmp, Skp = pad8(dq), pad8(dm), pad8(Sk)
        scale = 1.0 / math.sqrt(dk)
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        w_q, w_k, w_v, w_o = SHADOWS.weight(wq), SHADOWS.weight(wk), SHADOWS.weight(wv), SHADOWS.weight(wo)
        zeros = torch.zeros if dmp != dm else torch.empty
        # out projection: dWo, dbo; d(attention output) back through its dropout, with the column sums = dbv
        dyb, dbo = _cast_dy(dy, rows, dq, p_drop, s_res, need[11])
        dOb = torch.empty(rows, D, dtype=_BF16, device=dev)
        dbv = SCRATCH.f32(D, device=dev) if need[9] else None
        dwo, _ = _linear_bwd(dyb, ldx, rows, dq, Ob, D, D, w_o, need_dw=need[10], need_db=False, need_dx=True,
                             dx_bf16=dOb, lddxb=D, dx_drop=p_drop, dx_seed=s_attn, dx_colsum=dbv)
        # O_h = Cx_h Wv_h^T + bv_h
        dwv = None
        if need[8]:
            dwv = SCRATCH.f32(D, dm, device=dev, zero=not ops.gemm_overwrites(dk, dm, rows, H))
            ops.gemm(dOb, Cx, dk, dm, rows, lda=D, ldb=H * dmp, a_trans=True, b_trans=True, batch=(1, H), a_strides=(0, dk),
                     b_strides=(0, dmp), C_f32=dwv, ldc=dm, c_strides=(0, dk * dm), allow_split_k=True)
        dCx = zeros(rows, H * dmp, dtype=_BF16, device=dev)
        ops.gemm(dOb, w_v, rows, dm, dk, lda=D, ldb=w_v.shape[1], b_trans=True, batch=(1, H), a_strides=(0, dk),
                 b_strides=(0, dk * w_v.shape[1]), C_bf16=dCx, ldcb=H * dmp, cb_strides=(0, dmp))
        # softmax backward: dS = P (dP - delta) * scale with dP = dCx_h mem^T and delta = sum_k P dP.  The fused kernels take
        # delta = sum_n dCx Cx (the same number through the context); the materialised few-query path forms it from P and dP
        # themselves (ops.softmax_bwd_rows), which keeps the rows of dS summing to zero under bf16 rounding
        delta = None
        if flash:
            delta = torch.empty(B, H, L, device=dev)
            ops.attn_delta(dCx, H * dmp, Cx, H * dmp, delta, B, H, L, dmp)
        dQp = zeros(rows, H * dmp, dtype=_BF16, device=dev)
        dmem = None
        dxn = None
        if flash and FUSED_ATTN_BWD:
            # fused: P and dS are recomputed per tile from the forward's statistics and stay on chip; one call yields
            # dQ'_h = dS_h mem and d(mem) = sum_h P_h^T dCx_h + dS_h^T Q'_h (bmhrl_attention_shared128_bwd)
            if self_att:
                dxn = torch.empty(rows, dq, device=dev)       # d(LN(x)) starts as the key / value side's gradient ...
                target = dxn
            elif need[1]:
                dmem = torch.empty(B, Sk, dm, device=dev)
                target = dmem
            else:
                target = None
            ops.attention_shared128_bwd(Qp, memb, dCx, stats[0], stats[1], delta, m8, msb, dQp, target, False, B, H, L, Sk,
                                        scale, H * dmp, dmp, H * dmp, H * dmp, lddx=dm)
        else:
            dS = _padded_bf16(B * L * H, Sk, dev).view(B, L, H, Skp)
            pstr = (L * H * Skp, Skp)
            if flash:      # P (B, L, H, Sk) recomputed from the statistics of the fused forward
                P = _padded_bf16(B * L * H, Sk, dev).view(B, L, H, Skp)
                ops.gemm(Qp, memb, L, Sk, dm, lda=H * dmp, ldb=dmp, batch=(B, H), a_strides=(L * H * dmp, dmp),
                         b_strides=(Sk * dmp, 0), C_bf16=P, ldcb=H * Skp, cb_strides=pstr, epilogue=ops.EPI_PROB, alpha=scale,
                         mask=m8, mask_sb1=msb, mask_sm=0, rowvec=stats[0], rowvec2=stats[1], rv_strides=(H * L, L))
                ops.gemm(dCx, memb, L, Sk, dm, lda=H * dmp, ldb=dmp, batch=(B, H), a_strides=(L * H * dmp, dmp),
                         b_strides=(Sk * dmp, 0), C_bf16=dS, ldcb=H * Skp, cb_strides=pstr, epilogue=ops.EPI_DSCORE, alpha=scale,
                         rowvec=delta, rv_strides=(H * L, L), aux=P, ldaux=H * Skp, aux_strides=pstr, mask=m8, mask_sb1=msb)
            else:
                P = stats[0]
                dP = torch.empty(B, L, H, Skp, device=dev)
                ops.gemm(dCx, memb, L, Sk, dm, lda=H * dmp, ldb=dmp, batch=(B, H), a_strides=(L * H * dmp, dmp),
                         b_strides=(Sk * dmp, 0), C_f32=dP, ldc=H * Skp, c_strides=pstr)
                ops.softmax_bwd_rows(P, Skp, dP, Skp, dS, Skp, B * L * H, Sk, scale, m8, msb, msq, H, L)
            # d(mem)[b] = sum_h P_h^T dCx_h + dS_h^T Q'_h : two GEMMs with K = L*H (rows (l, h) of the (B, L, H, .) tensors)
            def grad_mem(target, first_accumulates):
                ops.gemm(P, dCx, Sk, dm, L * H, lda=Skp, ldb=dmp, a_trans=True, b_trans=True, batch=(B, 1),
                         a_strides=(L * H * Skp, 0), b_strides=(L * H * dmp, 0), C_f32=target, ldc=dm, c_strides=(Sk * dm, 0),
                         accumulate=first_accumulates)
                ops.gemm(dS, Qp, Sk, dm, L * H, lda=Skp, ldb=dmp, a_trans=True, b_trans=True, batch=(B, 1),
                         a_strides=(L * H * Skp, 0), b_strides=(L * H * dmp, 0), C_f32=target, ldc=dm, c_strides=(Sk * dm, 0),
                         accumulate=True)
            if need[1] and not self_att:
                dmem = torch.empty(B, Sk, dm, device=dev)
                grad_mem(dmem, False)
            # scores = scale * Q'_h mem^T (scale is already inside dS)
            ops.gemm(dS, memb, L, dm, Sk, lda=H * Skp, ldb=dmp, b_trans=True, batch=(B, H), a_strides=pstr, b_strides=(Sk * dmp, 0),
                     C_bf16=dQp, ldcb=H * dmp, cb_strides=(L * H * dmp, dmp))
        # Q'_h = Q_h Wk_h
        dwk = None
        if need[6]:
            dwk = SCRATCH.f32(D, dm, device=dev, zero=not ops.gemm_overwrites(dk, dm, rows, H))
            ops.gemm(Qb, dQp, dk, dm, rows, lda=D, ldb=H * dmp, a_trans=True, b_trans=True, batch=(1, H), a_strides=(0, dk),
                     b_strides=(0, dmp), C_f32=dwk, ldc=dm, c_strides=(0, dk * dm), allow_split_k=True)
        dbq = SCRATCH.f32(D, device=dev) if need[5] else None
        dQb = torch.empty(rows, D, dtype=_BF16, device=dev)
        ops.gemm(dQp, w_k, rows, dk, dm, lda=H * dmp, ldb=w_k.shape[1], batch=(1, H), a_strides=(0, dmp),
                 b_strides=(0, dk * w_k.shape[1]), C_bf16=dQb, ldcb=D, cb_strides=(0, dk), colsum=dbq, colsum_sb2=dk)
        dbk = SCRATCH.f32(D, device=dev) if need[7] else None       # exactly zero: a shift of all keys' scores
        # Q projection and LayerNorm
        fused_self = dxn is not None           # ... and the query path's gradient is added by the GEMM below
        if dxn is None:
            dxn = torch.empty(rows, dq, device=dev)
        dwq, _ = _linear_bwd(dQb, D, rows, D, xb, ldx, dq, w_q, need_dw=need[4], need_db=False, need_dx=True, dx_f32=dxn,
                             dx_accumulate=fused_self)
        if self_att and not fused_self:
            grad_mem(dxn, True)        # the keys / values are LN(x) too: their gradient joins the query path's
        dx = torch.empty(B, L, dq, device=dev)
        dlnw = SCRATCH.f32(dq, device=dev) if need[2] else None
        dlnb = SCRATCH.f32(dq, device=dev) if need[3] else None
        ops.layernorm_bwd(dxn, x, ln_w, mean, rstd, dx, dy, dlnw, dlnb, rows, dq)
        return (dx, dmem, dlnw, dlnb, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, None, None, None, None)


class PairMemAttnFn(torch.autograd.Function):
    """MemAttnFn for TWO attention modules of the same shape at once: the worker and the manager fusion stacks
    (model/bm_hrl_agent.py:523,528) run identical layers on different weights over the same encoder memory, and at 30 caption
    positions every one of their ~50 launches per attention is bound by latency, not by work.  x2 is (2, B, L, dq), the
    weights of the two modules are used through stacked bf16 shadows ([w_a; w_b]), and each product runs as ONE launch with the
    pair as a GEMM batch dimension (weights: batch stride N * ld; activations: B * L rows) or, where no weight is involved
    (scores, probabilities, context, softmax), as a batch of 2 B samples.  The memory is the same for both (for a self
    attention it is each half's own LN(x)); its bf16 copy is laid out twice so that sample index 2 B addresses it.
    Arithmetic per element is the one of MemAttnFn (same kernels, same epilogues, same dropout element ids per half up to the
    half's offset).  Argument order: x2, mem, mask, H, p_drop, then the ten parameters (ln_w, ln_b, wq, bq, wk, bk, wv, bv,
    wo, bo) of module a, then of module b."""

    @staticmethod
    def forward(ctx, x2, mem, mask, H, p_drop, *params):
        pa, pb = params[:10], params[10:]
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = zip(pa, pb)
        dev = x2.device
        _, B, L, dq = x2.shape
        self_att = mem is None
        Sk, dm = (L, dq) if self_att else mem.shape[1:]
        D = wq[0].shape[0]
        dk = D // H
        R, B2 = B * L, 2 * B
        x2 = x2.contiguous()
        ldx, dmp, Skp = pad8(dq), pad8(dm), pad8(Sk)
        scale = 1.0 / math.sqrt(dk)
        xb = SCRATCH.bf16(2 * R, dq, dev)
        mean = torch.empty(2 * R, device=dev)
        rstd = torch.empty(2 * R, device=dev)
        ops.layernorm_fwd_groups(x2, SHADOWS.bias(*ln_w), SHADOWS.bias(*ln_b), xb, ldx, None, mean, rstd, R, dq, 2)
        s_attn, s_res = SEEDS.next(), SEEDS.next()
        w_q, w_k, w_v, w_o = SHADOWS.weight(*wq), SHADOWS.weight(*wk), SHADOWS.weight(*wv), SHADOWS.weight(*wo)
        Qb = torch.empty(2 * R, D, dtype=_BF16, device=dev)
        ops.gemm(xb, w_q, R, D, dq, lda=ldx, ldb=w_q.shape[1], batch=(1, 2), a_strides=(0, R * ldx), b_strides=(0, D * w_q.shape[1]),
                 C_bf16=Qb, ldcb=D, cb_strides=(0, R * D), bias=SHADOWS.bias(*bq), bias_sb2=D)
        # score -> softmax -> context (and dP -> dS -> dQ' in the backward) as ONE launch per direction where the shape allows
        # (csrc/memory_attention.hip: 30 queries against 256 x 1024 / 800 x 128 memories)
        fused = FUSED_MEMATTN and not self_att and dmp == dm and dm <= FUSED_MEMATTN_MAXD and ops.memory_attention_ok(L, Sk, dm)
        memT, ldt = None, 0
        if fused:
            memb, memT, ldt = SCRATCH.memo_mem(mem, B, Sk, dm)
        else:
            memb = xb if self_att else SCRATCH.memo_bf16(mem, B * Sk, dm, copies=2)
        zeros = torch.zeros if dmp != dm else torch.empty
        # Q' and (in the backward) dCx share one buffer, P and dS another, interleaved per query: row (sample, query) holds
        # [slot 0: H heads | slot 1: H heads].  Slot 0 = P / dCx, slot 1 = dS / Q' -- so that d(mem) = sum_h P_h^T dCx_h +
        # dS_h^T Q'_h of a sample is ONE product over the 2 L H rows of the sample (one pass over the fp32 d(mem) instead of a
        # write, a read-modify-write and a sum of the two stacks)
        QD = zeros(2 * R, 2 * H * dmp, dtype=_BF16, device=dev)
        ldqd, ldpd = 2 * H * dmp, 2 * H * Skp
        ldk = w_k.shape[1]
        ops.gemm(Qb, w_k, R, dm, dk, lda=D, ldb=ldk, b_trans=True, batch=(2, H), a_strides=(R * D, dk),
                 b_strides=(D * ldk, dk * ldk), C_bf16=QD, ldcb=ldqd, cb_off=H * dmp, cb_strides=(R * ldqd, dmp))
        m8, msb, msq = _mask_u8(mask)            # the caller passes the mask of 2 B samples
        assert m8 is None or m8.shape[0] == B2
        Cx = zeros(2 * R, H * dmp, dtype=_BF16, device=dev)
        PD = _padded_bf16(B2 * L * 2 * H, Sk, dev)                  # (B2, L, 2, H, Skp)
        fused = fused and (m8 is None or msq == 0)
        if fused:
            ops.memory_attention(False, QD, H * dmp, ldqd, memb, memT, ldt, PD, ldpd, H * Skp, Cx, H * dmp, m8, msb, B, B2, H, L, Sk,
                                 dm, scale)
        else:
            S = torch.empty(B2, L, H, Skp, device=dev)
            ops.gemm(QD, memb, L, Sk, dm, lda=ldqd, ldb=dmp, a_off=H * dmp, batch=(B2, H), a_strides=(L * ldqd, dmp),
                     b_strides=(Sk * dmp, 0), C_f32=S, ldc=H * Skp, c_strides=(L * H * Skp, Skp), alpha=scale, mask=m8, mask_sb1=msb,
                     mask_sm=msq)
            ops.softmax_rows(S, Skp, PD, Skp, B2 * L * H, Sk, rows_per_group=H, group_stride=ldpd)
            ops.gemm(PD, memb, L, dm, Sk, lda=ldpd, ldb=dmp, b_trans=True, batch=(B2, H), a_strides=(L * ldpd, Skp),
                     b_strides=(Sk * dmp, 0), C_bf16=Cx, ldcb=H * dmp, cb_strides=(L * H * dmp, dmp))
        Ob = torch.empty(2 * R, D, dtype=_BF16, device=dev)
        ldv = w_v.shape[1]
        ops.gemm(Cx, w_v, R, dk, dm, lda=H * dmp, ldb=ldv, batch=(2, H), a_strides=(R * H * dmp, dmp), b_strides=(D * ldv, dk * ldv),
                 C_bf16=Ob, ldcb=D, cb_strides=(R * D, dk), bias=SHADOWS.bias(*bv), bias_sb1=D, bias_sb2=dk,
                 dropout_p=p_drop, seed=s_attn, seed_dev=SEEDS.dev, drop_strides=(R * D, dk, D))
        y = torch.empty(2, B, L, dq, device=dev)
        ops.gemm(Ob, w_o, R, dq, D, lda=D, ldb=w_o.shape[1], batch=(1, 2), a_strides=(0, R * D), b_strides=(0, dq * w_o.shape[1]),
                 C_f32=y, ldc=dq, c_strides=(0, R * dq), bias=SHADOWS.bias(*bo), bias_sb2=dq, residual=x2, ldr=dq,
                 r_strides=(0, R * dq), dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev, drop_strides=(0, R * dq, dq))
        ctx.save_for_backward(x2, mean, rstd, xb, memb, Qb, QD, Cx, Ob, m8, PD, memT, *params)
        ctx.cfg = (B, L, Sk, dq, dm, D, H, dk, p_drop, s_attn, s_res, self_att, msb, msq, fused, ldt)
        return y

    @staticmethod
    def backward(ctx, dy2):
        B, L, Sk, dq, dm, D, H, dk, p_drop, s_attn, s_res, self_att, msb, msq, fused, ldt = ctx.cfg
        x2, mean, rstd, xb, memb, Qb, QD, Cx, Ob, m8, PD, memT = ctx.saved_tensors[:12]
        params = ctx.saved_tensors[12:]
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = zip(params[:10], params[10:])
        dev = dy2.device
        R, B2 = B * L, 2 * B
        ldx, dmp, Skp = pad8(dq), pad8(dm), pad8(Sk)
        scale = 1.0 / math.sqrt(dk)
        need = ctx.needs_input_grad
        dy2 = dy2.contiguous()
        w_q, w_k, w_v, w_o = SHADOWS.weight(*wq), SHADOWS.weight(*wk), SHADOWS.weight(*wv), SHADOWS.weight(*wo)
        ldk, ldv = w_k.shape[1], w_v.shape[1]
        zeros = torch.zeros if dmp != dm else torch.empty
        # out projection: dWo, dbo (column sums of the cast, one vector per half); d(attention output) through its dropout
        dyb = SCRATCH.bf16(2 * R, dq, dev)
        dbo = SCRATCH.f32(2 * dq, device=dev)
        ops.cast_colsum_bf16(dy2, dq, dyb, ldx, 2 * R, dq, dbo, dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev, group_rows=R,
                             colsum_stride=dq)
        leaf = []                                       # weight-gradient products: one grouped launch at the end (ops.gemm_flush)
        dwo = SCRATCH.f32(2 * dq, D, device=dev)
        ops.gemm(dyb, Ob, dq, D, R, lda=ldx, ldb=D, a_trans=True, b_trans=True, batch=(1, 2), a_strides=(0, R * ldx),
                 b_strides=(0, R * D), C_f32=dwo, ldc=D, c_strides=(0, dq * D), allow_split_k=True, defer=leaf)
        dOb = torch.empty(2 * R, D, dtype=_BF16, device=dev)
        dbv = SCRATCH.f32(2 * D, device=dev)
        ops.gemm(dyb, w_o, R, D, dq, lda=ldx, ldb=w_o.shape[1], b_trans=True, batch=(1, 2), a_strides=(0, R * ldx),
                 b_strides=(0, dq * w_o.shape[1]), C_bf16=dOb, ldcb=D, cb_strides=(0, R * D), dropout_p=p_drop, seed=s_attn,
                 seed_dev=SEEDS.dev, drop_strides=(0, R * D, D), colsum=dbv, colsum_sb2=D)
        # O_h = Cx_h Wv_h^T + bv_h
        dwv = SCRATCH.f32(2 * D, dm, device=dev, zero=not ops.gemm_overwrites(dk, dm, R, 2 * H))
        ops.gemm(dOb, Cx, dk, dm, R, lda=D, ldb=H * dmp, a_trans=True, b_trans=True, batch=(2, H), a_strides=(R * D, dk),
                 b_strides=(R * H * dmp, dmp), C_f32=dwv, ldc=dm, c_strides=(D * dm, dk * dm), allow_split_k=True, defer=leaf)
        ldqd, ldpd = 2 * H * dmp, 2 * H * Skp       # (see forward: slot 0 = P / dCx, slot 1 = dS / Q')
        ops.gemm(dOb, w_v, R, dm, dk, lda=D, ldb=ldv, b_trans=True, batch=(2, H), a_strides=(R * D, dk), b_strides=(D * ldv, dk * ldv),
                 C_bf16=QD, ldcb=ldqd, cb_strides=(R * ldqd, dmp))
        # dS = scale * P (dP - sum_k P dP), the row term from P and dP themselves (see MemAttnFn.backward)
        dQp = zeros(2 * R, H * dmp, dtype=_BF16, device=dev)
        if fused:       # dP -> dS -> dQ' in one launch (dS also lands in its slot of PD for d(mem) below)
            ops.memory_attention(True, QD, 0, ldqd, memb, memT, ldt, PD, ldpd, H * Skp, dQp, H * dmp, m8, msb, B, B2, H, L, Sk, dm,
                                 scale)
        else:
            dP = torch.empty(B2, L, H, Skp, device=dev)
            ops.gemm(QD, memb, L, Sk, dm, lda=ldqd, ldb=dmp, batch=(B2, H), a_strides=(L * ldqd, dmp), b_strides=(Sk * dmp, 0),
                     C_f32=dP, ldc=H * Skp, c_strides=(L * H * Skp, Skp))
            ops.softmax_bwd_rows(PD, Skp, dP, Skp, PD, Skp, B2 * L * H, Sk, scale, m8, msb, msq, H, L, rows_per_group=H,
                                 group_stride=ldpd, ds_off=H * Skp)

        def grad_mem(target, c_off, c_sb, accumulate):
            """d(mem)[b] = sum_h P_h^T dCx_h + dS_h^T Q'_h: one product over the sample's 2 L H rows, stack by stack (the
            second stack's launch adds to what the first wrote when both read the same memory)"""
            for half in range(2):
                ops.gemm(PD, QD, Sk, dm, 2 * L * H, lda=Skp, ldb=dmp, a_off=half * B * L * ldpd, b_off=half * B * L * ldqd,
                         a_trans=True, b_trans=True, batch=(B, 1), a_strides=(L * ldpd, 0), b_strides=(L * ldqd, 0), C_f32=target,
                         ldc=dm, c_off=c_off[half], c_strides=(c_sb, 0), accumulate=accumulate[half])
        dmem = None
        if need[1] and not self_att:
            dmem = torch.empty(B, Sk, dm, device=dev)   # both halves read the same memory
            grad_mem(dmem, (0, 0), Sk * dm, (False, True))
        if not fused:
            ops.gemm(PD, memb, L, dm, Sk, lda=ldpd, ldb=dmp, a_off=H * Skp, b_trans=True, batch=(B2, H), a_strides=(L * ldpd, Skp),
                     b_strides=(Sk * dmp, 0), C_bf16=dQp, ldcb=H * dmp, cb_strides=(L * H * dmp, dmp))
        # Q'_h = Q_h Wk_h
        dwk = SCRATCH.f32(2 * D, dm, device=dev, zero=not ops.gemm_overwrites(dk, dm, R, 2 * H))
        ops.gemm(Qb, dQp, dk, dm, R, lda=D, ldb=H * dmp, a_trans=True, b_trans=True, batch=(2, H), a_strides=(R * D, dk),
                 b_strides=(R * H * dmp, dmp), C_f32=dwk, ldc=dm, c_strides=(D * dm, dk * dm), allow_split_k=True, defer=leaf)
        dbq = SCRATCH.f32(2 * D, device=dev)
        dQb = torch.empty(2 * R, D, dtype=_BF16, device=dev)
        ops.gemm(dQp, w_k, R, dk, dm, lda=H * dmp, ldb=ldk, batch=(2, H), a_strides=(R * H * dmp, dmp), b_strides=(D * ldk, dk * ldk),
                 C_bf16=dQb, ldcb=D, cb_strides=(R * D, dk), colsum=dbq, colsum_sb1=D, colsum_sb2=dk)
        dbk = SCRATCH.f32(2 * D, device=dev)            # exactly zero: a shift of all keys' scores
        # Q projection and LayerNorm
        dwq = SCRATCH.f32(2 * D, dq, device=dev)
        ops.gemm(dQb, xb, D, dq, R, lda=D, ldb=ldx, a_trans=True, b_trans=True, batch=(1, 2), a_strides=(0, R * D),
                 b_strides=(0, R * ldx), C_f32=dwq, ldc=dq, c_strides=(0, D * dq), allow_split_k=True, defer=leaf)
        ops.gemm_flush(leaf)
        dxn = SCRATCH.f32(2 * R, dq, device=dev)         # (zeroed: split K, as in PairSelfAttnFn)
        ops.gemm(dQb, w_q, R, dq, D, lda=D, ldb=w_q.shape[1], b_trans=True, batch=(1, 2), a_strides=(0, R * D),
                 b_strides=(0, D * w_q.shape[1]), C_f32=dxn, ldc=dq, c_strides=(0, R * dq), allow_split_k=True,
                 split_ws=_split_ws(R, dq, D, 2, dev))
        if self_att:                                    # the keys / values are LN(x) too: each half its own
            grad_mem(dxn, (0, R * dq), Sk * dm, (True, True))
        dx2 = torch.empty(2, B, L, dq, device=dev)
        dlnw2, dlnb2 = SCRATCH.f32(2 * dq, device=dev), SCRATCH.f32(2 * dq, device=dev)
        ops.layernorm_bwd_groups(dxn, x2, SHADOWS.bias(*ln_w), mean, rstd, dx2, dy2, dlnw2, dlnb2, R, dq, 2)
        dln = [(dlnw2[i * dq:(i + 1) * dq], dlnb2[i * dq:(i + 1) * dq]) for i in range(2)]
        out = []
        for i in range(2):
            g = (dln[i][0], dln[i][1], dwq[i * D:(i + 1) * D], dbq[i * D:(i + 1) * D], dwk[i * D:(i + 1) * D], dbk[i * D:(i + 1) * D],
                 dwv[i * D:(i + 1) * D], dbv[i * D:(i + 1) * D], dwo[i * dq:(i + 1) * dq], dbo[i * dq:(i + 1) * dq])
            out += [gi if need[5 + 10 * i + j] else None for j, gi in enumerate(g)]
        return (dx2, dmem, None, None, None, *out)


class PairSelfAttnFn(torch.autograd.Function):
    """MHAFn's self-attention branch (x + dropout(d2Q(attention(Q2d, K2d, V2d of LN(x))))) for the two fusion stacks at once:
    the projections run with the pair as a GEMM batch dimension over stacked weights ([q_a; k_a; v_a; q_b; k_b; v_b]), the
    attention core over 2 B samples.  Same kernels and the same projected form as the single-stack call (the caption self
    attention has as many keys as queries: absorbing the projections would only add launches).  Arguments: x2 (2, B, L, dq),
    mask of 2 B samples, H, p_drop, then (ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo) of module a and of module b."""

    @staticmethod
    def forward(ctx, x2, mask, H, p_drop, *params):
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = zip(params[:10], params[10:])
        dev = x2.device
        _, B, L, dq = x2.shape
        D = wq[0].shape[0]
        dk = D // H
        R, B2 = B * L, 2 * B
        x2 = x2.contiguous()
        ldx = pad8(dq)
        xb = SCRATCH.bf16(2 * R, dq, dev)
        mean = torch.empty(2 * R, device=dev)
        rstd = torch.empty(2 * R, device=dev)
        ops.layernorm_fwd_groups(x2, SHADOWS.bias(*ln_w), SHADOWS.bias(*ln_b), xb, ldx, None, mean, rstd, R, dq, 2)
        s_attn, s_res = SEEDS.next(), SEEDS.next()
        w_qkv = SHADOWS.weight(wq[0], wk[0], wv[0], wq[1], wk[1], wv[1])
        b_qkv = SHADOWS.bias(bq[0], bk[0], bv[0], bq[1], bk[1], bv[1])
        QKV = torch.empty(2 * R, 3 * D, dtype=_BF16, device=dev)
        ops.gemm(xb, w_qkv, R, 3 * D, dq, lda=ldx, ldb=w_qkv.shape[1], batch=(1, 2), a_strides=(0, R * ldx),
                 b_strides=(0, 3 * D * w_qkv.shape[1]), C_bf16=QKV, ldcb=3 * D, cb_strides=(0, R * 3 * D), bias=b_qkv, bias_sb2=3 * D)
        m8, msb, msq = _mask_u8(mask)
        assert m8 is None or m8.shape[0] == B2
        Ob, stats = _attn_core_fwd(QKV, 0, 3 * D, QKV, D, 3 * D, QKV, 2 * D, 3 * D, m8, msb, msq, B2, H, L, L, dk, p_drop, s_attn)
        assert stats[0] == "mat"
        w_o = SHADOWS.weight(*wo)
        y = torch.empty(2, B, L, dq, device=dev)
        ops.gemm(Ob, w_o, R, dq, D, lda=D, ldb=w_o.shape[1], batch=(1, 2), a_strides=(0, R * D), b_strides=(0, dq * w_o.shape[1]),
                 C_f32=y, ldc=dq, c_strides=(0, R * dq), bias=SHADOWS.bias(*bo), bias_sb2=dq, residual=x2, ldr=dq,
                 r_strides=(0, R * dq), dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev, drop_strides=(0, R * dq, dq))
        ctx.save_for_backward(x2, mean, rstd, xb, QKV, Ob, m8, stats[1], *params)
        ctx.cfg = (B, L, dq, D, H, dk, p_drop, s_attn, s_res, msb, msq)
        return y

    @staticmethod
    def backward(ctx, dy2):
        B, L, dq, D, H, dk, p_drop, s_attn, s_res, msb, msq = ctx.cfg
        x2, mean, rstd, xb, QKV, Ob, m8, P = ctx.saved_tensors[:8]
        params = ctx.saved_tensors[8:]
        ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo = zip(params[:10], params[10:])
        dev = dy2.device
        R, B2 = B * L, 2 * B
        ldx = pad8(dq)
        need = ctx.needs_input_grad
        dy2 = dy2.contiguous()
        dyb = SCRATCH.bf16(2 * R, dq, dev)
        dbo = SCRATCH.f32(2 * dq, device=dev)
        ops.cast_colsum_bf16(dy2, dq, dyb, ldx, 2 * R, dq, dbo, dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev, group_rows=R,
                             colsum_stride=dq)
        w_o = SHADOWS.weight(*wo)
        leaf = []                                       # the two weight-gradient products: one grouped launch
        dwo = SCRATCH.f32(2 * dq, D, device=dev)
        ops.gemm(dyb, Ob, dq, D, R, lda=ldx, ldb=D, a_trans=True, b_trans=True, batch=(1, 2), a_strides=(0, R * ldx),
                 b_strides=(0, R * D), C_f32=dwo, ldc=D, c_strides=(0, dq * D), allow_split_k=True, defer=leaf)
        dOb = torch.empty(2 * R, D, dtype=_BF16, device=dev)
        ops.gemm(dyb, w_o, R, D, dq, lda=ldx, ldb=w_o.shape[1], b_trans=True, batch=(1, 2), a_strides=(0, R * ldx),
                 b_strides=(0, dq * w_o.shape[1]), C_bf16=dOb, ldcb=D, cb_strides=(0, R * D), dropout_p=p_drop, seed=s_attn,
                 seed_dev=SEEDS.dev, drop_strides=(0, R * D, D))
        dQKV = torch.empty(2 * R, 3 * D, dtype=_BF16, device=dev)
        _attn_core_bwd(dOb, Ob, ("mat", P), QKV, 0, 3 * D, QKV, D, 3 * D, QKV, 2 * D, 3 * D, dQKV, 0, 3 * D, dQKV, D, 3 * D,
                       dQKV, 2 * D, 3 * D, m8, msb, msq, B2, H, L, L, dk, p_drop)
        db = SCRATCH.f32(2 * 3 * D, device=dev)           # bias gradients [q | k | v] of each half: column sums of its rows
        ops.colsum_bf16_groups(dQKV, 3 * D, db, R, 3 * D, 2, 3 * D)
        w_qkv = SHADOWS.weight(wq[0], wk[0], wv[0], wq[1], wk[1], wv[1])
        dw = SCRATCH.f32(2 * 3 * D, dq, device=dev, zero=not ops.gemm_overwrites(3 * D, dq, R, 2))
        ops.gemm(dQKV, xb, 3 * D, dq, R, lda=3 * D, ldb=ldx, a_trans=True, b_trans=True, batch=(1, 2), a_strides=(0, R * 3 * D),
                 b_strides=(0, R * ldx), C_f32=dw, ldc=dq, c_strides=(0, 3 * D * dq), allow_split_k=True, defer=leaf)
        ops.gemm_flush(leaf)
        dxn = SCRATCH.f32(2 * R, dq, device=dev)         # (zeroed: 80 output tiles over a reduction of 3 D -> split K)
        ops.gemm(dQKV, w_qkv, R, dq, 3 * D, lda=3 * D, ldb=w_qkv.shape[1], b_trans=True, batch=(1, 2), a_strides=(0, R * 3 * D),
                 b_strides=(0, 3 * D * w_qkv.shape[1]), C_f32=dxn, ldc=dq, c_strides=(0, R * dq), allow_split_k=True,
                 split_ws=_split_ws(R, dq, 3 * D, 2, dev))
        dx2 = torch.empty(2, B, L, dq, device=dev)
        out = []
        dlnw2, dlnb2 = SCRATCH.f32(2 * dq, device=dev), SCRATCH.f32(2 * dq, device=dev)
        ops.layernorm_bwd_groups(dxn, x2, SHADOWS.bias(*ln_w), mean, rstd, dx2, dy2, dlnw2, dlnb2, R, dq, 2)
        for i in range(2):
            dlnw, dlnb = dlnw2[i * dq:(i + 1) * dq], dlnb2[i * dq:(i + 1) * dq]
            w0, b0 = i * 3 * D, i * 3 * D
            g = (dlnw, dlnb, dw[w0:w0 + D], db[b0:b0 + D], dw[w0 + D:w0 + 2 * D], db[b0 + D:b0 + 2 * D],
                 dw[w0 + 2 * D:w0 + 3 * D], db[b0 + 2 * D:b0 + 3 * D], dwo[i * dq:(i + 1) * dq], dbo[i * dq:(i + 1) * dq])
            out += [gi if need[4 + 10 * i + j] else None for j, gi in enumerate(g)]
        return (dx2, None, None, None, *out)


class PairRowFn(torch.autograd.Function):
    """LayerNorm (normCA / normCV) of the two stacks' rows in one (2, B, L, D) buffer: two launches, no copies."""

    @staticmethod
    def forward(ctx, x2, wa, ba, wb, bb):
        x2 = x2.contiguous()
        D = x2.shape[-1]
        R = x2[0].numel() // D
        y = torch.empty_like(x2)
        mean = torch.empty(2 * R, device=x2.device)
        rstd = torch.empty(2 * R, device=x2.device)
        for i, (w, b) in enumerate(((wa, ba), (wb, bb))):
            ops.layernorm_fwd(x2[i], w.detach(), b.detach(), None, 0, y[i], mean[i * R:], rstd[i * R:], R, D)
        ctx.save_for_backward(x2, wa, wb, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, wa, wb, mean, rstd = ctx.saved_tensors
        D = x2.shape[-1]
        R = x2[0].numel() // D
        dy = dy.contiguous()
        dx = torch.empty_like(x2)
        gs = []
        for i, w in enumerate((wa, wb)):
            dw, db = SCRATCH.f32(D, device=x2.device), SCRATCH.f32(D, device=x2.device)
            ops.layernorm_bwd(dy[i], x2[i], w, mean[i * R:], rstd[i * R:], dx[i], None, dw, db, R, D)
            gs += [dw, db]
        return dx, gs[0], gs[1], gs[2], gs[3]


class PairGateFn(torch.autograd.Function):
    """GateFn of the two stacks' rows in one (2, B, L, D) buffer (each half with its own a_v_constant)."""

    @staticmethod
    def forward(ctx, cv2, ca2, a_va, a_vb):
        cv2, ca2 = cv2.contiguous(), ca2.contiguous()
        D = cv2.shape[-1]
        R = cv2[0].numel() // D
        out = torch.empty_like(cv2)
        for i, a in enumerate((a_va, a_vb)):
            ops.gate_fwd(cv2[i], ca2[i], a.detach(), out[i], None, 0, R, D)
        ctx.save_for_backward(cv2, ca2, a_va, a_vb)
        return out

    @staticmethod
    def backward(ctx, dout):
        cv2, ca2, a_va, a_vb = ctx.saved_tensors
        D = cv2.shape[-1]
        R = cv2[0].numel() // D
        dout = dout.contiguous()
        dcv, dca = torch.empty_like(cv2), torch.empty_like(ca2)
        das = []
        for i, a in enumerate((a_va, a_vb)):
            da = SCRATCH.f32(1, device=cv2.device)
            ops.gate_bwd(dout[i], cv2[i], ca2[i], a, dcv[i], dca[i], da, R, D)
            das.append(da)
        return dcv, dca, das[0], das[1]


class FusionTailFn(torch.autograd.Function):
    """g * normCV(cv) + (1 - g) * normCA(ca), g = sigmoid(clamp(a_v, -2, 2)) -- the tail of BMFusionLayer.forward
    (model/bm_hrl_agent.py:107-114) as ONE launch forward and ONE backward (two LayerNorms + the gate; it was five / six).
    cv, ca: (..., D) of one stack, or (2, B, L, D) of the paired stacks with `groups` = 2 parameter sets
    (normCA.weight, normCA.bias, normCV.weight, normCV.bias, a_v) x groups, passed flat.  unstack (paired form): the two
    stacks' results are returned as two tensors -- after the last layer they go to different heads, and the backward then reads
    the two incoming gradients where they are (two select-backward fills, two copies and an add otherwise)."""

    @staticmethod
    def forward(ctx, cv, ca, groups, unstack, *params):
        D = cv.shape[-1]
        rows = cv.numel() // D
        assert len(params) == 5 * groups and rows % groups == 0
        cv, ca = cv.contiguous(), ca.contiguous()
        out = torch.empty_like(cv)
        stats = torch.empty(4, rows, device=cv.device)
        gs = [tuple(t.detach() for t in params[5 * i:5 * i + 5]) for i in range(groups)]
        ctx.unstack = bool(unstack) and groups == 2 and out.shape[0] == 2
        # the parted outputs feed projections next (manager linear, goal attention, value heads): their bf16 operands
        # leave this launch too and wait in the step's memo
        ob = SCRATCH.bf16(rows, D, cv.device) if (ctx.unstack and SCRATCH.armed) else None
        ops.fusion_tail_fwd(ca, cv, gs, rows // groups, D, out, stats, ob, ob.shape[1] if ob is not None else 0)
        ctx.save_for_backward(cv, ca, stats, *params)
        ctx.groups = groups
        if not ctx.unstack:
            return out
        o0, o1 = out[0], out[1]
        if ob is not None:
            h = rows // 2
            SCRATCH.offer_bf16(o0, h, D, ob[:h])
            SCRATCH.offer_bf16(o1, h, D, ob[h:])
        return o0, o1

    @staticmethod
    def backward(ctx, *douts):
        cv, ca, stats = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        groups = ctx.groups
        D = cv.shape[-1]
        rows = cv.numel() // D
        dev = cv.device
        need = ctx.needs_input_grad
        dcv, dca = torch.empty_like(cv), torch.empty_like(ca)
        gs = [tuple(t.detach() for t in params[5 * i:5 * i + 5]) for i in range(groups)]
        grads = []
        for i in range(groups):
            g = []
            for j in range(5):
                g.append(SCRATCH.f32(1 if j == 4 else D, device=dev) if need[4 + 5 * i + j] else None)
            grads.append(tuple(g))
        if ctx.unstack:
            halves = []
            for g in douts:                  # a stack nothing was computed from (frozen phase): zero; rows may be strided views
                if g is None:
                    g = torch.zeros(rows // 2, D, device=dev)
                elif g.stride(-1) != 1 or g.dim() < 2 or any(g.stride(i) != g.stride(i + 1) * g.shape[i + 1] for i in range(g.dim() - 2)):
                    g = g.contiguous()
                halves.append(g)
            ops.fusion_tail_bwd(halves[0], ca, cv, stats, gs, grads, rows // groups, D, dca, dcv, dout1=halves[1],
                                ldd0=halves[0].stride(-2), ldd1=halves[1].stride(-2))
        else:
            ops.fusion_tail_bwd(douts[0].contiguous(), ca, cv, stats, gs, grads, rows // groups, D, dca, dcv)
        return (dcv, dca, None, None) + tuple(t for g in grads for t in g)


class AttnCoreFn(torch.autograd.Function):
    """dropout( softmax(q k^T / sqrt(d_k), masked with -1e9) v ) on already projected fp32 (B,S,D) tensors, heads being
    d_k-wide column slices -- model/multihead_attention.py:7-31.  The general entry (q, k and v from three different
    inputs) used by the post-norm layers of model/encoder.py:59-69 and model/decoder.py:66-100; the bimodal hot path
    goes through MHAFn, which fuses the projections around the same kernels."""

    @staticmethod
    def forward(ctx, q, k, v, mask, H, p_drop):
        dev = q.device
        B, Sq, D = q.shape
        Sk = k.shape[1]
        dk = D // H
        if D % 8 or dk % 8:
            raise ValueError("AttnCoreFn needs d_model and d_k to be multiples of 8")
        rows_q, rows_k = B * Sq, B * Sk
        Qb = torch.empty(rows_q, D, dtype=_BF16, device=dev)
        KV = torch.empty(rows_k, 2 * D, dtype=_BF16, device=dev)
        ops.cast_bf16(q.contiguous(), D, Qb, D, rows_q, D)
        ops.cast_bf16(k.contiguous(), D, KV, 2 * D, rows_k, D)
        ops.cast_bf16(v.contiguous(), D, KV, 2 * D, rows_k, D, y_off=D)
        m8, msb, msq = _mask_u8(mask)
        seed = SEEDS.next()
        Ob, stats = _attn_core_fwd(Qb, 0, D, KV, 0, 2 * D, KV, D, 2 * D, m8, msb, msq, B, H, Sq, Sk, dk, p_drop, seed)
        ctx.save_for_backward(Qb, KV, Ob, m8, *stats[1:])
        ctx.cfg = (B, H, Sq, Sk, D, dk, p_drop, seed, stats[0], msb, msq)
        return Ob.float().view(B, Sq, D)

    @staticmethod
    def backward(ctx, dO):
        B, H, Sq, Sk, D, dk, p_drop, seed, kind, msb, msq = ctx.cfg
        Qb, KV, Ob, m8 = ctx.saved_tensors[:4]
        stats = (kind,) + tuple(ctx.saved_tensors[4:])
        dev = dO.device
        rows_q, rows_k = B * Sq, B * Sk
        dOb = torch.empty(rows_q, D, dtype=_BF16, device=dev)   # gradient w.r.t. the pre-dropout output
        ops.cast_bf16(dO.contiguous(), D, dOb, D, rows_q, D, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
        dQb = torch.empty(rows_q, D, dtype=_BF16, device=dev)
        dKV = torch.empty(rows_k, 2 * D, dtype=_BF16, device=dev)
        _attn_core_bwd(dOb, Ob, stats, Qb, 0, D, KV, 0, 2 * D, KV, D, 2 * D, dQb, 0, D, dKV, 0, 2 * D, dKV, D, 2 * D,
                       m8, msb, msq, B, H, Sq, Sk, dk, p_drop)
        need = ctx.needs_input_grad
        dq = dQb.float().view(B, Sq, D) if need[0] else None
        dk_ = dKV[:, :D].float().view(B, Sk, D) if need[1] else None
        dv = dKV[:, D:].float().view(B, Sk, D) if need[2] else None
        return dq, dk_, dv, None, None, None


class FFNFn(torch.autograd.Function):
    """x + dropout( fc2( dropout( relu( fc1( LN(x) ) ) ) ) ) -- model/blocks.py:135-144 around :175-187."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, p_drop):
        dev = x.device
        B, S, d = x.shape
        rows = B * S
        dff = w1.shape[0]
        x = x.contiguous()
        ldx = pad8(d)
        xb = SCRATCH.bf16(rows, d, dev)
        mean = torch.empty(rows, device=dev)
        rstd = torch.empty(rows, device=dev)
        ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), xb, ldx, None, mean, rstd, rows, d)
        s_in, s_res = SEEDS.next(), SEEDS.next()
        wb1, wb2 = SHADOWS.weight(w1), SHADOWS.weight(w2)
        hb = SCRATCH.bf16(rows, dff, dev)
        ops.gemm(xb, wb1, rows, dff, d, lda=ldx, ldb=wb1.shape[1], C_bf16=hb, ldcb=hb.shape[1], bias=b1.detach(), relu=True,
                 dropout_p=p_drop, seed=s_in, seed_dev=SEEDS.dev)
        y = torch.empty(B, S, d, device=dev)
        ops.gemm(hb, wb2, rows, d, dff, lda=hb.shape[1], ldb=wb2.shape[1], C_f32=y, ldc=d, bias=b2.detach(), residual=x, ldr=d,
                 dropout_p=p_drop, seed=s_res, seed_dev=SEEDS.dev)
        ctx.save_for_backward(x, ln_w, mean, rstd, xb, hb, w1, w2)
        ctx.cfg = (B, S, d, dff, p_drop, s_in, s_res)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, d, dff, p_drop, s_in, s_res = ctx.cfg
        x, ln_w, mean, rstd, xb, hb, w1, w2 = ctx.saved_tensors
        dev = dy.device
        rows = B * S
        ldx = pad8(d)
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        dyb, db2 = _cast_dy(dy, rows, d, p_drop, s_res, need[6])
        wb1, wb2 = SHADOWS.weight(w1), SHADOWS.weight(w2)
        # dz = (dy W2) * [h > 0] / (1-p): h already carries relu and the inner dropout mask
        dzb = SCRATCH.bf16(rows, dff, dev)
        keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        db1 = SCRATCH.f32(dff, device=dev) if need[4] else None     # column sums of dz, from the epilogue that writes dz
        dw2, _ = _linear_bwd(dyb, ldx, rows, d, hb, hb.shape[1], dff, wb2, need_dw=need[5], need_db=False, need_dx=True,
                             dx_bf16=dzb, lddxb=dzb.shape[1], dx_epilogue=ops.EPI_RELU_BWD, dx_alpha=keep, dx_aux=hb,
                             ldaux=hb.shape[1], dx_colsum=db1)
        dxn = torch.empty(rows, d, device=dev)
        dw1, _ = _linear_bwd(dzb, dzb.shape[1], rows, dff, xb, ldx, d, wb1, need_dw=need[3], need_db=False, need_dx=True,
                             dx_f32=dxn)
        dx = torch.empty(B, S, d, device=dev)
        dlnw = SCRATCH.f32(d, device=dev) if need[1] else None
        dlnb = SCRATCH.f32(d, device=dev) if need[2] else None
        ops.layernorm_bwd(dxn, x, ln_w, mean, rstd, dx, dy, dlnw, dlnb, rows, d)
        return dx, dlnw, dlnb, dw1, db1, dw2, db2, None


class LinearFn(torch.autograd.Function):
    """y = dropout(relu?(x W^T + b)) for fp32 activations (bf16 operands inside).  Small layers: Manager.linear,
    value heads (model/bm_hrl_agent.py:439,259-260)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, p_drop):
        dev = x.device
        shp = x.shape
        K = shp[-1]
        rows = x.numel() // K
        N = w.shape[0]
        x2 = x.contiguous().view(rows, K)
        xb = SCRATCH.memo_bf16(x2, rows, K)              # (its producer may have offered the bf16 copy: StepScratch.offer_bf16)
        wb = SHADOWS.weight(w)
        y = torch.empty(rows, N, device=dev)
        seed = SEEDS.next()
        ops.gemm(xb, wb, rows, N, K, lda=xb.shape[1], ldb=wb.shape[1], C_f32=y, ldc=N, bias=None if b is None else b.detach(),
                 relu=relu, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
        ctx.save_for_backward(xb, w, y if (relu or p_drop > 0) else None)
        ctx.cfg = (shp, rows, N, K, relu, p_drop, seed, b is not None)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        shp, rows, N, K, relu, p_drop, seed, has_b = ctx.cfg
        xb, w, y = ctx.saved_tensors
        dev = dy.device
        need = ctx.needs_input_grad
        dy2 = dy.contiguous().view(rows, N)
        if y is not None:
            # y = relu(z) * keep-mask: gradient passes where y != 0 with the same 1/(1-p) factor
            keep = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
            dy2 = torch.where(y != 0, dy2 * keep, torch.zeros_like(dy2)) if relu else None
            if dy2 is None:
                dy2 = dy.contiguous().view(rows, N)
        drop = p_drop if not relu else 0.0
        dyb, db = _cast_dy(dy2, rows, N, drop, seed, has_b and need[2])
        wb = SHADOWS.weight(w)
        dx = torch.empty(rows, K, device=dev) if need[0] else None
        dw, _ = _linear_bwd(dyb, dyb.shape[1], rows, N, xb, xb.shape[1], K, wb, need_dw=need[1], need_db=False,
                            need_dx=need[0], dx_f32=dx)
        return (dx.view(shp) if dx is not None else None), dw, db, None, None


class Conv1dSameFn(torch.autograd.Function):
    """nn.Conv1d(C_in, C_out, k, padding='same') over the time axis of x (B, T, C_in) -> (B, T, C_out) (the reference applies
    it to the transposed (B, C, T) tensor: model/det_bmhrl_agent.py:79-86,169-174).  A GEMM over the unfolded operand
    (csrc/conv_gn.hip): rows (b, t), columns (j, c_in); the weight (C_out, C_in, k) is used as (C_out, k * C_in).
    'same' padding: k - 1 zeros in total, (k - 1) // 2 of them in front."""

    @staticmethod
    def _operand(w):
        co, ci, k = w.shape
        w2 = w.detach().permute(0, 2, 1).reshape(co, k * ci)           # [c_out][j * C_in + c_in]
        wb = ops.bf16_zeros(co, k * ci, w.device)
        ops.cast_bf16(w2.contiguous(), k * ci, wb, wb.shape[1], co, k * ci)
        return wb

    @staticmethod
    def forward(ctx, x, w, b):
        B, T, C = x.shape
        co, ci, k = w.shape
        assert ci == C and C % 4 == 0
        left = (k - 1) // 2
        dev = x.device
        K = k * C
        unf = ops.bf16_zeros(B * T, K, dev)
        ops.unfold1d_bf16(x.contiguous(), unf, unf.shape[1], B, T, C, k, left)
        wb = Conv1dSameFn._operand(w)
        y = torch.empty(B * T, co, device=dev)
        ops.gemm(unf, wb, B * T, co, K, lda=unf.shape[1], ldb=wb.shape[1], C_f32=y, ldc=co, bias=None if b is None else b.detach())
        ctx.save_for_backward(unf, w)
        ctx.cfg = (B, T, C, co, k, left, b is not None)
        return y.view(B, T, co)

    @staticmethod
    def backward(ctx, dy):
        B, T, C, co, k, left, has_b = ctx.cfg
        unf, w = ctx.saved_tensors
        dev = dy.device
        K = k * C
        need = ctx.needs_input_grad
        rows = B * T
        dy2 = dy.contiguous().view(rows, co)
        dyb, db = _cast_dy(dy2, rows, co, 0.0, 0, has_b and need[2])
        dx = dw = None
        if need[1]:
            dw2 = torch.zeros(co, K, device=dev)
            ops.gemm(dyb, unf, co, K, rows, lda=dyb.shape[1], ldb=unf.shape[1], a_trans=True, b_trans=True, C_f32=dw2, ldc=K,
                     allow_split_k=True)
            dw = dw2.view(co, k, C).permute(0, 2, 1).contiguous()
        if need[0]:
            wb = Conv1dSameFn._operand(w)
            du = torch.empty(rows, K, device=dev)
            ops.gemm(dyb, wb, rows, K, co, lda=dyb.shape[1], ldb=wb.shape[1], b_trans=True, C_f32=du, ldc=K)
            dx = torch.empty(B, T, C, device=dev)
            ops.fold1d(du, K, dx, B, T, C, k, left)
        return dx, dw, db


class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(G, C) of the reference's input projection (model/det_bmhrl_agent.py:83) on x (B, T, C): statistics over
    the T time steps and C / G channels of a (sample, group)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps):
        B, T, C = x.shape
        dev = x.device
        x = x.contiguous()
        y = torch.empty_like(x)
        mean, rstd = torch.empty(B * G, device=dev), torch.empty(B * G, device=dev)
        ops.groupnorm_fwd(x, gamma.detach(), beta.detach(), y, mean, rstd, B, T, C, G, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.cfg = (B, T, C, G)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, C, G = ctx.cfg
        x, gamma, mean, rstd = ctx.saved_tensors
        dev = dy.device
        dx = torch.empty(B, T, C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.groupnorm_bwd(dy.contiguous(), x, gamma.detach(), mean, rstd, dx, dg, db, B, T, C, G)
        return dx, dg, db, None, None


class LayerNormFn(torch.autograd.Function):
    """fp32 LayerNorm (normCA / normCV, model/bm_hrl_agent.py:107-108)."""

    @staticmethod
    def forward(ctx, x, w, b):
        dev = x.device
        D = x.shape[-1]
        rows = x.numel() // D
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=dev)
        rstd = torch.empty(rows, device=dev)
        ops.layernorm_fwd(x, w.detach(), b.detach(), None, 0, y, mean, rstd, rows, D)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        need = ctx.needs_input_grad
        dx = torch.empty_like(x)
        dw = SCRATCH.f32(D, device=x.device) if need[1] else None
        db = SCRATCH.f32(D, device=x.device) if need[2] else None
        ops.layernorm_bwd(dy.contiguous(), x, w, mean, rstd, dx, None, dw, db, rows, D)
        return dx, dw, db


class GateFn(torch.autograd.Function):
    """g*Cv + (1-g)*Ca, g = sigmoid(clamp(a_v, -2, 2)) -- model/bm_hrl_agent.py:111-114."""

    @staticmethod
    def forward(ctx, cv, ca, a_v):
        D = cv.shape[-1]
        rows = cv.numel() // D
        cv, ca = cv.contiguous(), ca.contiguous()
        out = torch.empty_like(cv)
        ops.gate_fwd(cv, ca, a_v.detach(), out, None, 0, rows, D)
        ctx.save_for_backward(cv, ca, a_v)
        return out

    @staticmethod
    def backward(ctx, dout):
        cv, ca, a_v = ctx.saved_tensors
        D = cv.shape[-1]
        rows = cv.numel() // D
        dcv, dca = torch.empty_like(cv), torch.empty_like(ca)
        da = SCRATCH.f32(1, device=cv.device)
        ops.gate_bwd(dout.contiguous(), cv, ca, a_v, dcv, dca, da, rows, D)
        return dcv, dca, da


class EmbedFn(torch.autograd.Function):
    """(emb(y)*(1-f) + emb(yhat)*f) * sqrt(d) and the same + PE (dropout on the second output only).
    model/blocks.py:44-48,105-112; model/bm_hrl_agent.py:611-625,642.  Output 0 feeds the critic (no grad)."""

    @staticmethod
    def forward(ctx, table, tok, tok2, mix, pe, p_drop):
        B, L = tok.shape
        D = table.shape[1]
        dev = table.device
        emb = torch.empty(B, L, D, device=dev)
        out = torch.empty(B, L, D, device=dev)
        seed = SEEDS.next()
        tok = tok.contiguous()
        tok2 = tok2.contiguous() if tok2 is not None else None
        ops.embed_posenc(tok, tok2, float(mix), table.detach(), pe, emb, out, B, L, D, math.sqrt(D), p_drop, seed, SEEDS.dev)
        ctx.save_for_backward(tok, tok2)
        ctx.cfg = (B, L, D, float(mix), p_drop, seed, table.shape[0])
        ctx.mark_non_differentiable(emb)
        return emb, out

    @staticmethod
    def backward(ctx, _demb, dout):
        B, L, D, mix, p_drop, seed, V = ctx.cfg
        tok, tok2 = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        dev = dout.device
        dout = dout.contiguous()
        if p_drop > 0:
            # regenerate the forward mask: cast kernel applies it, then back to fp32 rows for the scatter
            tmp = SCRATCH.bf16(B * L, D, dev)
            ops.cast_bf16(dout, D, tmp, tmp.shape[1], B * L, D, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
            dout = tmp[:, :D].float().contiguous()
        dtable = SCRATCH.f32(V, D, device=dev)
        ops.embed_bwd(tok, tok2, mix, dout, dtable, B, L, D, math.sqrt(D))
        return dtable, None, None, None, None, None


class ExpandGoalsFn(torch.autograd.Function):
    """Manager.expand_goals (model/bm_hrl_agent.py:415-429) as gather / scatter-add over a precomputed row map.
    explore = (mean_factor, std_factor) adds the manager's exploration vector (reference :444-452) in the same launch; the
    noise is detached there, so the backward does not change.  noise_out: optional (D,) tensor that receives the vector."""

    @staticmethod
    def forward(ctx, goals, seg, explore=None, noise_out=None):
        B, L, D = goals.shape
        dev = goals.device
        seg2 = seg.reshape(B, L).to(torch.int32).contiguous()
        src = torch.empty(B * L, dtype=torch.int32, device=dev)
        out = torch.empty(B, L, D, device=dev)
        ob = SCRATCH.bf16(B * L, D, dev) if SCRATCH.armed else None        # (the goal attention's query operand: no cast launch)
        ldob = ob.shape[1] if ob is not None else 0
        if explore is not None:
            assert L <= 1024 and D <= 1024, "exploration noise: caption length and goal width up to 1024"
            ops.expand_goals_explore(seg2, goals.contiguous(), src, out, ob, ldob, B, L, D, explore[0], explore[1], SEEDS.next(),
                                     SEEDS.dev, noise_out)
        elif L <= 1024:
            ops.expand_goals(seg2, goals.contiguous(), src, out, ob, ldob, B, L, D)
        else:
            ops.expand_goals_index(seg2, src, B, L)
            ops.gather_rows(goals.contiguous(), src, out, ob, ldob, B * L, D)
        if ob is not None:
            SCRATCH.offer_bf16(out, B * L, D, ob)
        ctx.save_for_backward(src)
        ctx.cfg = (B, L, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, L, D = ctx.cfg
        (src,) = ctx.saved_tensors
        dx = SCRATCH.f32(B, L, D, device=dout.device)
        ops.scatter_add_rows(dout.contiguous(), src, dx, B * L, D)
        return dx, None, None, None


# WorkerHeadFn <-> TokenLossFn hand-over.  Nothing is keyed on a raw address: a head registers the log-prob TENSOR it returned
# (weak reference + version + a token it keeps in its ctx); a TokenLossFn(sole_consumer=True) that is given that very tensor
# object takes the token, and in backward leaves the bf16 d logits its gradient kernel wrote under the token.  The gradient it
# RETURNS for the log-probs is a zero-stride view of one NaN (no memory, no launch): the head's backward checks that exactly
# that placeholder arrives -- a second consumer of the log-probs, a tensor hook or anything else that changes the gradient on
# its way raises there, and whatever reads the placeholder elsewhere (retain_grad) sees NaN, never uninitialised memory.
_HEAD_LOGP = {}          # id(logp) -> (weakref(logp), version, shape, token)
_GRAD_TWIN = {}          # token -> bf16 d logits
_HEAD_TOKENS = itertools.count(1)
_NAN_SCALAR = {}         # device -> 0-d NaN tensor (the placeholder gradient is an expand() of it)
# The warmstart trainer announces its loss before the agent's forward (request_head_loss): the worker head then runs the
# log-softmax, the label-smoothing row sums, the token-normalised loss and d logits as ONE launch (ops.head_loss) and
# TokenLossFn, finding the result under the head's token, launches nothing -- forward or backward.
_HEAD_LOSS_REQ = []
_HEAD_LOSS_OUT = {}      # token -> (request, flat targets, [loss, scale], bf16 d logits)
_HEAD_LOSS_COUNTER = {}  # device -> the kernel's four sync words when no trainer's ScratchState is bound (unit tests)


def _nan_placeholder(shape, device):
    n = _NAN_SCALAR.get(device)
    if n is None:
        n = _NAN_SCALAR[device] = torch.full((), float("nan"), device=device)
    return n.expand(*shape)


def _is_placeholder(t: torch.Tensor) -> bool:
    n = _NAN_SCALAR.get(t.device)
    return n is not None and t.data_ptr() == n.data_ptr() and all(st == 0 for st in t.stride()) and n._version == 0


def _head_token_of(logp: torch.Tensor):
    """token of the WorkerHeadFn that returned exactly this tensor object, unchanged since; None otherwise"""
    e = _HEAD_LOGP.get(id(logp))
    if e is None or e[0]() is not logp or e[1] != logp._version or e[2] != tuple(logp.shape):
        return None
    return e[3]


FUSED_HEAD_LOSS = os.environ.get("BMHRL_FUSED_HEAD_LOSS", "1") == "1"


def request_head_loss(trg, smoothing, pad_idx, factor, weight, dloss):
    """trg (B, L) int64 targets, weight: device scalar or None, dloss: the tensor backward() will be given (a constant)."""
    del _HEAD_LOSS_REQ[:]
    if FUSED_HEAD_LOSS and trg.is_cuda:
        _HEAD_LOSS_REQ.append((trg, float(smoothing), int(pad_idx), float(factor), weight, dloss))


def cancel_head_loss():
    del _HEAD_LOSS_REQ[:]


class WorkerHeadFn(torch.autograd.Function):
    """log_softmax( Linear_{(d_in+d_goal) -> V}( cat[x, goal_completion] ) ) -- model/bm_hrl_agent.py:483-484,463-466.
    The concatenation only exists as the bf16 GEMM operand; logits / log-probs stay fp32.  The log-probs are what north_star
    bounds (1e-3 relative) and plain bf16 operands of this last product alone cost 6.6e-4 of it (0.3 % operand rounding over
    K = 364 against |log p| ~ 9), so both operands are split hi + lo: activation [x_hi | x_hi | x_lo] against the weight shadow
    [W_hi | W_lo | W_hi] (ShadowCache.weight_split3), ONE GEMM with K = 3 * 384.  Backward uses the hi blocks only."""

    @staticmethod
    def forward(ctx, x, gc, w, b):
        dev = x.device
        B, L, d1 = x.shape
        d2 = gc.shape[-1]
        rows, K, V = B * L, d1 + d2, w.shape[0]
        part = ShadowCache.split_part(K)
        ld = 3 * part
        xb = SCRATCH.zeroed_bf16(rows, ld, dev)                 # (padding columns K .. part of each block stay zero)
        ops.cast_split3_bf16(x.contiguous(), d1, xb, ld, part, 2, rows, d1, x2=gc.contiguous(), ldx2=d2, cols2=d2)
        wb = SHADOWS.weight_split3(w)
        logp = torch.empty(B, L, V, device=dev)
        ops.gemm(xb, wb, rows, V, ld, lda=ld, ldb=ld, C_f32=logp, ldc=V, bias=b.detach())
        req = _HEAD_LOSS_REQ.pop() if _HEAD_LOSS_REQ else None
        _HEAD_LOSS_OUT.clear()
        token = next(_HEAD_TOKENS)
        gb = SCRATCH.bf16(rows, V, dev) if req is not None else None
        if gb is not None and req[0].numel() == rows and ops.head_loss_ok(V, V, gb.shape[1]):
            trg, smoothing, pad_idx, factor, weight, dloss = req
            counter = SCRATCH.sync_words(dev)                # 64-bit sum, arrival count, non-finite flags: the bound trainer's own
            trg = trg.contiguous().view(-1)
            row_loss = torch.empty(rows, device=dev)
            out = torch.empty(2, device=dev)                 # [loss, scale]
            ops.head_loss(logp, V, trg, smoothing, pad_idx, weight, factor, dloss, row_loss, out, gb, gb.shape[1], counter, rows, V)
            _HEAD_LOSS_OUT[token] = (req, trg, out, gb)
        else:
            ops.log_softmax_(logp, V, rows, V)
        ctx.save_for_backward(xb, w, logp)
        ctx.cfg = (B, L, d1, d2, V)
        ctx.token = token
        for k in [k for k, e in _HEAD_LOGP.items() if e[0]() is None]:      # (entries of tensors that are gone)
            del _HEAD_LOGP[k]
        _HEAD_LOGP[id(logp)] = (weakref.ref(logp), logp._version, tuple(logp.shape), token)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        B, L, d1, d2, V = ctx.cfg
        xb, w, logp = ctx.saved_tensors
        dev = dlogp.device
        rows, K = B * L, d1 + d2
        need = ctx.needs_input_grad
        gb = _GRAD_TWIN.pop(ctx.token, None)              # TokenLossFn(sole_consumer=True): d logits, already bf16
        _GRAD_TWIN.clear()
        if gb is not None and not _is_placeholder(dlogp):
            raise RuntimeError("TokenLossFn(sole_consumer=True) handed its d logits to the worker head, but the gradient that "
                               "reached the head is not that node's placeholder: the log-probabilities have a second consumer, "
                               "a tensor hook or a retained gradient. Pass sole_consumer=False.")
        if gb is None and _is_placeholder(dlogp):
            raise RuntimeError("the worker head received TokenLossFn's placeholder gradient without the d logits that go with it")
        if gb is None:
            gb = SCRATCH.bf16(rows, V, dev)
            ops.log_softmax_bwd(dlogp.contiguous(), logp, V, gb, gb.shape[1], rows, V)
        wb = SHADOWS.weight_split3(w)                           # block 0 of both operands = the plain bf16 copies
        # d cat[x, gc] = d logits W: 480 x 364 outputs over a reduction of V = 10 172 -- 48 tiles; split over K it fills the chip
        dcat = SCRATCH.f32(rows, K, device=dev) if (need[0] or need[1]) else None
        dw, db = _linear_bwd(gb, gb.shape[1], rows, V, xb, xb.shape[1], K, wb, need_dw=need[2], need_db=need[3],
                             need_dx=dcat is not None, dx_f32=dcat, dx_split_k=True)
        dx = dgc = None
        if dcat is not None:
            dcat = dcat.view(B, L, K)
            dx, dgc = dcat[..., :d1], dcat[..., d1:]
        return dx, dgc, dw, db


class PosEncFn(torch.autograd.Function):
    """a (+ b) + PE[:S] with dropout (K1: rgb + flow + PE, audio + PE).  The features are leaf inputs without
    gradients in the reference's loops; a gradient is still passed through (scaled by the keep mask)."""

    @staticmethod
    def forward(ctx, a, b, pe, p_drop):
        B, S, D = a.shape
        out = torch.empty(B, S, D, device=a.device)
        seed = SEEDS.next()
        ops.add_posenc(a.contiguous(), None if b is None else b.contiguous(), pe, out, None, 0, B, S, D, p_drop, seed, SEEDS.dev)
        ctx.cfg = (B, S, D, p_drop, seed, b is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, D, p_drop, seed, has_b = ctx.cfg
        g = dout
        if p_drop > 0:
            tmp = SCRATCH.bf16(B * S, D, dout.device)
            ops.cast_bf16(dout.contiguous(), D, tmp, tmp.shape[1], B * S, D, dropout_p=p_drop, seed=seed, seed_dev=SEEDS.dev)
            g = tmp[:, :D].float().view(B, S, D)
        return g, (g if has_b else None), None, None


class SmoothKLFn(torch.autograd.Function):
    """Unreduced-sum form of LabelSmoothing / BiasedKL: returns the per-row sums (B*S,) of the (B*S, V) divergence
    and differentiates w.r.t. the log-probs (incl. the amplitude path).  loss/label_smoothing.py:12-32,
    loss/biased_kl.py:22-53, epoch_loops/captioning_bmrl_loops.py:285,321-322."""

    @staticmethod
    def forward(ctx, logp, trg, biased_trg, score, n_row, smoothing, pad_idx):
        B, S, V = logp.shape
        rows = B * S
        dev = logp.device
        logp = logp.contiguous()
        trg = trg.contiguous().view(-1)
        bt = biased_trg.contiguous().view(-1) if biased_trg is not None else None
        sc = score.contiguous().view(-1).float() if score is not None else None
        nr = n_row.contiguous().view(-1).float() if n_row is not None else None
        row_loss = torch.empty(rows, device=dev)
        amp = torch.empty(rows, device=dev) if bt is not None else None
        ops.smooth_kl_fwd(logp, V, trg, bt, sc, nr, smoothing, pad_idx, -1, row_loss, amp, rows, V)
        ctx.save_for_backward(logp, trg, bt, sc, nr)
        ctx.cfg = (B, S, V, smoothing, pad_idx)
        ctx.mark_non_differentiable(amp) if amp is not None else None
        return row_loss, amp

    @staticmethod
    def backward(ctx, drow, _damp):
        B, S, V, smoothing, pad_idx = ctx.cfg
        logp, trg, bt, sc, nr = ctx.saved_tensors
        rows = B * S
        dev = logp.device
        one = torch.ones(1, device=dev)
        g = torch.empty(rows, V, device=dev)
        ops.smooth_kl_bwd(logp, V, trg, bt, sc, nr, smoothing, pad_idx, -1, one, None, 0, g, rows, V, wrt_logits=False)
        g = g * drow.contiguous().view(rows, 1)
        return g.view(B, S, V), None, None, None, None, None, None


class GivenAmpKLFn(torch.autograd.Function):
    """BiasedKL.forward of the reference as its signature reads (loss/biased_kl.py:22-53): the amplitude `biased_offset` is an
    ordinary tensor ARGUMENT.  Whether the divergence's gradient reaches the prediction through it is autograd's business -- an
    amplitude computed from the prediction (the loops' clamp(score * p(a) * n, 0, 1)) carries it, a detached one does not --, so
    this node differentiates w.r.t. both inputs: d rows / d log-probs with the amplitude held fixed, and d rows / d amplitude.
    Same kernels as SmoothKLFn (which forms the amplitude itself and therefore always includes the path through p(a)): the
    given amplitude is reproduced as score' * p(a) with score' = amp / p(a); bmhrl_smooth_kl_amp_grad gives the share e of the
    gradient that SmoothKLFn's backward sends through the row's own token, which is taken out of d log-probs again and,
    divided by the amplitude, is d rows / d amp.  Amplitudes are expected in [0, 1] (every caller clamps them)."""

    @staticmethod
    def forward(ctx, logp, trg, biased_trg, amp, smoothing, pad_idx):
        B, S, V = logp.shape
        rows = B * S
        dev = logp.device
        logp = logp.contiguous()
        trg = trg.contiguous().view(-1)
        bt = biased_trg.contiguous().view(-1)
        a = amp.detach().float().contiguous().view(-1)
        p = torch.gather(logp.view(rows, V), 1, bt.view(rows, 1)).squeeze(1).exp().clamp_min(1e-30)
        one = torch.ones_like(a)
        row_loss = torch.empty(rows, device=dev)
        ops.smooth_kl_fwd(logp, V, trg, bt, (a / p).contiguous(), one, smoothing, pad_idx, -1, row_loss, None, rows, V)
        ctx.save_for_backward(logp, trg, bt, a, p)
        ctx.cfg = (B, S, V, smoothing, pad_idx, tuple(amp.shape))
        return row_loss

    @staticmethod
    def backward(ctx, drow):
        B, S, V, smoothing, pad_idx, amp_shape = ctx.cfg
        logp, trg, bt, a, p = ctx.saved_tensors
        rows = B * S
        dev = logp.device
        one = torch.ones(rows, device=dev)
        drow = drow.contiguous().view(rows)
        g_logp = g_amp = None
        sc = (a / p).contiguous()
        if ctx.needs_input_grad[0]:
            g = torch.empty(rows, V, device=dev)
            ops.smooth_kl_bwd(logp, V, trg, bt, sc, one, smoothing, pad_idx, -1, one[:1], None, 0, g, rows, V, wrt_logits=False)
            e = torch.empty(rows, device=dev)
            ops.smooth_kl_amp_grad(logp, V, trg, bt, sc, one, smoothing, pad_idx, -1, e, rows, V)
            g.scatter_add_(1, bt.view(rows, 1), (-e).view(rows, 1))        # the amplitude is an input here, not a function of p(a)
            g_logp = (g * drow.view(rows, 1)).view(B, S, V)
        if ctx.needs_input_grad[3]:
            # d rows / d amp = e / amp; evaluated at >= 1e-6 so that an amplitude of exactly 0 has its (finite) slope
            # (and at most 1 - 1e-6: (amp / p) * p may round to just above 1, which the kernel reads as a clamped amplitude)
            ac = a.clamp(1e-6, 1.0 - 1e-6)
            e = torch.empty(rows, device=dev)
            ops.smooth_kl_amp_grad(logp, V, trg, bt, (ac / p).contiguous(), one, smoothing, pad_idx, -1, e, rows, V)
            g_amp = (e / ac * drow).view(amp_shape)
        return g_logp, None, None, g_amp, None, None


class TokenLossFn(torch.autograd.Function):
    """weight * sum(LabelSmoothing / BiasedKL rows) / (n_tokens * factor) as ONE autograd node -- the reduction the loops
    write as `torch.sum(criterion(pred, y)) / n_tokens` (epoch_loops/captioning_bmrl_loops.py:1156-1158; `/ (n_tokens *
    loss_factor)` at :846-862).  Same kernels as SmoothKLFn; the token count, the division and the data-parallel weight are
    a one-block reduce, and the backward hands its scalar straight to the gradient kernel (no (rows, V) multiply pass)."""

    @staticmethod
    def forward(ctx, logp, trg, biased_trg, score, n_row, smoothing, pad_idx, factor, weight, sole_consumer=False):
        """sole_consumer: the caller guarantees that nothing else differentiates through `logp` -- when logp is the output of
        WorkerHeadFn its backward then takes the bf16 d logits straight from this node's gradient kernel (log-softmax
        backward folded in: no fp32 (rows, V) gradient, no log_softmax_bwd launch)."""
        B, S, V = logp.shape
        rows = B * S
        dev = logp.device
        token = _head_token_of(logp) if (sole_consumer and logp.is_contiguous()) else None
        ctx.twin = token is not None
        ctx.token = token
        ctx.fused = None
        done = _HEAD_LOSS_OUT.pop(token, None) if ctx.twin and biased_trg is None else None
        _HEAD_LOSS_OUT.clear()
        if done is not None:
            (r_trg, r_s, r_pad, r_f, r_w, r_dl), trg_flat, out, gb = done
            if (r_trg.data_ptr() == trg.data_ptr() and r_trg.shape == trg.shape and r_s == float(smoothing) and r_pad == int(pad_idx)
                    and r_f == float(factor) and (r_w is weight or (r_w is not None and weight is not None
                                                                     and r_w.data_ptr() == weight.data_ptr()))):
                # the head already ran the whole tail (ops.head_loss): nothing to launch here
                ctx.fused = (gb, None if r_dl is None else r_dl.data_ptr())
                ctx.save_for_backward(logp, trg_flat, None, None, None, out)
                ctx.cfg = (B, S, V, smoothing, pad_idx)
                return out[0]
            # (announced for other targets / parameters than this call's: the log-probs are log-probs all the same; the
            # unfused kernels below take over)
        logp = logp.contiguous()
        trg = trg.contiguous().view(-1)
        bt = biased_trg.contiguous().view(-1) if biased_trg is not None else None
        sc = score.contiguous().view(-1).float() if score is not None else None
        nr = n_row.contiguous().view(-1).float() if n_row is not None else None
        row_loss = torch.empty(rows, device=dev)
        amp = torch.empty(rows, device=dev) if bt is not None else None
        ops.smooth_kl_fwd(logp, V, trg, bt, sc, nr, smoothing, pad_idx, -1, row_loss, amp, rows, V)
        out = torch.empty(2, device=dev)                     # [loss, scale]
        ops.token_loss_reduce(row_loss, trg, rows, pad_idx, weight, float(factor), out[0:1], out[1:2])
        ctx.save_for_backward(logp, trg, bt, sc, nr, out)
        ctx.cfg = (B, S, V, smoothing, pad_idx)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        B, S, V, smoothing, pad_idx = ctx.cfg
        logp, trg, bt, sc, nr, out = ctx.saved_tensors
        rows = B * S
        dl = dloss.reshape(1)                                # (one element; multiplied in by the kernel)
        if ctx.twin:
            # the head's backward finds the bf16 d logits under its token; what travels through autograd is a NaN placeholder
            if ctx.fused is not None and ctx.fused[1] == dloss.data_ptr():
                gb = ctx.fused[0]                            # d logits were final in the forward
            else:
                gb = ctx.fused[0] if ctx.fused is not None else SCRATCH.bf16(rows, V, logp.device)
                ops.smooth_kl_bwd(logp, V, trg, bt, sc, nr, smoothing, pad_idx, -1, out[1:2], gb, gb.shape[1], None, rows, V,
                                  wrt_logits=True, loss_scale2=dl)
            _GRAD_TWIN[ctx.token] = gb
            return _nan_placeholder((B, S, V), logp.device), None, None, None, None, None, None, None, None, None
        g = torch.empty(rows, V, device=logp.device)
        ops.smooth_kl_bwd(logp, V, trg, bt, sc, nr, smoothing, pad_idx, -1, out[1:2], None, 0, g, rows, V, wrt_logits=False,
                          loss_scale2=dl)
        return g.view(B, S, V), None, None, None, None, None, None, None, None, None


class ManagerKLFn(torch.autograd.Function):
    """BiasedKL of the manager branch of biased_kl() (epoch_loops/captioning_bmrl_loops.py:299-334): the amplitude of a
    position is clamp(score * PROD_{j in its segment} p(a_j) * n_segments, 0, 1) -- the product of the arg-max tokens'
    probabilities over the segment, still attached to the prediction.  Same kernels as SmoothKLFn: a row's amplitude is
    written as score * p(a_row) * n_eff with n_eff = n_segments * (segment product / p(a_row)), which gives the value
    and the gradient through the row's own token; the gradient through the OTHER tokens of the segment is the per-row
    amplitude gradient (bmhrl_smooth_kl_amp_grad) summed over the segment.  Returns (row sums (B*S,), amplitude (B*S,))."""

    @staticmethod
    def forward(ctx, logp, trg, biased_trg, score, n_seg, segments, smoothing, pad_idx):
        from . import rl_glue
        B, S, V = logp.shape
        rows = B * S
        dev = logp.device
        logp = logp.contiguous()
        trg = trg.contiguous().view(-1)
        bt = biased_trg.contiguous().view(-1)
        p = torch.gather(logp, 2, biased_trg.unsqueeze(-1)).squeeze(-1).exp()            # p(a), (B, S)
        sid, n_segs, in_seg = rl_glue._segment_layout(segments)
        segprob = torch.where(in_seg, rl_glue._segment_reduce(p.float(), sid, in_seg, "prod"), torch.zeros_like(p))
        n_eff = (n_seg.float().expand_as(p) * segprob / p.clamp_min(1e-30)).contiguous().view(-1)
        sc = score.contiguous().view(-1).float()
        row_loss = torch.empty(rows, device=dev)
        amp = torch.empty(rows, device=dev)
        ops.smooth_kl_fwd(logp, V, trg, bt, sc, n_eff, smoothing, pad_idx, -1, row_loss, amp, rows, V)
        ctx.save_for_backward(logp, trg, bt, sc, n_eff, sid, in_seg)
        ctx.cfg = (B, S, V, smoothing, pad_idx)
        ctx.mark_non_differentiable(amp)
        return row_loss, amp

    @staticmethod
    def backward(ctx, drow, _damp):
        from . import rl_glue
        B, S, V, smoothing, pad_idx = ctx.cfg
        logp, trg, bt, sc, n_eff, sid, in_seg = ctx.saved_tensors
        rows = B * S
        dev = logp.device
        one = torch.ones(1, device=dev)
        g = torch.empty(rows, V, device=dev)
        ops.smooth_kl_bwd(logp, V, trg, bt, sc, n_eff, smoothing, pad_idx, -1, one, None, 0, g, rows, V, wrt_logits=False)
        drow = drow.contiguous().view(rows)
        g = g * drow.view(rows, 1)
        e = torch.empty(rows, device=dev)
        ops.smooth_kl_amp_grad(logp, V, trg, bt, sc, n_eff, smoothing, pad_idx, -1, e, rows, V)
        e = (e * drow).view(B, S)
        seg_sum = torch.where(in_seg, rl_glue._segment_reduce(e, sid, in_seg, "sum"), torch.zeros_like(e))
        cross = (seg_sum - torch.where(in_seg, e, torch.zeros_like(e))).view(rows, 1)     # the other tokens of the segment
        g.scatter_add_(1, bt.view(rows, 1), cross)
        return g.view(B, S, V), None, None, None, None, None, None, None


class ReinforceFn(torch.autograd.Function):
    """-mean(adv.detach * log clamp(p(a), 1e-5, 1-1e-5)) + mean(adv^2), adv = value - critic_value
    (loss/biased_kl.py:69-81; `probs` are probabilities, as the reference passes them)."""

    @staticmethod
    def forward(ctx, probs, action, value, critic_value):
        B, S, V = probs.shape
        rows = B * S
        dev = probs.device
        probs = probs.contiguous()
        a = action.reshape(-1).contiguous()
        v = value.reshape(-1).float().contiguous()
        c = critic_value.reshape(-1).float().contiguous()
        rp = torch.empty(rows, device=dev)
        rv = torch.empty(rows, device=dev)
        ops.reinforce_fwd(probs, V, a, v, c, rp, rv, rows, V, is_logp=False)
        ctx.save_for_backward(probs, a, v, c)
        ctx.cfg = (B, S, V, value.shape, critic_value.shape)
        return rp.mean() + rv.mean()

    @staticmethod
    def backward(ctx, dloss):
        B, S, V, vshape, cshape = ctx.cfg
        probs, a, v, c = ctx.saved_tensors
        rows = B * S
        dev = probs.device
        need = ctx.needs_input_grad
        dprobs = torch.empty(B, S, V, device=dev)
        dv = torch.empty(rows, device=dev) if need[2] else None
        dc = torch.empty(rows, device=dev) if need[3] else None
        ops.reinforce_bwd(probs, V, a, v, c, dloss.reshape(1).float().contiguous(), dprobs, dv, dc, rows, V)
        return (dprobs if need[0] else None), None, (dv.view(vshape) if dv is not None else None), \
            (dc.view(cshape) if dc is not None else None)
