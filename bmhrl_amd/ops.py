"""Thin torch-tensor wrappers over the C ABI (include/bmhrl_hip.h).  Tensors only lend their device pointers;
every call is enqueued on torch's current HIP stream.  No wrapper has a CPU path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib

EPI_LINEAR, EPI_PROB, EPI_DSCORE, EPI_RELU_BWD = 0, 1, 2, 3


def attention_max_keys() -> int:
    """largest Sk of the fused attention kernels (bmhrl_attention_max_keys)"""
    return _lib.load().bmhrl_attention_max_keys()


def pad8(n: int) -> int:
    return (n + 7) & ~7


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("bmhrl_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def bf16_zeros(rows: int, cols: int, device) -> torch.Tensor:
    """bf16 (rows, pad8(cols)) buffer with zeroed padding columns (no fill when there is no padding)."""
    if cols % 8 == 0:
        return torch.empty(rows, cols, dtype=torch.bfloat16, device=device)
    return torch.zeros(rows, pad8(cols), dtype=torch.bfloat16, device=device)


def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, lda: int, ldb: int, a_trans=False, b_trans=False,
         batch: Tuple[int, int] = (1, 1), a_strides=(0, 0), b_strides=(0, 0), a_off=0, b_off=0,
         C_f32: Optional[torch.Tensor] = None, ldc=0, c_strides=(0, 0), c_off=0,
         C_bf16: Optional[torch.Tensor] = None, ldcb=0, cb_strides=(0, 0), cb_off=0,
         epilogue=EPI_LINEAR, alpha=1.0, relu=False, accumulate=False, bias=None, residual=None, ldr=0,
         r_strides=(0, 0), mask=None, mask_sb1=0, mask_sm=0, rowvec=None, rowvec2=None, rv_strides=(0, 0),
         aux=None, ldaux=0, aux_strides=(0, 0), aux_off=0, dropout_p=0.0, seed=0, seed_dev=None, drop_strides=(0, 0, 0), allow_split_k=False,
         colsum: Optional[torch.Tensor] = None, colsum_off=0, colsum_sb2=0, bias_sb2=0, colsum_sb1=0, bias_sb1=0,
         defer: Optional[list] = None, split_ws: Optional[torch.Tensor] = None) -> None:
    """C[b] = epilogue(A[b] @ B[b]); offsets are in elements from the tensors' data pointers.
    split_ws: fp32 workspace of >= gemm_splits(M, N, K, batch) * batch * M * N elements: a K split then runs as partial tiles +
    an ordered second pass instead of fp32 atomics (reproducible; C needs no zeroing).
    defer: a list -- the problem is appended to it instead of launched; gemm_flush(list) then launches up to four of them as
    ONE kernel (bmhrl_gemm_group: leaf products nothing in between depends on)."""
    _need_cuda(A, B, C_f32, C_bf16)
    d = _lib.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch1, d.batch2 = batch
    d.A = A.data_ptr() + 2 * a_off; d.lda = lda; d.a_sb1, d.a_sb2 = a_strides; d.a_trans = int(a_trans)
    d.B = B.data_ptr() + 2 * b_off; d.ldb = ldb; d.b_sb1, d.b_sb2 = b_strides; d.b_trans = int(b_trans)
    d.C = None if C_f32 is None else C_f32.data_ptr() + 4 * c_off; d.ldc = ldc; d.c_sb1, d.c_sb2 = c_strides
    d.Cb = None if C_bf16 is None else C_bf16.data_ptr() + 2 * cb_off; d.ldcb = ldcb; d.cb_sb1, d.cb_sb2 = cb_strides
    d.epilogue = epilogue; d.alpha = alpha; d.relu = int(relu); d.accumulate = int(accumulate)
    d.allow_split_k = int(allow_split_k)
    d.bias = _p(bias)
    d.residual = _p(residual); d.ldr = ldr; d.r_sb1, d.r_sb2 = r_strides
    d.mask = _p(mask); d.mask_sb1 = mask_sb1; d.mask_sm = mask_sm
    d.rowvec = _p(rowvec); d.rowvec2 = _p(rowvec2); d.rv_sb1, d.rv_sb2 = rv_strides
    d.aux = None if aux is None else aux.data_ptr() + 2 * aux_off; d.ldaux = ldaux; d.aux_sb1, d.aux_sb2 = aux_strides
    d.dropout_p = dropout_p; d.seed = seed; d.seed_dev = _p(seed_dev)
    d.drop_sb1, d.drop_sb2, d.drop_sm = drop_strides
    d.colsum = None if colsum is None else colsum.data_ptr() + 4 * colsum_off; d.colsum_sb2 = colsum_sb2
    d.bias_sb2 = bias_sb2; d.bias_sb1 = bias_sb1; d.colsum_sb1 = colsum_sb1
    d.split_ws = _p(split_ws); d.split_ws_elems = 0 if split_ws is None else split_ws.numel()
    if defer is not None and _GROUP_LEAVES:
        defer.append((d, (A, B, C_f32, C_bf16, bias, residual, mask, rowvec, rowvec2, aux, seed_dev, colsum)))   # (operands kept alive)
        return
    _lib.check(_lib.load().bmhrl_gemm(C.byref(d), stream()), "bmhrl_gemm")


_GROUP_LEAVES = os.environ.get("BMHRL_GEMM_GROUP", "1") == "1"      # (A/B switch: 0 launches deferred problems at once)


def gemm_flush(deferred: list) -> None:
    """launch the problems collected with gemm(..., defer=deferred): one kernel per up to four of them when they share a
    kernel variant, one by one otherwise (bmhrl_gemm_group decides)"""
    if not deferred:
        return
    arr = (_lib.GemmDesc * len(deferred))(*[d for d, _ in deferred])
    _lib.check(_lib.load().bmhrl_gemm_group(arr, len(deferred), stream()), "bmhrl_gemm_group")
    deferred.clear()


_SPLITS = {}


def gemm_splits(M: int, N: int, K: int, batch: int = 1) -> int:
    """K splits the launcher uses for a plain fp32 (M, N) output over a reduction of K with allow_split_k (bmhrl_gemm_splits)"""
    key = (M, N, K, batch, "n")
    r = _SPLITS.get(key)
    if r is None:
        r = _SPLITS[key] = int(_lib.load().bmhrl_gemm_splits(M, N, K, batch))
        if r < 1:
            raise RuntimeError(f"bmhrl_gemm_splits{key[:4]} failed: {r}")
    return r


def gemm_overwrites(M: int, N: int, K: int, batch: int = 1) -> bool:
    """True when gemm(..., C_f32=..., allow_split_k=True) of this shape stores every element of C exactly once (no K
    split, no atomics): the output may then be uninitialised memory.  The launcher's own decision (bmhrl_gemm_splits)."""
    key = (M, N, K, batch)
    r = _SPLITS.get(key)
    if r is None:
        n = int(_lib.load().bmhrl_gemm_splits(M, N, K, batch))
        if n < 1:
            raise RuntimeError(f"bmhrl_gemm_splits{key} failed: {n}")
        r = _SPLITS[key] = n == 1
    return r


def attention_fwd(Q, K, V, O, row_max, row_sum, mask, mask_sb, mask_sq, B, H, Sq, Sk, dk, scale, ldq, ldk, ldv, ldo,
                  q_off=0, k_off=0, v_off=0, dropout_p=0.0, seed=0, seed_dev=None):
    _need_cuda(Q, K, V, O)
    f16 = Q.dtype == torch.float16          # IEEE-half operands: the fp16 build of the same kernel
    if any((t.dtype == torch.float16) != f16 for t in (K, V, O)):
        raise RuntimeError("attention_fwd: Q, K, V and O must share one 16-bit type")
    fn = _lib.load().bmhrl_attention_fwd_f16 if f16 else _lib.load().bmhrl_attention_fwd
    _lib.check(fn(Q.data_ptr() + 2 * q_off, ldq, K.data_ptr() + 2 * k_off, ldk,
                                               V.data_ptr() + 2 * v_off, ldv, O.data_ptr(), ldo, row_max.data_ptr(),
                                               row_sum.data_ptr(), _p(mask), mask_sb, mask_sq, B, H, Sq, Sk, dk, scale,
                                               dropout_p, seed, _p(seed_dev), stream()), "bmhrl_attention_fwd")


def small_attention_ok(Sq: int, Sk: int, dk: int) -> bool:
    """shapes bmhrl_small_attention_fwd / _bwd serve (pure host query)"""
    return bool(_lib.load().bmhrl_small_attention_ok(Sq, Sk, dk))


def small_attention_fwd(Q, K, V, O, P, ldp, mask, mask_sb, mask_sq, B, H, Sq, Sk, dk, scale, ldq, ldk, ldv, ldo,
                        q_off=0, k_off=0, v_off=0, dropout_p=0.0, seed=0, seed_dev=None):
    """one launch: O (bf16 [B*Sq, ldo]) = dropout(softmax(scale Q K^T, masked) V), P (bf16 (B, H, Sq, ldp)) kept for backward"""
    _need_cuda(Q, K, V, O, P)
    _lib.check(_lib.load().bmhrl_small_attention_fwd(Q.data_ptr() + 2 * q_off, ldq, K.data_ptr() + 2 * k_off, ldk,
                                                     V.data_ptr() + 2 * v_off, ldv, O.data_ptr(), ldo, P.data_ptr(), ldp, _p(mask),
                                                     mask_sb, mask_sq, B, H, Sq, Sk, dk, scale, dropout_p, seed, _p(seed_dev),
                                                     stream()), "bmhrl_small_attention_fwd")


def small_attention_bwd(dO, lddo, P, ldp, Q, K, V, dQ, dK, dV, mask, mask_sb, mask_sq, B, H, Sq, Sk, dk, scale, ldq, ldk, ldv,
                        lddq, lddk, lddv, q_off=0, k_off=0, v_off=0, dq_off=0, dk_off=0, dv_off=0, dbq=None, dbk=None, dbv=None,
                        dbq_off=0, dbk_off=0, dbv_off=0):
    """one launch: dQ / dK / dV (bf16, written into column slices) and optionally the bias gradients dbq / dbk / dbv
    (fp32 (H * dk) at the given element offsets, added to)"""
    _need_cuda(dO, P, Q, K, V, dQ, dK, dV)
    fp = lambda t, off: None if t is None else t.data_ptr() + 4 * off
    _lib.check(_lib.load().bmhrl_small_attention_bwd(dO.data_ptr(), lddo, P.data_ptr(), ldp, Q.data_ptr() + 2 * q_off, ldq,
                                                     K.data_ptr() + 2 * k_off, ldk, V.data_ptr() + 2 * v_off, ldv,
                                                     dQ.data_ptr() + 2 * dq_off, lddq, dK.data_ptr() + 2 * dk_off, lddk,
                                                     dV.data_ptr() + 2 * dv_off, lddv, fp(dbq, dbq_off), fp(dbk, dbk_off),
                                                     fp(dbv, dbv_off), _p(mask), mask_sb, mask_sq, B, H, Sq, Sk, dk, scale,
                                                     stream()), "bmhrl_small_attention_bwd")


def memory_attention_ok(L: int, Sk: int, dm: int) -> bool:
    """shapes bmhrl_memory_attention serves (pure host query)"""
    return bool(_lib.load().bmhrl_memory_attention_ok(L, Sk, dm))


def cast_memory(x, y, y_t, B, Sk, dm, ldt):
    """fp32 memory (B * Sk, dm) -> bf16 copy y (B * Sk, dm) and per-sample transposed copy y_t (B, dm, ldt), one launch"""
    _need_cuda(x, y, y_t)
    _lib.check(_lib.load().bmhrl_cast_memory(x.data_ptr(), y.data_ptr(), y_t.data_ptr(), B, Sk, dm, ldt, stream()), "bmhrl_cast_memory")


def memory_attention(backward, X, x_off, ldx, mem, mem_t, ldt, PD, pd_row, pd_slot, Y, ldy, mask, mask_sb, n_mem, B2, H, L, Sk, dm,
                     scale):
    """few-query attention core over a memory, one launch (see bmhrl_memory_attention in include/bmhrl_hip.h)"""
    _need_cuda(X, mem, mem_t, PD, Y)
    _lib.check(_lib.load().bmhrl_memory_attention(int(backward), X.data_ptr() + 2 * x_off, ldx, mem.data_ptr(), Sk * dm,
                                                  mem_t.data_ptr(), dm * ldt, ldt, PD.data_ptr(), pd_row, pd_slot, Y.data_ptr(), ldy,
                                                  _p(mask), mask_sb, n_mem, B2, H, L, Sk, dm, scale, stream()),
               "bmhrl_memory_attention")


def attention_shared128_fwd(Qp, X, ctx, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, ldq, ldx, ldo):
    """absorbed-projection attention: Qp (B,Sq,H,128), X (B,Sk,128) shared by all heads -> ctx (B,Sq,H,128)"""
    _need_cuda(Qp, X, ctx)
    f16 = Qp.dtype == torch.float16
    if any((t.dtype == torch.float16) != f16 for t in (X, ctx)):
        raise RuntimeError("attention_shared128_fwd: Qp, X and ctx must share one 16-bit type")
    fn = _lib.load().bmhrl_attention_shared128_fwd_f16 if f16 else _lib.load().bmhrl_attention_shared128_fwd
    _lib.check(fn(Qp.data_ptr(), ldq, X.data_ptr(), ldx, ctx.data_ptr(), ldo,
                                                         row_max.data_ptr(), row_sum.data_ptr(), _p(mask), mask_sb,
                                                         B, H, Sq, Sk, scale, stream()), "bmhrl_attention_shared128_fwd")


def attention_shared128_bwd(Qp, X, dCx, row_max, row_sum, delta, mask, mask_sb, dQp, dX, accumulate_dx, B, H, Sq, Sk, scale,
                            ldq, ldx, lddo, lddq, lddx=128):
    """fused backward of attention_shared128_fwd: dQp (bf16) and, when dX is given, dX (fp32 (B,Sk,128), += when
    accumulate_dx) -- P / dS are recomputed per tile and never written to HBM"""
    _need_cuda(Qp, X, dCx, dQp)
    lib = _lib.load()
    ws = None
    if dX is not None:
        ws = torch.empty(int(lib.bmhrl_attention_shared128_bwd_workspace(B, H, Sk)), device=Qp.device)
    _lib.check(lib.bmhrl_attention_shared128_bwd(Qp.data_ptr(), ldq, X.data_ptr(), ldx, dCx.data_ptr(), lddo, row_max.data_ptr(),
                                                 row_sum.data_ptr(), delta.data_ptr(), _p(mask), mask_sb, dQp.data_ptr(), lddq,
                                                 _p(dX), lddx, int(bool(accumulate_dx)), _p(ws), B, H, Sq, Sk, scale, stream()),
               "bmhrl_attention_shared128_bwd")


def softmax_bwd_rows(P, ldp, dP, lddp, dS, ldds, rows, cols, scale, mask=None, mask_sb=0, mask_sq=0, rows_per_query=1, queries=1,
                     rows_per_group=0, group_stride=0, p_off=0, ds_off=0):
    """dS = scale * P * (dP - rowsum(P * dP)), 0 at masked keys; P bf16, dP fp32, dS bf16; rows = (sample, query, rows_per_query).
    rows_per_group > 0: P and dS rows in groups of that many, group g at g * group_stride elements (+ p_off / ds_off)"""
    _need_cuda(P, dP, dS)
    _lib.check(_lib.load().bmhrl_softmax_bwd_rows(P.data_ptr() + 2 * p_off, ldp, dP.data_ptr(), lddp, dS.data_ptr() + 2 * ds_off, ldds,
                                                  rows, cols, scale, _p(mask), mask_sb, mask_sq, rows_per_query, queries,
                                                  rows_per_group, group_stride, stream()), "bmhrl_softmax_bwd_rows")


def softmax_rows(S, lds, P, ldp, rows, cols, rows_per_group=0, group_stride=0, p_off=0):
    _lib.check(_lib.load().bmhrl_softmax_rows(S.data_ptr(), lds, P.data_ptr() + 2 * p_off, ldp, rows, cols, rows_per_group,
                                              group_stride, stream()), "bmhrl_softmax_rows")


def attn_delta(dO, lddo, O, ldo, delta, B, H, Sq, dk, scale=1.0):
    _lib.check(_lib.load().bmhrl_attn_delta(dO.data_ptr(), lddo, O.data_ptr(), ldo, delta.data_ptr(), scale, B, H, Sq, dk, stream()),
               "bmhrl_attn_delta")


def attention_bwd_scores256_ok(Sq, Sk, dk, msq):
    return bool(_lib.load().bmhrl_attention_bwd_scores256_ok(Sq, Sk, dk, msq))


def attention_bwd_scores256(Q, ldq, K, ldk, V, ldv, dO, lddo, row_max, row_sum, mask, msb, P, dS, ldp, B, H, Sq, Sk, scale,
                            q_off=0, k_off=0, v_off=0):
    """P and dS (bf16 (B, H, Sq, ldp)) of a head-dimension-256 attention with at most 256 keys from the forward's statistics:
    one launch (csrc/attention_bwd256.hip); offsets in elements"""
    _need_cuda(Q)
    _lib.check(_lib.load().bmhrl_attention_bwd_scores256(Q.data_ptr() + 2 * q_off, ldq, K.data_ptr() + 2 * k_off, ldk,
                                                         V.data_ptr() + 2 * v_off, ldv, dO.data_ptr(), lddo, row_max.data_ptr(),
                                                         row_sum.data_ptr(), _p(mask), msb, P.data_ptr(), dS.data_ptr(), ldp, B, H,
                                                         Sq, Sk, scale, stream()), "bmhrl_attention_bwd_scores256")


def layernorm_fwd(x, gamma, beta, y_bf16, ldy, y_f32, mean, rstd, rows, D):
    _need_cuda(x)
    _lib.check(_lib.load().bmhrl_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(y_bf16), ldy, _p(y_f32),
                                               _p(mean), _p(rstd), rows, D, stream()), "bmhrl_layernorm_fwd")


def layernorm_fwd_groups(x, gamma, beta, y_bf16, ldy, y_f32, mean, rstd, rows_per_group, D, groups):
    """`groups` LayerNorms of rows_per_group rows each, back to back, in one launch; gamma / beta are (groups, D)"""
    _need_cuda(x)
    if gamma.numel() != groups * D or beta.numel() != groups * D:
        raise RuntimeError("layernorm_fwd_groups: gamma / beta must hold one row of D per group")
    _lib.check(_lib.load().bmhrl_layernorm_fwd_groups(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(y_bf16), ldy, _p(y_f32),
                                                      _p(mean), _p(rstd), rows_per_group, D, groups, stream()),
               "bmhrl_layernorm_fwd_groups")


def deterministic() -> bool:
    """BMHRL_DETERMINISTIC as the LIBRARY reads it (bmhrl_deterministic_enabled: one parse for both sides)"""
    global _DETERMINISTIC
    if _DETERMINISTIC is None:
        _DETERMINISTIC = bool(_lib.load().bmhrl_deterministic_enabled())
    return _DETERMINISTIC


_DETERMINISTIC = None


def layernorm_bwd_groups(dy, x, gamma, mean, rstd, dx, dx_add, dgamma, dbeta, rows_per_group, D, groups):
    """backward of layernorm_fwd_groups; dgamma / dbeta (groups, D) are ADDED to (zero them first)"""
    _need_cuda(x)
    if deterministic():         # ordered parameter gradients: the two-stage form with a workspace per call, group by group
        R = rows_per_group
        for g in range(groups):
            sl = lambda t, n: None if t is None else t.reshape(-1)[g * n:(g + 1) * n]
            layernorm_bwd(sl(dy, R * D), sl(x, R * D), sl(gamma, D), sl(mean, R), sl(rstd, R), sl(dx, R * D), sl(dx_add, R * D),
                          sl(dgamma, D), sl(dbeta, D), R, D)
        return
    if gamma.numel() != groups * D or any(t is not None and t.numel() != groups * D for t in (dgamma, dbeta)):
        raise RuntimeError("layernorm_bwd_groups: gamma / dgamma / dbeta must hold one row of D per group")
    _lib.check(_lib.load().bmhrl_layernorm_bwd_groups(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                      dx.data_ptr(), _p(dx_add), _p(dgamma), _p(dbeta), rows_per_group, D, groups,
                                                      stream()), "bmhrl_layernorm_bwd_groups")


def layernorm_bwd_workspace(rows, D) -> int:
    """fp32 elements of the scratch the two-stage column sums of layernorm_bwd go through (uninitialised is fine)."""
    return int(_lib.load().bmhrl_layernorm_bwd_workspace(rows, D))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dx_add, dgamma, dbeta, rows, D, workspace=None):
    if workspace is None:
        workspace = torch.empty(layernorm_bwd_workspace(rows, D), device=x.device)
    _lib.check(_lib.load().bmhrl_layernorm_bwd_ws(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  dx.data_ptr(), _p(dx_add), _p(dgamma), _p(dbeta), rows, D,
                                                  workspace.data_ptr(), workspace.numel(), stream()),
               "bmhrl_layernorm_bwd_ws")


def add_posenc(a, b, pe, out, out_bf16, ldob, B, S, D, dropout_p=0.0, seed=0, seed_dev=None):
    _need_cuda(a)
    _lib.check(_lib.load().bmhrl_add_posenc(a.data_ptr(), _p(b), pe.data_ptr(), out.data_ptr(), _p(out_bf16), ldob, B, S, D,
                                            dropout_p, seed, _p(seed_dev), stream()), "bmhrl_add_posenc")


def embed_posenc(tok, tok2, mix, table, pe, emb_out, out, B, L, D, scale, dropout_p=0.0, seed=0, seed_dev=None):
    _need_cuda(tok, table)
    _lib.check(_lib.load().bmhrl_embed_posenc(tok.data_ptr(), _p(tok2), mix, table.data_ptr(), pe.data_ptr(), _p(emb_out),
                                              out.data_ptr(), B, L, D, scale, dropout_p, seed, _p(seed_dev), stream()), "bmhrl_embed_posenc")


def embed_bwd(tok, tok2, mix, dC, dtable, B, L, D, scale):
    _lib.check(_lib.load().bmhrl_embed_bwd(tok.data_ptr(), _p(tok2), mix, dC.data_ptr(), dtable.data_ptr(), B, L, D, scale,
                                           stream()), "bmhrl_embed_bwd")


def cast_bf16(x, ldx, y, ldy, rows, cols, scale=1.0, dropout_p=0.0, seed=0, y_off=0, seed_dev=None):
    _need_cuda(x, y)
    _lib.check(_lib.load().bmhrl_cast_bf16(x.data_ptr(), ldx, y.data_ptr() + 2 * y_off, ldy, rows, cols, scale, dropout_p,
                                           seed, _p(seed_dev), stream()), "bmhrl_cast_bf16")


def cast_split3_bf16(x, ldx, y, ldy, part, lo_slot, rows, cols, y_off=0, x2=None, ldx2=0, cols2=0):
    """y blocks [hi | . | .] of width `part`: bf16(x) in block 0, bf16(x - hi) in block lo_slot, hi again in the other;
    x2 (rows, cols2): a second source whose columns follow x's inside every block"""
    _need_cuda(x, y)
    _lib.check(_lib.load().bmhrl_cast_split3_bf16(x.data_ptr(), ldx, y.data_ptr() + 2 * y_off, ldy, part, lo_slot, rows, cols,
                                                  _p(x2), ldx2, cols2, stream()), "bmhrl_cast_split3_bf16")


def cast_colsum_bf16(x, ldx, y, ldy, rows, cols, colsum, scale=1.0, dropout_p=0.0, seed=0, seed_dev=None, colsum_off=0,
                     group_rows=None, colsum_stride=0):
    """y = bf16(x * scale * dropout); colsum[colsum_off + n] += column sums of y (colsum must start zeroed).
    group_rows: the rows form groups of that many, group g adds into colsum[colsum_off + g * colsum_stride + n]"""
    _need_cuda(x, y, colsum)
    if group_rows is not None:
        _lib.check(_lib.load().bmhrl_cast_colsum_bf16_groups(x.data_ptr(), ldx, y.data_ptr(), ldy, rows, cols, scale, dropout_p,
                                                             seed, _p(seed_dev), colsum.data_ptr() + 4 * colsum_off, group_rows,
                                                             colsum_stride, stream()), "bmhrl_cast_colsum_bf16_groups")
        return
    _lib.check(_lib.load().bmhrl_cast_colsum_bf16(x.data_ptr(), ldx, y.data_ptr(), ldy, rows, cols, scale, dropout_p, seed,
                                                  _p(seed_dev), colsum.data_ptr() + 4 * colsum_off, stream()),
               "bmhrl_cast_colsum_bf16")


SEG_ELEMS_PER_BLOCK = 4096


def cast_segments(table: torch.Tensor, n_segments: int, n_blocks: int):
    """table: int64 (n_segments, 6) on the device -- see bmhrl_cast_segments in include/bmhrl_hip.h"""
    _need_cuda(table)
    _lib.check(_lib.load().bmhrl_cast_segments(table.data_ptr(), n_segments, n_blocks, stream()), "bmhrl_cast_segments")


def colsum_bf16(dY, ld, db, accumulate, rows, cols, dy_off=0, db_off=0):
    _lib.check(_lib.load().bmhrl_colsum_bf16(dY.data_ptr() + 2 * dy_off, ld, db.data_ptr() + 4 * db_off, int(accumulate),
                                             rows, cols, stream()), "bmhrl_colsum_bf16")


def colsum_bf16_groups(dY, ld, db, rows_per_group, cols, groups, db_stride):
    """db[g * db_stride + n] += sum over the rows of group g of dY[., n] (bf16 (groups * rows_per_group, ld)); one launch"""
    _lib.check(_lib.load().bmhrl_colsum_bf16_groups(dY.data_ptr(), ld, db.data_ptr(), rows_per_group, cols, groups, db_stride,
                                                    stream()), "bmhrl_colsum_bf16_groups")


def cast_bf16_copies(x, ldx, y, ldy, rows, cols, copies, copy_stride):
    """y[c * copy_stride + r * ldy + n] = bf16(x[r * ldx + n]) for every copy c; one launch"""
    _need_cuda(x, y)
    _lib.check(_lib.load().bmhrl_cast_bf16_copies(x.data_ptr(), ldx, y.data_ptr(), ldy, rows, cols, copies, copy_stride, stream()),
               "bmhrl_cast_bf16_copies")


def gate_fwd(cv, ca, a_v, out, out_bf16, ldob, rows, D):
    _lib.check(_lib.load().bmhrl_gate_fwd(cv.data_ptr(), ca.data_ptr(), a_v.data_ptr(), out.data_ptr(), _p(out_bf16), ldob,
                                          rows, D, stream()), "bmhrl_gate_fwd")


def gate_bwd(dout, cv, ca, a_v, dcv, dca, da_v, rows, D):
    _lib.check(_lib.load().bmhrl_gate_bwd(dout.data_ptr(), cv.data_ptr(), ca.data_ptr(), a_v.data_ptr(), dcv.data_ptr(),
                                          dca.data_ptr(), _p(da_v), rows, D, stream()), "bmhrl_gate_bwd")


def _tail_groups(groups, grads=None):
    arr = (_lib.FusionTailParams * len(groups))()
    for i, (g, a) in enumerate(zip(groups, arr)):
        a.gamma_ca, a.beta_ca, a.gamma_cv, a.beta_cv, a.a_v = (t.data_ptr() for t in g)
        if grads is not None:
            a.dgamma_ca, a.dbeta_ca, a.dgamma_cv, a.dbeta_cv, a.da_v = (_p(t) for t in grads[i])
    return arr


def fusion_tail_fwd(ca, cv, groups, rows_per_group, D, out, stats, out_bf16=None, ldob=0):
    """out = g * LN_CV(cv) + (1 - g) * LN_CA(ca); groups: list (1 or 2) of (gamma_ca, beta_ca, gamma_cv, beta_cv, a_v);
    out_bf16 (rows, ldob): the same values in bf16 as well"""
    _need_cuda(ca, cv, out, stats)
    arr = _tail_groups(groups)
    _lib.check(_lib.load().bmhrl_fusion_tail_fwd(ca.data_ptr(), cv.data_ptr(), C.cast(arr, C.c_void_p), len(groups), rows_per_group, D,
                                                 out.data_ptr(), stats.data_ptr(), _p(out_bf16), ldob, stream()), "bmhrl_fusion_tail_fwd")


def fusion_tail_bwd(dout, ca, cv, stats, groups, grads, rows_per_group, D, dca, dcv, dout1=None, ldd0=0, ldd1=0):
    """grads: per group (dgamma_ca, dbeta_ca, dgamma_cv, dbeta_cv, da_v), zeroed fp32 tensors or None.
    dout: gradient of group 0's rows (row stride ldd0, 0 = D); dout1: of group 1's (None: behind group 0's in dout)"""
    _need_cuda(dout, ca, cv, stats, dca, dcv)
    arr = _tail_groups(groups, grads)
    _lib.check(_lib.load().bmhrl_fusion_tail_bwd(dout.data_ptr(), _p(dout1), ldd0, ldd1, ca.data_ptr(), cv.data_ptr(), stats.data_ptr(),
                                                 C.cast(arr, C.c_void_p), len(groups), rows_per_group, D, dca.data_ptr(),
                                                 dcv.data_ptr(), stream()), "bmhrl_fusion_tail_bwd")


def expand_goals_index(seg, src, B, L):
    _lib.check(_lib.load().bmhrl_expand_goals_index(seg.data_ptr(), src.data_ptr(), B, L, stream()), "bmhrl_expand_goals_index")


def expand_goals(seg, x, src, out, out_bf16, ldob, B, L, D):
    """row map of Manager.expand_goals from the int32 labels `seg` (B, L) + the gather along it, one launch"""
    _need_cuda(seg, x, src, out)
    _lib.check(_lib.load().bmhrl_expand_goals(seg.data_ptr(), x.data_ptr(), src.data_ptr(), out.data_ptr(), _p(out_bf16), ldob,
                                              B, L, D, stream()), "bmhrl_expand_goals")


def expand_goals_explore(seg, x, src, out, out_bf16, ldob, B, L, D, mean_factor, std_factor, seed, seed_dev=None, noise_out=None):
    """expand_goals with the manager's exploration vector (reference :444-452) added in the same launch"""
    _need_cuda(seg, x, src, out)
    _lib.check(_lib.load().bmhrl_expand_goals_explore(seg.data_ptr(), x.data_ptr(), src.data_ptr(), out.data_ptr(), _p(out_bf16),
                                                      ldob, B, L, D, float(mean_factor), float(std_factor), int(seed),
                                                      _p(seed_dev), _p(noise_out), stream()), "bmhrl_expand_goals_explore")


def gather_rows(x, src, out, out_bf16, ldob, rows, D):
    _lib.check(_lib.load().bmhrl_gather_rows(x.data_ptr(), src.data_ptr(), out.data_ptr(), _p(out_bf16), ldob, rows, D,
                                             stream()), "bmhrl_gather_rows")


def scatter_add_rows(dout, src, dx, rows, D):
    _lib.check(_lib.load().bmhrl_scatter_add_rows(dout.data_ptr(), src.data_ptr(), dx.data_ptr(), rows, D, stream()),
               "bmhrl_scatter_add_rows")


def log_softmax_(logits, ld, rows, V):
    _lib.check(_lib.load().bmhrl_log_softmax(logits.data_ptr(), ld, rows, V, stream()), "bmhrl_log_softmax")


def smooth_kl_fwd(logp, ld, trg, biased_trg, score, n_row, smoothing, pad_idx, zero_pad_rows, row_loss, amp_out, rows, V):
    _lib.check(_lib.load().bmhrl_smooth_kl_fwd(logp.data_ptr(), ld, trg.data_ptr(), _p(biased_trg), _p(score), _p(n_row),
                                               smoothing, pad_idx, zero_pad_rows, row_loss.data_ptr(), _p(amp_out), rows, V,
                                               stream()), "bmhrl_smooth_kl_fwd")


def smooth_kl_full(logp, ld, trg, biased_trg, score, n_row, smoothing, pad_idx, zero_pad_rows, out, rows, V):
    """the unreduced (rows, V) divergence of LabelSmoothing / BiasedKL (no gradient: the differentiable form is the row sums)"""
    _need_cuda(logp, out)
    _lib.check(_lib.load().bmhrl_smooth_kl_full(logp.data_ptr(), ld, trg.data_ptr(), _p(biased_trg), _p(score), _p(n_row), smoothing,
                                                pad_idx, zero_pad_rows, out.data_ptr(), rows, V, stream()), "bmhrl_smooth_kl_full")


def smooth_kl_bwd(logp, ld, trg, biased_trg, score, n_row, smoothing, pad_idx, zero_pad_rows, loss_scale, g_bf16, ldg,
                  g_f32, rows, V, wrt_logits=True, loss_scale2=None):
    """loss_scale (and the optional loss_scale2, multiplied in) are one-element device tensors"""
    _lib.check(_lib.load().bmhrl_smooth_kl_bwd(logp.data_ptr(), ld, trg.data_ptr(), _p(biased_trg), _p(score), _p(n_row),
                                               smoothing, pad_idx, zero_pad_rows, loss_scale.data_ptr(), _p(loss_scale2),
                                               int(wrt_logits), _p(g_bf16), ldg, _p(g_f32), rows, V, stream()), "bmhrl_smooth_kl_bwd")


def smooth_kl_amp_grad(logp, ld, trg, biased_trg, score, n_row, smoothing, pad_idx, zero_pad_rows, out, rows, V):
    _lib.check(_lib.load().bmhrl_smooth_kl_amp_grad(logp.data_ptr(), ld, trg.data_ptr(), biased_trg.data_ptr(), score.data_ptr(),
                                                    n_row.data_ptr(), smoothing, pad_idx, zero_pad_rows, out.data_ptr(), rows, V,
                                                    stream()), "bmhrl_smooth_kl_amp_grad")


def token_loss_reduce(row_loss, trg, rows, pad_idx, weight, factor, loss, scale):
    """loss = weight * sum(row_loss) / (#(trg != pad) * factor); scale = weight / (#tokens * factor) (device scalars)"""
    _need_cuda(row_loss, trg, loss, scale)
    _lib.check(_lib.load().bmhrl_token_loss_reduce(row_loss.data_ptr(), trg.data_ptr(), rows, pad_idx, _p(weight), factor,
                                                   loss.data_ptr(), scale.data_ptr(), stream()), "bmhrl_token_loss_reduce")


def head_loss_ok(V, ld, ldg):
    return V % 4 == 0 and ld % 4 == 0 and ldg % 4 == 0 and 2 < V <= 12288


def head_loss(logits, ld, trg, smoothing, pad_idx, weight, factor, dloss, row_loss, loss_scale, g_bf16, ldg, counter, rows, V):
    """log-softmax in place + label-smoothing row sums + the loops' token-normalised loss (loss_scale = [loss, scale]) + bf16
    d logits for the incoming gradient `dloss` (device scalar or None = 1), one launch"""
    _need_cuda(logits, trg, row_loss, loss_scale, g_bf16, counter)
    _lib.check(_lib.load().bmhrl_head_loss(logits.data_ptr(), ld, trg.data_ptr(), smoothing, pad_idx, _p(weight), factor, _p(dloss),
                                           row_loss.data_ptr(), loss_scale.data_ptr(), g_bf16.data_ptr(), ldg, counter.data_ptr(),
                                           rows, V, stream()), "bmhrl_head_loss")


def log_softmax_bwd(dlogp, logp, ld, g_bf16, ldg, rows, V):
    _lib.check(_lib.load().bmhrl_log_softmax_bwd(dlogp.data_ptr(), logp.data_ptr(), ld, g_bf16.data_ptr(), ldg, rows, V, stream()),
               "bmhrl_log_softmax_bwd")


def sample_tokens(logp, ld, out, p_out, rows, V, greedy, seed, seed_dev=None, row_offset=0):
    _lib.check(_lib.load().bmhrl_sample_tokens(logp.data_ptr(), ld, out.data_ptr(), _p(p_out), rows, V, int(greedy), seed,
                                               _p(seed_dev), row_offset, stream()), "bmhrl_sample_tokens")


def reinforce_fwd(pred, ld, action, value, critic_value, row_policy, row_value, rows, V, is_logp=True):
    _need_cuda(pred)
    _lib.check(_lib.load().bmhrl_reinforce_fwd(pred.data_ptr(), ld, int(is_logp), action.data_ptr(), value.data_ptr(),
                                               critic_value.data_ptr(), row_policy.data_ptr(), row_value.data_ptr(), rows, V,
                                               stream()), "bmhrl_reinforce_fwd")


def reinforce_bwd(probs, ld, action, value, critic_value, gscale, dprobs, dvalue, dcritic, rows, V):
    _lib.check(_lib.load().bmhrl_reinforce_bwd(probs.data_ptr(), ld, action.data_ptr(), value.data_ptr(),
                                               critic_value.data_ptr(), gscale.data_ptr(), dprobs.data_ptr(), _p(dvalue),
                                               _p(dcritic), rows, V, stream()), "bmhrl_reinforce_bwd")


def adam_step(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, step_dev=None):
    _lib.check(_lib.load().bmhrl_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n, lr,
                                           beta1, beta2, eps, weight_decay, step, _p(step_dev), grad_scale, stream()), "bmhrl_adam_step")


def gemm_f32(A, W, bias1, bias2, C, M, N, K):
    _need_cuda(A, W, C)
    _lib.check(_lib.load().bmhrl_gemm_f32(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), _p(bias1), _p(bias2),
                                          C.data_ptr(), C.stride(0), M, N, K, stream()), "bmhrl_gemm_f32")


def rnn_step(gates, xproj, whh, bhh, h_prev, c_prev, h_out, c_out, seq_out, ar_alpha, ar_beta, B, L, H, t):
    _lib.check(_lib.load().bmhrl_rnn_step(gates, xproj.data_ptr(), whh.data_ptr(), _p(bhh), _p(h_prev), _p(c_prev),
                                          h_out.data_ptr(), _p(c_out), seq_out.data_ptr(), _p(ar_alpha), _p(ar_beta), B, L, H, t,
                                          stream()), "bmhrl_rnn_step")


def rnn_wavefront(layers, B, L, H, chunk=1):
    """layers: list of dicts with the fields of bmhrl_rnn_layer (tensors / None); runs the whole stack in
    L + chunk * (len(layers) - 1) launches"""
    arr = (_lib.RnnLayer * len(layers))()
    for d, a in zip(layers, arr):
        a.w_ih, a.w_hh, a.b_ih, a.b_hh = d["w_ih"].data_ptr(), d["w_hh"].data_ptr(), d["b_ih"].data_ptr(), d["b_hh"].data_ptr()
        a.in_seq, a.in_ld, a.in_dim, a.gates = d["in_seq"].data_ptr(), d["in_ld"], d["in_dim"], d["gates"]
        a.seq_out = d["seq_out"].data_ptr()
        a.h[0], a.h[1] = d["h"][0].data_ptr(), d["h"][1].data_ptr()
        a.c[0], a.c[1] = (d["c"][0].data_ptr(), d["c"][1].data_ptr()) if d.get("c") is not None else (None, None)
        a.arelu_alpha, a.arelu_beta = _p(d.get("arelu_alpha")), _p(d.get("arelu_beta"))
        a.xproj = _p(d.get("xproj"))
    _lib.check(_lib.load().bmhrl_rnn_wavefront(C.cast(arr, C.c_void_p), len(layers), B, L, H, chunk, stream()),
               "bmhrl_rnn_wavefront")


def critic_head(x, w, b, threshold, score, labels, rows, H):
    _lib.check(_lib.load().bmhrl_critic_head(x.data_ptr(), w.data_ptr(), b.data_ptr(), threshold, _p(score), _p(labels), rows, H,
                                             stream()), "bmhrl_critic_head")


def adam_segments(table, n_segments, n_blocks, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step,
                  grad_scale=1.0, step_dev=None):
    """Adam over the flat bucket through a per-parameter table that also names each weight's bf16 shadow (see
    bmhrl_adam_segments in include/bmhrl_hip.h)"""
    _need_cuda(table, param)
    _lib.check(_lib.load().bmhrl_adam_segments(table.data_ptr(), n_segments, n_blocks, param.data_ptr(), grad.data_ptr(),
                                               exp_avg.data_ptr(), exp_avg_sq.data_ptr(), lr, beta1, beta2, eps, weight_decay,
                                               step, _p(step_dev), grad_scale, stream()), "bmhrl_adam_segments")


def make_masks(rgb, audio, trg, pad_idx, copies=1):
    """V_mask (copies*B, 1, Tv), A_mask (copies*B, 1, Ta), C_mask (copies*B, L, L) as bool tensors, one launch
    (model/masking.py:18-55, bimodal case; copies > 1: the same masks laid out that many times)"""
    _need_cuda(rgb, audio, trg)
    B, Tv = rgb.shape[:2]
    Ta, L = audio.shape[1], trg.shape[1]
    if rgb.stride(2) != 1 or rgb.stride(0) != Tv * rgb.stride(1) or audio.stride(2) != 1 or audio.stride(0) != Ta * audio.stride(1):
        raise RuntimeError("make_masks: feature stacks must be (B, T, D) with contiguous rows")
    trg = trg.contiguous()
    vm = torch.empty(copies * B, 1, Tv, dtype=torch.bool, device=rgb.device)
    am = torch.empty(copies * B, 1, Ta, dtype=torch.bool, device=rgb.device)
    cm = torch.empty(copies * B, L, L, dtype=torch.bool, device=rgb.device)
    _lib.check(_lib.load().bmhrl_make_masks(rgb.data_ptr(), rgb.stride(1), audio.data_ptr(), audio.stride(1), trg.data_ptr(), B, Tv,
                                            Ta, L, pad_idx, copies, vm.data_ptr(), am.data_ptr(), cm.data_ptr(), stream()),
               "bmhrl_make_masks")
    return vm, am, cm


def batch_head(rgb, audio, captions, pad_idx, copies=1, bump64=None, bump32=()):
    """Head of a training step, one launch: captions (B, L + 1) -> (trg_in, trg_y) = (captions[:, :-1], captions[:, 1:])
    as contiguous tensors, make_masks() of trg_in, and `bump64` (int64 device word: the seed) / up to two int32 device
    counters in `bump32` (Adam steps) advanced by one.  Returns (vm, am, cm, trg_in, trg_y)."""
    _need_cuda(rgb, audio, captions)
    B, Tv = rgb.shape[:2]
    Ta, L = audio.shape[1], captions.shape[1] - 1
    if rgb.stride(2) != 1 or rgb.stride(0) != Tv * rgb.stride(1) or audio.stride(2) != 1 or audio.stride(0) != Ta * audio.stride(1):
        raise RuntimeError("batch_head: feature stacks must be (B, T, D) with contiguous rows")
    if captions.dtype != torch.int64 or captions.stride(1) != 1 or L < 1:
        raise RuntimeError("batch_head: captions must be (B, L + 1) int64 with contiguous rows")
    if len(bump32) > 2 or any(t.dtype != torch.int32 for t in bump32) or (bump64 is not None and bump64.dtype != torch.int64):
        raise RuntimeError("batch_head: counters are one int64 word and at most two int32 words")
    dev = rgb.device
    vm = torch.empty(copies * B, 1, Tv, dtype=torch.bool, device=dev)
    am = torch.empty(copies * B, 1, Ta, dtype=torch.bool, device=dev)
    cm = torch.empty(copies * B, L, L, dtype=torch.bool, device=dev)
    trg_in = torch.empty(B, L, dtype=torch.int64, device=dev)
    trg_y = torch.empty(B, L, dtype=torch.int64, device=dev)
    b32 = list(bump32) + [None, None]
    _lib.check(_lib.load().bmhrl_batch_head(rgb.data_ptr(), rgb.stride(1), audio.data_ptr(), audio.stride(1), captions.data_ptr(),
                                            captions.stride(0), B, Tv, Ta, L, pad_idx, copies, vm.data_ptr(), am.data_ptr(),
                                            cm.data_ptr(), trg_in.data_ptr(), trg_y.data_ptr(), _p(bump64), _p(b32[0]), _p(b32[1]),
                                            stream()), "bmhrl_batch_head")
    return vm, am, cm, trg_in, trg_y


# ---- Conv1d('same') + GroupNorm input projection of the DETR-mode agent (csrc/conv_gn.hip)
def unfold1d_bf16(x, out, ldo, B, T, C, k, left):
    """out[(b, t)][j * C + c] = x[b][t + j - left][c] (0 outside the clip), bf16: the operand of a Conv1d as a GEMM"""
    _need_cuda(x, out)
    _lib.check(_lib.load().bmhrl_unfold1d_bf16(x.data_ptr(), out.data_ptr(), ldo, B, T, C, k, left, stream()), "bmhrl_unfold1d_bf16")


def fold1d(du, ldu, dx, B, T, C, k, left):
    """dx[b][t][c] = sum_j du[(b, t - j + left)][j * C + c]: the data gradient of the unfolded operand"""
    _need_cuda(du, dx)
    _lib.check(_lib.load().bmhrl_fold1d(du.data_ptr(), ldu, dx.data_ptr(), B, T, C, k, left, stream()), "bmhrl_fold1d")


def groupnorm_fwd(x, gamma, beta, y, mean, rstd, B, T, C, G, eps):
    _need_cuda(x, y)
    _lib.check(_lib.load().bmhrl_groupnorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), B, T, C, G, float(eps), stream()), "bmhrl_groupnorm_fwd")


def groupnorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, B, T, C, G):
    _need_cuda(dy, x, dx)
    _lib.check(_lib.load().bmhrl_groupnorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx.data_ptr(), _p(dgamma), _p(dbeta), B, T, C, G, stream()), "bmhrl_groupnorm_bwd")
