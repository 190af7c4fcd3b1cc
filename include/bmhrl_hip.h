/*
 * bmhrl_hip.h -- C ABI of the MI355X (gfx950) hot path of Berghojo/bmhrl.
 *
 * The reference has no native code: its hot path is a chain of stock torch ops
 * (model/multihead_attention.py, model/blocks.py, model/bm_hrl_agent.py, loss/,
 * epoch_loops/captioning_bmrl_loops.py).  Each entry point below replaces one such chain
 * and cites it.  All pointers are DEVICE pointers owned by the caller; nothing is
 * allocated or freed inside; every call enqueues work on `stream` and returns
 * immediately (0 = ok, negative = -errno style argument error, positive = hipError_t).
 * No global mutable state: calls are re-entrant across processes (one process per GPU).
 *
 * bf16 tensors are passed as `void*` to 16-bit storage.  Every 2-D bf16 buffer has an
 * explicit leading dimension (elements) that must be a multiple of 8; padding columns
 * must hold finite values (the library never writes them; allocate zeroed).
 */
#ifndef BMHRL_HIP_H
#define BMHRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bmhrl_stream_t; /* hipStream_t */

/* ---------------------------------------------------------------------------------------------
 * Batched MFMA GEMM with fused epilogue:  C[b](M,N) = epilogue( sum_k A[b](m,k) * B[b](k,n) )
 * Replaces nn.Linear forward/backward (model/multihead_attention.py:70-72,89; model/blocks.py:178-184;
 * model/bm_hrl_agent.py:439,464) and the Q.K^T / P.V products of attention() backward
 * (model/multihead_attention.py:13,25).  bf16 operands, fp32 accumulate.
 *   a_trans = 0: A stored [M][K] (k contiguous, leading dim lda)   1: stored [K][M]
 *   b_trans = 0: B stored [N][K] (k contiguous; nn.Linear weight)  1: stored [K][N]
 *   batch index = b1 * batch2 + b2; element offset of operand X = b1 * x_sb1 + b2 * x_sb2.
 * ------------------------------------------------------------------------------------------- */
enum {
  BMHRL_EPI_LINEAR = 0, /* v = alpha*acc (+bias[n]) ; mask==0 -> -1e9 ; relu ; dropout ; (+residual[m][n]) */
  BMHRL_EPI_PROB = 1,   /* p = exp(masked(alpha*acc) - rowvec[m]) / rowvec2[m]        (recompute softmax P) */
  BMHRL_EPI_DSCORE = 2, /* ds = aux[m][n] * (acc - rowvec[m]) * alpha, 0 where mask == 0 (softmax backward) */
  BMHRL_EPI_RELU_BWD = 3 /* dz = aux[m][n] > 0 ? alpha*acc : 0     (ReLU + inverted-dropout backward of the FFN) */
};

typedef struct bmhrl_gemm_desc {
  int32_t M, N, K;
  int32_t batch1, batch2;
  const void* A; int64_t lda, a_sb1, a_sb2; int32_t a_trans;
  const void* B; int64_t ldb, b_sb1, b_sb2; int32_t b_trans;
  float* C;  int64_t ldc, c_sb1, c_sb2;          /* fp32 output, may be NULL */
  void* Cb;  int64_t ldcb, cb_sb1, cb_sb2;       /* bf16 output, may be NULL */
  int32_t epilogue;
  float alpha;
  int32_t relu;
  int32_t accumulate;                             /* C += v instead of C = v (fp32 output only) */
  int32_t allow_split_k;                          /* C is zero-initialised: long reductions may be split (fp32 atomics) */
  const float* bias;                              /* [N] or NULL */
  const float* residual; int64_t ldr, r_sb1, r_sb2; /* fp32 [M][N] or NULL */
  const uint8_t* mask; int64_t mask_sb1, mask_sm; /* byte mask[b1][m*mask_sm + n], mask_sm may be 0 */
  const float* rowvec; const float* rowvec2; int64_t rv_sb1, rv_sb2; /* per-row fp32 vectors: (row max, row sum) or delta */
  const void* aux; int64_t ldaux, aux_sb1, aux_sb2; /* bf16 [M][N] (P for DSCORE) */
  float dropout_p; uint64_t seed;                 /* inverted dropout on v; element id = b1*drop_sb1 + b2*drop_sb2 + m*drop_sm + n */
  int64_t drop_sb1, drop_sb2, drop_sm;            /* (all 0 -> id = (batch*M + m)*N + n) */
  const uint64_t* seed_dev;                       /* optional device word added to seed (changes under graph replay) */
  float* colsum; int64_t colsum_sb2;              /* optional: colsum[b2*colsum_sb2 + n] += sum_m v (fp32 atomics; the bias
                                                     gradient of the layer whose dY this GEMM writes); not with split-K */
  int64_t bias_sb2;                               /* bias of batch entry (b1, b2) starts at bias + b1*bias_sb1 + b2*bias_sb2 (per-head bias slices) */
  int64_t colsum_sb1, bias_sb1;                   /* batch1 strides of colsum / bias (two weight sets in one launch: ABI 8) */
  float* split_ws; int64_t split_ws_elems;        /* optional workspace of >= bmhrl_gemm_splits() * batch1 * batch2 * M * N floats
                                                     (ABI 14): a K split then stores its partial tiles there and a second launch
                                                     adds them in split order -- C needs no zeroing, `accumulate` is allowed, and
                                                     the result is the same from run to run (fp32 atomics are not) */
} bmhrl_gemm_desc;

int bmhrl_gemm(const bmhrl_gemm_desc* d, bmhrl_stream_t stream);
/* n independent problems; up to four at a time run as ONE launch when they are 64 x 64-tile problems of the same operand
 * layout on the register-staged main loop (reductions that are no multiple of 64), any other mix one by one.  Same results
 * as n calls of bmhrl_gemm.  Meant for leaf products (the caption-side weight gradients of a fusion layer's backward). */
int bmhrl_gemm_group(const bmhrl_gemm_desc* descs, int32_t n, bmhrl_stream_t stream);
/* K splits bmhrl_gemm uses for a plain fp32 output of (M, N) over a reduction of K with allow_split_k set and `batch` =
 * batch1 * batch2 (the weight-gradient products): 1 = every element of C is stored exactly once, so C may be uninitialised
 * memory; > 1 = fp32 atomics into a C the caller must have zeroed.  Same decision function as the launcher's.  < 0: bad
 * arguments. */
int bmhrl_gemm_splits(int32_t M, int32_t N, int32_t K, int32_t batch);

/* ---------------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention forward (flash style, S x S never materialised).
 * Replaces attention() + the head split/merge views, model/multihead_attention.py:7-31,75-86:
 *   O[b,q,h,:] = softmax_k( Q[b,q,h,:].K[b,k,h,:] * scale ; mask==0 -> -1e9 ) . V[b,k,h,:]
 * Q,K,V,O are bf16 (B, S, H*DK) row-major with leading dims ldq/ldk/ldv/ldo (heads are column
 * slices, exactly the .view(B,-1,H,d_k) of the reference; base pointers 16-byte aligned, leading
 * dims multiples of 8 elements).  DK must be 256 (the reference's d_model 1024 / H 4).  mask: bytes, mask[b*mask_sb + q*mask_sq + k], mask_sq = 0 for key-padding
 * masks (B,1,Sk).  row_max / row_sum (B,H,Sq) fp32 = softmax statistics of the masked scaled scores, kept
 * separately (not as one log-sum-exp) so that fully masked rows (all scores -1e9) stay exact in backward.
 * dropout_p > 0 applies the reference's dropout on the attention OUTPUT (:27-28).
 * ------------------------------------------------------------------------------------------- */
int bmhrl_attention_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                        void* O, int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                        int64_t mask_sq,
                        int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t DK, float scale,
                        float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream);

/* The same attention in the absorbed-projection form for keys / values that are 128 wide before their projection (the
 * audio stream): scores_h = scale * Qp_h X^T with Qp_h = Q_h Wk_h (B,Sq,H,128), context_h = softmax(scores_h) X --
 * X (B,Sk,128) serves every head as keys and values; Q_h K_h^T differs from Qp_h X^T only by a per-row constant, and
 * P_h V_h = context_h Wv_h^T + bv_h (model/multihead_attention.py:7-31 with K = X Wk^T + bk, V = X Wv^T + bv).
 * ctx (B,Sq,H,128) bf16, normalised, no dropout (it applies after the Wv projection).  mask: (B,Sk) bytes or NULL.
 * row_max / row_sum as bmhrl_attention_fwd.  (Both attention entry points accept Sk up to 10 112.) */
int bmhrl_attention_shared128_fwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo,
                                  float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                                  int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale, bmhrl_stream_t stream);

/* IEEE-half (fp16) operand variants of the two forward entry points: Q / K / V / X / Qp in, O / ctx out are _Float16 instead
 * of bfloat16 (v_mfma_f32_32x32x16_f16; scores, softmax statistics and accumulation stay fp32); every other argument as
 * above.  BASELINE configs[4] names "fp16/bf16 MFMA cross-attention"; the training path of this package uses bf16. */
int bmhrl_attention_fwd_f16(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                            void* O, int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                            int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t DK, float scale,
                            float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream);
int bmhrl_attention_shared128_fwd_f16(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo,
                                      float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                                      int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale, bmhrl_stream_t stream);

/* Fused backward of bmhrl_attention_shared128_fwd (autograd of attention(), model/multihead_attention.py:7-31, in the
 * absorbed-projection form): given dCx (B,Sq,H,128) bf16, the forward's statistics and delta[b,h,q] = sum_d dCx*Cx
 * (bmhrl_attn_delta), writes dQp (B,Sq,H,128) bf16 and -- when dX != NULL -- dX (B,Sk,128) fp32 (leading dim lddx; += when
 * accumulate_dx) = sum over heads of P_h^T dCx_h + dS_h^T Qp_h.  P and dS are recomputed per tile from the statistics and
 * never written to memory; masked keys get no score gradient (masked_fill); deterministic (no atomics).  workspace:
 * bmhrl_attention_shared128_bwd_workspace(B,H,Sk) fp32 elements (the per-head partials of dX), uninitialised is fine. */
int64_t bmhrl_attention_shared128_bwd_workspace(int32_t B, int32_t H, int32_t Sk);
int bmhrl_attention_shared128_bwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, const void* dCx, int64_t lddo,
                                  const float* row_max, const float* row_sum, const float* delta, const uint8_t* mask,
                                  int64_t mask_sb, void* dQp, int64_t lddq, float* dX, int64_t lddx, int32_t accumulate_dx,
                                  float* workspace, int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale,
                                  bmhrl_stream_t stream);

/* Largest Sk the fused attention kernels accept (their key-mask ballots live in LDS); longer memories take the materialised
 * GEMM path of the host code. */
int bmhrl_attention_max_keys(void);

/* Tuning aid: pin the (query blocks x key splits) shape of the attention workgroups -- head_dim 256 or 128; code = 10 * QW + KW
 * (41: 4 x 1, 22: 2 x 2; head_dim 256 also 256: the two-phase exact-softmax kernel for at most 256 keys, which the automatic
 * choice takes when the grid fills the chip; head_dim 128 also 24: 2 x 4 and 14: the r04 form in which one wave carries two heads over four key
 * splits, shapes it does not serve fall back to the automatic choice), 0 = automatic (the default).  Process-wide, not
 * thread-safe; results do not depend on it beyond bf16 rounding of P. */
int bmhrl_attention_config(int32_t head_dim, int32_t code);

/* Attention core of SHORT sequences (Sq, Sk <= 32; d_k in {64, 128, 256, 512}) as one launch per direction: the caption
 * self attention of the fusion layers and the worker's goal attention (model/multihead_attention.py:7-31 on the 30 caption
 * positions).  Q / K / V / O / dO / dQ / dK / dV are bf16 row-major matrices whose row b * S + i holds the heads side by side
 * (head h at columns [h * d_k, (h + 1) * d_k) of the given pointer), P is bf16 (B, H, Sq, ldp) with ldp = Sk rounded up to 8
 * (padding columns written as zero) -- the probabilities the backward needs.  mask: bytes, mask[b * mask_sb + q * mask_sq +
 * key] == 0 -> the score is the reference's -1e9 fill (mask_sq = 0: one key mask per batch row); dropout on the OUTPUT with
 * element id (b * Sq + q) * H * d_k + h * d_k + d, like the context GEMM's epilogue it replaces.  Backward: dO is the gradient
 * of the pre-dropout output; the row term of the softmax is sum_k P dP from the same rounded P; masked keys get no score
 * gradient; dbq / dbk / dbv (optional, fp32 (H * d_k)) receive the column sums of dQ / dK / dV by atomic adds.
 * bmhrl_small_attention_ok: 1 when the shape is served. */
int bmhrl_small_attention_ok(int32_t Sq, int32_t Sk, int32_t dk);
int bmhrl_small_attention_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                              int64_t ldo, void* P, int32_t ldp, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                              int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t dk, float scale, float dropout_p,
                              uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream);
int bmhrl_small_attention_bwd(const void* dO, int64_t lddo, const void* P, int32_t ldp, const void* Q, int64_t ldq,
                              const void* K, int64_t ldk, const void* V, int64_t ldv, void* dQ, int64_t lddq, void* dK,
                              int64_t lddk, void* dV, int64_t lddv, float* dbq, float* dbk, float* dbv, const uint8_t* mask,
                              int64_t mask_sb, int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t dk,
                              float scale, bmhrl_stream_t stream);

/* Row softmax of materialised scores (small-Sq path: caption self/cross attention, goal attention).
 * S fp32 (rows, lds) -> P bf16 (rows, ldp), cols valid columns; also writes nothing else. */
/* Few-query attention over a long memory in the absorbed-projection form (functional.PairMemAttnFn; model/bm_hrl_agent.py:
 * 87-101): L <= 32 queries of a (sample, head) against Sk <= 896 memory rows of width dm (multiple of 32, <= 1024), keys =
 * values = the memory rows.  One launch per direction instead of GEMM + row kernel + GEMM:
 *   forward  (backward = 0): P = softmax(scale X mem^T, -1e9 at masked keys) -> PD slot 0; Y = P mem          (X = Q', Y = context)
 *   backward (backward = 1): dS = scale P (X mem^T - rowsum(P X mem^T)), 0 at masked keys -> PD slot 1; Y = dS mem (X = d context, Y = dQ')
 * X / Y: bf16, row (b2 * L + q) * ld + h * dm; mem: bf16 (n_mem, Sk, dm) row-major, sample b2 reads memory b2 % n_mem; mem_t:
 * the same memory transposed per sample, (n_mem, dm, ldt) with the columns [Sk, ldt) zero and ldt >= Sk rounded up to 16
 * (bmhrl_cast_memory writes both from the fp32 memory in one pass); PD: bf16, element (b2 * L + q) * pd_row + slot * pd_slot +
 * h * pad8(Sk) + key; mask: bytes, b2 * mask_sb + key.  bmhrl_memory_attention_ok: 1 when the shape is served. */
int bmhrl_memory_attention_ok(int32_t L, int32_t Sk, int32_t dm);
int bmhrl_cast_memory(const float* x, void* y, void* y_t, int32_t B, int32_t Sk, int32_t dm, int32_t ldt, bmhrl_stream_t stream);
int bmhrl_memory_attention(int32_t backward, const void* X, int64_t ldx, const void* mem, int64_t mem_sb, const void* mem_t,
                           int64_t mem_t_sb, int32_t ldt, void* PD, int64_t pd_row, int64_t pd_slot, void* Y, int64_t ldy,
                           const uint8_t* mask, int64_t mask_sb, int32_t n_mem, int32_t B2, int32_t H, int32_t L, int32_t Sk,
                           int32_t dm, float scale, bmhrl_stream_t stream);

/* rows_per_group > 0: the bf16 rows (P here; P and dS in the backward) are laid out in groups of rows_per_group consecutive
 * rows, group g starting g * group_stride elements into the buffer (0: plain rows with the leading dimension) */
int bmhrl_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int64_t rows, int32_t cols, int32_t rows_per_group,
                       int64_t group_stride, bmhrl_stream_t stream);

/* Backward of the same row softmax (autograd of F.softmax + masked_fill in attention(), model/multihead_attention.py:22-25):
 * dS[r][c] = scale * P[r][c] * (dP[r][c] - sum_k P[r][k] dP[r][k]), 0 where the key is masked; P bf16 (rows, ldp), dP fp32
 * (rows, lddp) -> dS bf16.  Rows are ordered (sample, query, rows_per_query) with `queries` queries per sample; mask (optional)
 * is addressed mask[sample * mask_sb + query * mask_sq + c] (0 = masked). */
int bmhrl_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t lddp, void* dS, int64_t ldds,
                           int64_t rows, int32_t cols, float scale, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                           int32_t rows_per_query, int32_t queries, int32_t rows_per_group, int64_t group_stride,
                           bmhrl_stream_t stream);

/* Softmax part of the backward of the head-dimension-256 attentions with at most 256 keys (video self attention, A<-V cross
 * attention: autograd of attention(), model/multihead_attention.py:7-31, called at model/bm_hrl_agent.py:358-377) in ONE
 * launch: P = exp(masked scaled scores - row_max) / row_sum from the forward's statistics, dP = dO V^T,
 * delta = sum_k P dP, dS = P (dP - delta) scale (0 at masked keys: no gradient through masked_fill).  Q, K, V, dO are bf16
 * rows with head h at columns [h*256, h*256+256) (leading dims in elements); mask: key mask (B, Sk) bytes or NULL;
 * P and dS: bf16 (B, H, Sq, ldp), ldp = pad8(Sk), padding columns written as 0.  Replaces bmhrl_attn_delta + the PROB and
 * DSCORE GEMM epilogues of bmhrl_gemm for these shapes.  _ok: 1 when the shape is served (dk == 256, Sk <= 256, key mask). */
int bmhrl_attention_bwd_scores256_ok(int32_t Sq, int32_t Sk, int32_t dk, int64_t mask_sq);
int bmhrl_attention_bwd_scores256(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                  const void* dO, int64_t lddo, const float* row_max, const float* row_sum,
                                  const uint8_t* mask, int64_t mask_sb, void* P, void* dS, int64_t ldp, int32_t B, int32_t H,
                                  int32_t Sq, int32_t Sk, float scale, bmhrl_stream_t stream);

/* delta[b,h,q] = scale * sum_d dO[b,q,h,d] * O[b,q,h,d]   (softmax backward row term) */
int bmhrl_attn_delta(const void* dO, int64_t lddo, const void* O, int64_t ldo, float* delta, float scale,
                     int32_t B, int32_t H, int32_t Sq, int32_t DK, bmhrl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (eps 1e-5, affine) -- nn.LayerNorm inside ResidualConnection, model/blocks.py:132,138,
 * and normCA/normCV, model/bm_hrl_agent.py:68-69,107-108.
 *   fwd: x fp32 (rows, D) -> y bf16 (rows, ldy) and/or y32 fp32 (rows, D); saves mean, rstd.
 *   bwd: dx = [dx_add +] LN'(dy) ; dgamma/dbeta accumulated with fp32 atomics into zeroed buffers.
 * ------------------------------------------------------------------------------------------- */
int bmhrl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, int64_t ldy,
                        float* y_f32, float* mean, float* rstd, int64_t rows, int32_t D, bmhrl_stream_t stream);
int bmhrl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        float* dx, const float* dx_add /* optional: dx = dx_add + LN'(dy) */, float* dgamma,
                        float* dbeta, int64_t rows, int32_t D, bmhrl_stream_t stream);
/* Same result; the column sums go block partials -> `workspace` -> dgamma/dbeta instead of every block adding into
 * the same 2*D addresses.  workspace: bmhrl_layernorm_bwd_workspace(rows, D) floats, need not be initialised; with a
 * NULL / too small workspace the call is bmhrl_layernorm_bwd. */
int64_t bmhrl_layernorm_bwd_workspace(int64_t rows, int32_t D);
int bmhrl_layernorm_bwd_ws(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           float* dx, const float* dx_add, float* dgamma, float* dbeta, int64_t rows, int32_t D,
                           float* workspace, int64_t workspace_floats, bmhrl_stream_t stream);

/* `groups` independent LayerNorms of rows_per_group rows each in one launch (the worker and the manager half of a paired
 * fusion block, model/bm_hrl_agent.py:523,528): x, y, mean, rstd, dy, dx are group-major ((groups * rows_per_group, .)),
 * gamma / beta / dgamma / dbeta are (groups, D).  Same arithmetic per row as bmhrl_layernorm_fwd / _bwd. */
int bmhrl_layernorm_fwd_groups(const float* x, const float* gamma, const float* beta, void* y_bf16, int64_t ldy, float* y_f32,
                               float* mean, float* rstd, int64_t rows_per_group, int32_t D, int32_t groups,
                               bmhrl_stream_t stream);
int bmhrl_layernorm_bwd_groups(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                               float* dx, const float* dx_add, float* dgamma, float* dbeta, int64_t rows_per_group, int32_t D,
                               int32_t groups, bmhrl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Feature add + positional encoding (K1): out = a (+ b) + PE[s]  (epoch_loops/captioning_bmrl_loops.py:498,
 * model/blocks.py:105-112).  pe is the precomputed fp32 table (S_max, D).  Optional bf16 copy and dropout.
 * ------------------------------------------------------------------------------------------- */
int bmhrl_add_posenc(const float* a, const float* b, const float* pe, float* out, void* out_bf16, int64_t ldob,
                     int32_t B, int32_t S, int32_t D, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                     bmhrl_stream_t stream);

/* Embedding * sqrt(D) (+ second embedding mix) + PE : model/blocks.py:44-48, model/bm_hrl_agent.py:611-625,642.
 * emb_out (B,L,D) fp32 = un-positional-encoded embeddings (critic input); out = emb_out + PE. */
int bmhrl_embed_posenc(const int64_t* tok, const int64_t* tok2, float mix, const float* table, const float* pe,
                       float* emb_out, float* out, int32_t B, int32_t L, int32_t D, float scale,
                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream);
/* dTable[tok] += scale * (1-mix) * dC ; dTable[tok2] += scale * mix * dC  (fp32 atomics) */
int bmhrl_embed_bwd(const int64_t* tok, const int64_t* tok2, float mix, const float* dC, float* dtable,
                    int32_t B, int32_t L, int32_t D, float scale, bmhrl_stream_t stream);

/* fp32 -> bf16 cast of a (rows, cols) matrix into a padded-leading-dimension buffer, with optional scale
 * and inverted dropout (regenerates the forward mask from seed: element id = row*cols + col). */
int bmhrl_cast_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, float scale,
                    float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream);

/* Split operand of a GEMM that needs more than bf16's 8 mantissa bits (the vocabulary projection feeding log_softmax,
 * model/bm_hrl_agent.py:463-466,483-484): hi = bf16(x), lo = bf16(x - hi); y gets three column blocks of width `part`:
 * block 0 = hi, block lo_slot (1 or 2) = lo, the other block = hi.  Activations use lo_slot 2, weights lo_slot 1, so that one
 * GEMM with K = 3 * part yields x_hi W_hi + x_hi W_lo + x_lo W_hi.  Padding columns (cols .. part) are not written. */
/* x2 (optional): a second source whose cols2 columns follow x's cols in every block (the concatenated operand of the
 * vocabulary head, cat[x, goal completion], in one launch) */
int bmhrl_cast_split3_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part, int32_t lo_slot, int64_t rows,
                           int32_t cols, const float* x2, int64_t ldx2, int32_t cols2, bmhrl_stream_t stream);

/* bmhrl_cast_bf16 + column sums of the rounded result in one pass: colsum[n] += sum_m y[m][n] (fp32 atomics; colsum is
 * accumulated into, the caller zeroes it).  dY cast and bias gradient of one layer -- nn.Linear's db of
 * model/multihead_attention.py:53-56 / model/blocks.py:181-182 under autograd. */
int bmhrl_cast_colsum_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, float scale,
                           float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* colsum, bmhrl_stream_t stream);
/* the same over `rows / group_rows` row groups, group g adding into colsum + g * colsum_stride (the dY of two layers in one
 * buffer); rows % group_rows == 0.  Dropout element ids run over all rows, as in the one-group call. */
int bmhrl_cast_colsum_bf16_groups(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, float scale,
                                  float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* colsum,
                                  int64_t group_rows, int64_t colsum_stride, bmhrl_stream_t stream);

/* One launch for many casts: the per-step refresh of every bf16 weight shadow / concatenated bias (what the reference
 * gets for free by computing in fp32: model/multihead_attention.py:53-56, model/blocks.py:181-182 hold fp32 weights).
 * `segments` (device memory) holds 6 int64 per segment: source address (fp32), destination address, rows, cols,
 * destination leading dimension in elements (bf16 destination) or <= 0 (fp32 destination, packed), and the index of the
 * segment's first block; a block covers 4096 consecutive elements, n_blocks = sum over segments of
 * ceil(rows*cols / 4096). */
int bmhrl_cast_segments(const int64_t* segments, int32_t n_segments, int32_t n_blocks, bmhrl_stream_t stream);

/* db[n] (+)= sum_m dY[m][n]  (bias gradient of nn.Linear), dY bf16 */
int bmhrl_colsum_bf16(const void* dY, int64_t ld, float* db, int32_t accumulate, int64_t rows, int32_t cols,
                      bmhrl_stream_t stream);
/* the same for `groups` row groups laid out back to back ((groups * rows_per_group, ld)); the sums of group g are ADDED at
 * db + g * db_stride (zero them first).  One launch. */
int bmhrl_colsum_bf16_groups(const void* dY, int64_t ld, float* db, int64_t rows_per_group, int32_t cols, int32_t groups,
                             int64_t db_stride, bmhrl_stream_t stream);
/* y[c] = bf16(x) for c in [0, copies): copy c starts copy_stride elements after copy c - 1 (no scale, no dropout) */
int bmhrl_cast_bf16_copies(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, int32_t copies,
                           int64_t copy_stride, bmhrl_stream_t stream);

/* Fusion gate (K7): out = g*Cv + (1-g)*Ca, g = sigmoid(clamp(a,-2,2)), model/bm_hrl_agent.py:111-114 */
int bmhrl_gate_fwd(const float* cv, const float* ca, const float* a_v, float* out, void* out_bf16, int64_t ldob,
                   int64_t rows, int32_t D, bmhrl_stream_t stream);
int bmhrl_gate_bwd(const float* dout, const float* cv, const float* ca, const float* a_v, float* dcv, float* dca,
                   float* da_v /* atomic += */, int64_t rows, int32_t D, bmhrl_stream_t stream);

/* The tail of a BMFusionLayer in one launch (model/bm_hrl_agent.py:107-114): out = g * normCV(cv) + (1 - g) * normCA(ca),
 * g = sigmoid(clamp(a_v, -2, 2)); rows are n_groups (1 or 2) groups of rows_per_group rows, group i with its own parameters
 * (the worker and the manager stack advance together).  stats: (4, rows) fp32 out = mean / rstd of ca, mean / rstd of cv.
 * bwd: dca / dcv (rows, D); the parameter gradients are ADDED (fp32 atomics) into the group's d* pointers (NULL: skipped),
 * which the caller zeroes.  D <= 512. */
typedef struct bmhrl_fusion_tail_params {
  const float* gamma_ca; const float* beta_ca; const float* gamma_cv; const float* beta_cv; const float* a_v;
  float* dgamma_ca; float* dbeta_ca; float* dgamma_cv; float* dbeta_cv; float* da_v;          /* backward only */
} bmhrl_fusion_tail_params;
/* out_bf16 (optional, row stride ldob >= D): the result in bf16 as well -- the operand of the blocks that consume it */
int bmhrl_fusion_tail_fwd(const float* ca, const float* cv, const bmhrl_fusion_tail_params* groups, int32_t n_groups,
                          int64_t rows_per_group, int32_t D, float* out, float* stats, void* out_bf16, int64_t ldob,
                          bmhrl_stream_t stream);
/* dout: the incoming gradient of group 0 (row stride ldd0, <= 0: D); dout1: of group 1 (row stride ldd1; NULL: the rows
 * behind group 0's in the same tensor) -- the two stacks' outputs are consumed by different heads */
int bmhrl_fusion_tail_bwd(const float* dout, const float* dout1, int64_t ldd0, int64_t ldd1, const float* ca, const float* cv,
                          const float* stats, const bmhrl_fusion_tail_params* groups, int32_t n_groups, int64_t rows_per_group,
                          int32_t D, float* dca, float* dcv, bmhrl_stream_t stream);

/* Manager.expand_goals (K8), model/bm_hrl_agent.py:415-429: src[row] = row of goals_in copied to (b,l), or -1 = zero.
 * bmhrl_expand_goals_index builds the (B*L) int32 source map from the segment labels with the reference's
 * row-transition quirks; fwd is a gather, bwd a scatter-add. */
int bmhrl_expand_goals_index(const int32_t* seg, int32_t* src, int32_t B, int32_t L, bmhrl_stream_t stream);
/* bmhrl_expand_goals_index + bmhrl_gather_rows as one launch (Manager.expand_goals, model/bm_hrl_agent.py:415-429): out[b, l] =
 * x[src[b, l]] (0 where src = -1), `src` is written for the backward's scatter; out_bf16: optional bf16 copy.  L <= 1024. */
int bmhrl_expand_goals(const int32_t* seg, const float* x, int32_t* src, float* out, void* out_bf16, int64_t ldob, int32_t B,
                       int32_t L, int32_t D, bmhrl_stream_t stream);
/* The same with the manager's exploration noise (Manager.forward, model/bm_hrl_agent.py:444-452; on after the constructor and
 * left on by teach_warmstart / teach_manager, :572-589): ONE (D,) vector noise[c] = z_c * std' + 0.5 * mean', z ~ N(0, 1),
 * mean' = nanmean(x) / mean_factor, std' = sqrt(nanmean(|x - nanmean(x)|^2)) / std_factor over all B*L*D entries of x, added
 * to every goal before the segment copy (rows the copy zeroes stay zero).  z comes from the library's counter RNG over
 * (seed + seed_dev[0], column): reproducible, and it changes between replays of a captured step.  noise_out: optional (D,)
 * copy of the vector.  The noise is detached in the reference, so the backward is the plain scatter along `src`.
 * B*L*D < 2^24, D <= 1024. */
int bmhrl_expand_goals_explore(const int32_t* seg, const float* x, int32_t* src, float* out, void* out_bf16, int64_t ldob,
                               int32_t B, int32_t L, int32_t D, float mean_factor, float std_factor, uint64_t seed,
                               const uint64_t* seed_dev, float* noise_out, bmhrl_stream_t stream);
int bmhrl_gather_rows(const float* x, const int32_t* src, float* out, void* out_bf16, int64_t ldob, int64_t rows,
                      int32_t D, bmhrl_stream_t stream);
/* backward of the gather along a row map of bmhrl_expand_goals[_index]: dx[s] = sum of dout[r] over src[r] == s, every row of dx
 * written.  The map's structure is used (a run of consecutive rows points at its LAST row, other rows at themselves or at -1):
 * the sum runs in row order, no atomics -- the manager's goal gradient is the same from run to run. */
int bmhrl_scatter_add_rows(const float* dout, const int32_t* src, float* dx, int64_t rows, int32_t D,
                           bmhrl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-token loss step (K9-K11): log-softmax over the vocabulary fused with the loss.
 *  bmhrl_log_softmax:   logits fp32 (rows, V) -> log-probs in place (WorkerCore, model/bm_hrl_agent.py:463-466)
 *  bmhrl_smooth_kl_fwd: per-row sum of LabelSmoothing (biased_trg == NULL; loss/label_smoothing.py:12-32) or
 *                       BiasedKL (loss/biased_kl.py:22-53) over the unreduced (rows, V) divergence, never
 *                       materialising the target distribution; amp = clamp(score*p(a)*n_row, 0, 1) is computed
 *                       here (epoch_loops/captioning_bmrl_loops.py:285,321-322,409-416).
 *  bmhrl_smooth_kl_bwd: d(loss_scale * sum)/d logits (wrt_logits = 1, log-softmax folded in) or /d log-probs
 *                       (wrt_logits = 0) -> bf16 (rows, ldg) and/or fp32 (rows, V), incl. the path through amp.
 *  bmhrl_log_softmax_bwd: dlogits = dlogp - exp(logp) * rowsum(dlogp) -> bf16 (rows, ldg)
 *  zero_pad_rows: the reference's `idx.sum() > 0` guard: 1/0 forces it, -1 lets the kernel evaluate it on the device.
 * ------------------------------------------------------------------------------------------- */
int bmhrl_log_softmax(float* logits, int64_t ld, int64_t rows, int32_t V, bmhrl_stream_t stream);
int bmhrl_smooth_kl_fwd(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg,
                        const float* score, const float* n_row, float smoothing, int32_t pad_idx,
                        int32_t zero_pad_rows, float* row_loss, float* amp_out, int64_t rows, int32_t V,
                        bmhrl_stream_t stream);
/* the unreduced divergence itself, (rows, V) fp32 -- the reference criteria's return value (callers that inspect entries) */
int bmhrl_smooth_kl_full(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg, const float* score,
                         const float* n_row, float smoothing, int32_t pad_idx, int32_t zero_pad_rows, float* out,
                         int64_t rows, int32_t V, bmhrl_stream_t stream);
int bmhrl_smooth_kl_bwd(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg,
                        const float* score, const float* n_row, float smoothing, int32_t pad_idx,
                        int32_t zero_pad_rows, const float* loss_scale /* device scalar */,
                        const float* loss_scale2 /* second device scalar multiplied in (the incoming d loss), or NULL */,
                        int32_t wrt_logits, void* grad_bf16, int64_t ldg, float* grad_f32, int64_t rows, int32_t V,
                        bmhrl_stream_t stream);
/* The loops' reduction of the row sums: loss = weight * sum(row_loss) / (n_tokens * factor), n_tokens = #(trg != pad_idx)
 * (epoch_loops/captioning_bmrl_loops.py:1156-1158: factor 1; :829-833,846-862: factor 4/20); scale = weight / (n_tokens *
 * factor) is what bmhrl_smooth_kl_bwd takes as loss_scale.  weight: optional device scalar (data-parallel token weight). */
int bmhrl_token_loss_reduce(const float* row_loss, const int64_t* trg, int64_t rows, int64_t pad_idx, const float* weight,
                            float factor, float* loss, float* scale, bmhrl_stream_t stream);
/* The warmstart loss tail as one launch (r03): log_softmax in place over `logits` (model/bm_hrl_agent.py:463-466), the
 * LabelSmoothing row sums (loss/label_smoothing.py:12-32), loss_scale[0] = weight * sum(rows) / (n_tokens * factor) and
 * loss_scale[1] = weight / (n_tokens * factor) (epoch_loops/captioning_bmrl_loops.py:1156-1158), and d loss / d logits as
 * bf16 (loss.backward() of the loops, through the log-softmax) for the gradient `dloss` (device scalar or NULL = 1) the
 * caller will pass to backward.  V % 4 == 0, V <= 12288.  `counter`: four zero-initialised 32-bit words (8-byte aligned)
 * the kernel re-arms: the rows are summed as 2^-32 fixed point with integer atomics (order independent: deterministic), the
 * block that arrives last writes loss_scale.  Same values as bmhrl_log_softmax + bmhrl_smooth_kl_fwd +
 * bmhrl_token_loss_reduce + bmhrl_smooth_kl_bwd(wrt_logits) up to the rounding of the loss's sums. */
int bmhrl_head_loss(float* logits, int64_t ld, const int64_t* trg, float smoothing, int32_t pad_idx, const float* weight,
                    float factor, const float* dloss, float* row_loss, float* loss_scale, void* dlogits_bf16, int64_t ldg,
                    uint32_t* counter, int64_t rows, int32_t V, bmhrl_stream_t stream);
int bmhrl_log_softmax_bwd(const float* dlogp, const float* logp, int64_t ld, void* dlogits_bf16, int64_t ldg,
                          int64_t rows, int32_t V, bmhrl_stream_t stream);
/* out[row] = d rowloss / d log(raw amplitude) (0 where clamp(., 0, 1) is active): the share of the gradient that reaches the
 * row through its amplitude.  Manager branch of biased_kl(), epoch_loops/captioning_bmrl_loops.py:299-317: the amplitude holds
 * the product of p(a) over a segment, so this also flows to the segment's other tokens (bmhrl_amd/functional.py ManagerKLFn). */
int bmhrl_smooth_kl_amp_grad(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg, const float* score,
                             const float* n_row, float smoothing, int32_t pad_idx, int32_t zero_pad_rows, float* out,
                             int64_t rows, int32_t V, bmhrl_stream_t stream);
/* a ~ Categorical(exp(logp)) by inverse CDF with one uniform per row (counter RNG: seed (+ *seed_dev when given: a device
 * word advanced per step, so that a captured step draws fresh samples at every replay), row_offset + row -- row_offset = the
 * first row of this rank's share of the global batch, so that data-parallel ranks draw independent samples);
 * greedy != 0 -> argmax.  epoch_loops/captioning_bmrl_loops.py:283-284 */
int bmhrl_sample_tokens(const float* logp, int64_t ld, int64_t* out, float* p_out, int64_t rows, int32_t V,
                        int32_t greedy, uint64_t seed, const uint64_t* seed_dev, int64_t row_offset, bmhrl_stream_t stream);
/* Reinforce (loss/biased_kl.py:69-81): per-row terms of -adv*log(clamp(p(a), 1e-5, 1-1e-5)) and adv^2, adv = value -
 * critic_value; `pred` holds log-probs (is_logp = 1) or probabilities (0, the reference's input).  The backward
 * writes d(gscale * (mean(policy) + mean(value terms))) w.r.t. the (rows, V) probabilities, value and critic_value. */
int bmhrl_reinforce_fwd(const float* pred, int64_t ld, int32_t is_logp, const int64_t* action, const float* value,
                        const float* critic_value, float* row_policy, float* row_value, int64_t rows, int32_t V,
                        bmhrl_stream_t stream);
int bmhrl_reinforce_bwd(const float* probs, int64_t ld, const int64_t* action, const float* value,
                        const float* critic_value, const float* gscale /* device scalar */, float* dprobs, float* dvalue,
                        float* dcritic, int64_t rows, int32_t V, bmhrl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Frozen segment critic (K15; SegmentCritic.forward, model/bm_hrl_agent.py:204-215), fp32 throughout because its
 * output is thresholded into integer segment labels (:638-640).
 *  bmhrl_gemm_f32   : C[M,N] = A[M,K] W[N,K]^T + bias1 + bias2 on the f32-input MFMA (input projections W_ih x + b)
 *  bmhrl_rnn_step   : one time step of one LSTM (gates = 4, order i,f,g,o) or GRU (gates = 3, order r,z,n) layer:
 *                     xproj (B*L, gates*H) row b*L + t, W_hh (gates*H, H), b_hh (GRU only), state ping-pong buffers
 *                     (B, H); writes h_t (optionally through AReLU, :13-23) to seq_out (B*L, H)
 *  bmhrl_critic_head: score = lin(x), labels = sigmoid(score) > threshold
 * ------------------------------------------------------------------------------------------- */
int bmhrl_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias1, const float* bias2,
                   float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, bmhrl_stream_t stream);
int bmhrl_rnn_step(int32_t gates, const float* xproj, const float* whh, const float* bhh, const float* h_prev,
                   const float* c_prev, float* h_out, float* c_out, float* seq_out, const float* arelu_alpha,
                   const float* arelu_beta, int32_t B, int32_t L, int32_t H, int32_t t, bmhrl_stream_t stream);
/* The whole recurrent stack as a wavefront over (layer, time): cell (l, t) runs in launch l + t (chunk = 1), so the stack takes
 * L + n_layers - 1 launches (all cells of a diagonal in one grid) instead of n_layers * (1 + L).  Layer l reads
 * in_seq (B, L, in_ld) -- the embeddings for l = 0, seq_out of layer l - 1 otherwise -- and writes seq_out (B, L, H)
 * (through AReLU when arelu_alpha / arelu_beta are given: the last layer of the LSTM and of the GRU stack,
 * model/bm_hrl_agent.py:209-213); h / c are two (B, H) state buffers each (c unused for the GRU).  fp32 throughout.
 * With chunk = T > 1 cell (l, t) runs in launch t + T*l (L + T*(n_layers-1) launches). */
typedef struct bmhrl_rnn_layer {
  const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;   /* nn.LSTM / nn.GRU parameter layout */
  const float* in_seq; int64_t in_ld; int32_t in_dim; int32_t gates;            /* gates: 4 = LSTM (i,f,g,o), 3 = GRU (r,z,n) */
  float* seq_out; float* h[2]; float* c[2];
  const float* arelu_alpha; const float* arelu_beta;
  float* xproj;                                           /* (B, L, gates*H) scratch, needed when chunk > 1 */
} bmhrl_rnn_layer;
int bmhrl_rnn_wavefront(const bmhrl_rnn_layer* layers, int32_t n_layers, int32_t B, int32_t L, int32_t H,
                        int32_t chunk /* time steps a layer trails the one below: W_ih is read once per chunk; 1 = the
                                         plain diagonal, which runs the matrix-pipe cell (v_mfma_f32_16x16x4_f32) */,
                        bmhrl_stream_t stream);
int bmhrl_critic_head(const float* x, const float* w, const float* b, float threshold, float* score, int32_t* labels,
                      int64_t rows, int32_t H, bmhrl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser (K14): torch.optim.Adam semantics (scripts/train_rl_captioning_module.py:81-83, default
 * betas/eps, L2 weight decay added to the gradient) over one flat fp32 bucket; grad_scale multiplies
 * the gradient first (1/world for data parallel averaging).
 * ------------------------------------------------------------------------------------------- */
int bmhrl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int32_t step,
                    const int32_t* step_dev /* optional: step count read on the device (graph replay) */,
                    float grad_scale, bmhrl_stream_t stream);

/* The same update driven by a per-parameter table, writing each updated weight's bf16 shadow (or fp32 copy) in the same
 * pass.  segments: int64 (n_segments, 7) on the device = {offset of the parameter in the flat bucket (elements), shadow
 * address (0: none), rows, cols, shadow leading dimension in elements (<= 0: fp32 copy, tightly packed), first block,
 * address of the parameter's fp32 gradient (0: grad + offset, the flat bucket)};
 * a block owns 4096 consecutive elements of one parameter, n_blocks = sum over parameters of ceil(rows * cols / 4096). */
int bmhrl_adam_segments(const int64_t* segments, int32_t n_segments, int32_t n_blocks, float* param, const float* grad,
                        float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int32_t step, const int32_t* step_dev, float grad_scale, bmhrl_stream_t stream);

/* The masks of a bimodal step in one launch (model/masking.py:18-55): v_mask[b][t] = rgb[b][t][0] != 0 (row stride ld_rgb
 * elements), a_mask likewise from audio, c_mask[b][i][j] = trg[b][j] != pad_idx && j <= i; bytes 0 / 1, each mask written
 * `copies` times back to back ((copies, B, .) layout). */
int bmhrl_make_masks(const float* rgb, int64_t ld_rgb, const float* audio, int64_t ld_aud, const int64_t* trg, int32_t B,
                     int32_t Tv, int32_t Ta, int32_t L, int64_t pad_idx, int32_t copies, uint8_t* v_mask, uint8_t* a_mask,
                     uint8_t* c_mask, bmhrl_stream_t stream);

/* Head of a training step in one launch (epoch_loops/captioning_bmrl_loops.py:487-508 feature_getter: the input / target
 * shift of the captions; model/masking.py:28-55: the masks built from the SHIFTED input): captions (B, L + 1) int64 with row
 * stride ld_cap -> trg_in = captions[:, :-1], trg_y = captions[:, 1:] (contiguous (B, L)), the three masks of
 * bmhrl_make_masks from trg_in, and the step's device counters advanced by one: bump64 (the dropout / sampling seed word)
 * and bump32[0..1] (Adam step counters); each may be NULL. */
int bmhrl_batch_head(const float* rgb, int64_t ld_rgb, const float* audio, int64_t ld_aud, const int64_t* captions,
                     int64_t ld_cap, int32_t B, int32_t Tv, int32_t Ta, int32_t L, int64_t pad_idx, int32_t copies,
                     uint8_t* v_mask, uint8_t* a_mask, uint8_t* c_mask, int64_t* trg_in, int64_t* trg_y, int64_t* bump64,
                     int32_t* bump32_a, int32_t* bump32_b, bmhrl_stream_t stream);

/* Library self-description: returns the gfx target the kernels were built for ("gfx950"). */
const char* bmhrl_hip_arch(void);
/* ---------------------------------------------------------------------------------------------
 * Conv1d(padding='same') + GroupNorm input projection of the DETR-mode agent (model/det_bmhrl_agent.py:79-86,169-174),
 * activations (B, T, C) row major.  The convolution itself is bmhrl_gemm over the unfolded operand:
 *   bmhrl_unfold1d_bf16: out[(b, t)][j*C + c] = x[b][t + j - left][c] (0 outside the clip), bf16; 'same' padding: left = (k-1)/2
 *   bmhrl_fold1d:        dx[b][t][c] = sum_j dU[(b, t - j + left)][j*C + c]      (the data gradient; ordered, no atomics)
 *   bmhrl_groupnorm_fwd / _bwd: nn.GroupNorm(G, C) over (T, C/G) per sample and group; mean / rstd (B*G) saved for the
 *   backward; dgamma / dbeta are ADDED to (zero them first).
 * ------------------------------------------------------------------------------------------- */
int bmhrl_unfold1d_bf16(const float* x, void* out, int64_t ldo, int32_t B, int32_t T, int32_t C, int32_t k, int32_t left,
                        bmhrl_stream_t stream);
int bmhrl_fold1d(const float* du, int64_t ldu, float* dx, int32_t B, int32_t T, int32_t C, int32_t k, int32_t left,
                 bmhrl_stream_t stream);
int bmhrl_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int32_t B,
                        int32_t T, int32_t C, int32_t G, float eps, bmhrl_stream_t stream);
int bmhrl_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                        float* dgamma, float* dbeta, int32_t B, int32_t T, int32_t C, int32_t G, bmhrl_stream_t stream);

int bmhrl_hip_abi_version(void);
/* 1 when BMHRL_DETERMINISTIC selects the ordered sums (read once, by the library; atoi(value) != 0).  The host side asks
 * here instead of parsing the variable itself, so both sides always agree. */
int bmhrl_deterministic_enabled(void);

#ifdef __cplusplus
}
#endif
#endif /* BMHRL_HIP_H */
