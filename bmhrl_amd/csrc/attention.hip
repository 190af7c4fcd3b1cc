// Head-dimension-256 attention forward (bmhrl_attention_fwd: video self attention, A<-V cross attention), the softmax
// statistics helpers of the attention backward and the row softmax of the materialised small-Sq path.  The kernel template
// lives in attention_fwd.h; the head-dimension-128 (absorbed-projection) entry point is attention128.hip.
#include "attention_fwd.h"

void bmhrl_attn128_set_cfg(int code);   // attention128.hip
// attention_fwd_sk256.hip: the two-phase exact-softmax form for memories of at most 256 keys
bool bmhrl_attn256_sk_ok(int B, int H, int Sq, int Sk, long mask_sq, bool force);
int bmhrl_attn256_sk_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo,
                         float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb, int32_t B, int32_t H, int32_t Sq,
                         int32_t Sk, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream);

namespace {

// delta[b,h,q] = scale * sum_d dO[b,q,h,d] * O[b,q,h,d].  RPW (b,q,h) rows per wave, 64 / RPW lanes each, 16 bytes per lane and
// step: a head of 128 (256) columns keeps every lane busy with RPW = 4 (2) instead of 16 (32) of 64.
template <int RPW>
__global__ void attn_delta_kernel(const bf16_t* __restrict__ dO, long lddo, const bf16_t* __restrict__ O, long ldo,
                                  float* __restrict__ delta, float scale, int B, int H, int Sq, int dk) {
  constexpr int LPR = 64 / RPW;                       // lanes per row
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
  const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long total = (long)B * Sq * H;
  const long row = wid * RPW + sub;
  float acc = 0.f;
  if (row < total) {
    const int hd = row % H;
    const long bq = row / H;
    const bf16_t* a = dO + bq * lddo + (long)hd * dk;
    const bf16_t* c = O + bq * ldo + (long)hd * dk;
    for (int d = l * 8; d < dk; d += LPR * 8) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + d);
      const bf16x8 y = *reinterpret_cast<const bf16x8*>(c + d);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)x[j] * (float)y[j];
    }
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);   // stays inside the row's lane group
  if (l == 0 && row < total) {
    const int hd = row % H;
    const long bq = row / H;
    const long b = bq / Sq, q = bq % Sq;
    delta[(b * H + hd) * Sq + q] = acc * scale;
  }
}

// Row softmax for the materialised small-Sq path: one wave per row, fp32 in, bf16 out.
// (rpg > 0: the bf16 rows come in groups of rpg consecutive rows, group g at g * pgs elements -- P and dS of the paired
// memory attentions share a buffer, interleaved per query: functional.PairMemAttnFn)
__device__ __forceinline__ long grouped_row(long row, long ld, int rpg, long pgs) {
  return rpg > 0 ? (row / rpg) * pgs + (row % rpg) * ld : row * ld;
}
__global__ void softmax_rows_kernel(const float* __restrict__ S, long lds, bf16_t* __restrict__ P, long ldp, long rows,
                                    int cols, int rpg, long pgs) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* s = S + row * lds;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, s[c]);
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane; c < cols; c += 64) l += __expf(s[c] - m);
  l = wave_sum(l);
  const float inv = 1.f / l;
  bf16_t* pr = P + grouped_row(row, ldp, rpg, pgs);
  for (int c = lane; c < cols; c += 64) pr[c] = (bf16_t)(__expf(s[c] - m) * inv);
}

// Row softmax backward for the same path: dS[c] = scale * P[c] * (dP[c] - sum_k P[k] dP[k]), one wave per row.  The row term
// is formed from the SAME bf16 probabilities and fp32 dP the product uses, so the rows of dS sum to (almost) zero whatever
// rounding P carries -- with the row term taken from the rounded context instead (sum_d dO O), the few-query attentions over a
// long memory lost 3 % of their score gradient (30 queries, 256 / 800 keys: dQ' = dS mem is what is left of a cancellation).
__global__ void softmax_bwd_rows_kernel(const bf16_t* __restrict__ P, long ldp, const float* __restrict__ dP, long lddp,
                                        bf16_t* __restrict__ dS, long ldds, long rows, int cols, float scale,
                                        const uint8_t* __restrict__ mask, long mask_sb, long mask_sq, int rows_per_query,
                                        int queries, int rpg, long pgs) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bf16_t* p = P + grouped_row(row, ldp, rpg, pgs);
  const float* g = dP + row * lddp;
  float acc = 0.f;
  for (int c = lane; c < cols; c += 64) acc = fmaf((float)p[c], g[c], acc);
  const float delta = wave_sum(acc);
  bf16_t* o = dS + grouped_row(row, ldds, rpg, pgs);
  // masked_fill: the score of a masked key is a constant, no gradient reaches it (only a fully masked row has P != 0 there)
  const long q = row / rows_per_query;
  const uint8_t* m = mask ? mask + (q / queries) * mask_sb + (q % queries) * mask_sq : nullptr;
  for (int c = lane; c < cols; c += 64) o[c] = (m && !m[c]) ? (bf16_t)0.f : (bf16_t)(scale * (float)p[c] * (g[c] - delta));
}

int g_cfg256 = 0;   // (QW, KW) split: 0 = automatic; bmhrl_attention_config() pins one (tuning aid)

}  // namespace

extern "C" int bmhrl_attention_max_keys(void) { return AttnCfg<256, 4, 1, 3, false>::MAX_SK; }

extern "C" int bmhrl_attention_config(int32_t head_dim, int32_t code) {
  // code = 10 * QW + KW (41 or 22), 0 = the automatic choice
  if (head_dim == 256) g_cfg256 = code;
  else if (head_dim == 128) bmhrl_attn128_set_cfg(code);
  else return -22;
  return 0;
}

extern "C" int bmhrl_attention_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                   void* O, int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask,
                                   int64_t mask_sb, int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                                   int32_t dk, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                   bmhrl_stream_t stream) {
  if (dk == 256 && (g_cfg256 == 0 || g_cfg256 == 256) && bmhrl_attn256_sk_ok(B, H, Sq, Sk, mask_sq, g_cfg256 == 256))   // (code 256: this form)
    return bmhrl_attn256_sk_fwd(Q, ldq, K, ldk, V, ldv, O, ldo, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, dropout_p, seed,
                                seed_dev, (hipStream_t)stream);
  return attention256_entry(Q, ldq, K, ldk, V, ldv, O, ldo, row_max, row_sum, mask, mask_sb, mask_sq, B, H, Sq, Sk, dk, scale,
                            dropout_p, seed, seed_dev, g_cfg256 == 256 ? 0 : g_cfg256, (hipStream_t)stream);
}

extern "C" int bmhrl_attn_delta(const void* dO, int64_t lddo, const void* O, int64_t ldo, float* delta, float scale,
                                int32_t B, int32_t H, int32_t Sq, int32_t dk, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dO && O && delta && dk % 8 == 0 && lddo % 8 == 0 && ldo % 8 == 0);
  const long total = (long)B * Sq * H;
  dim3 block(256);
#define BMHRL_DELTA(RPW_)                                                                                              \
  hipLaunchKernelGGL(attn_delta_kernel<RPW_>, dim3((unsigned)((total + 4 * RPW_ - 1) / (4 * RPW_))), block, 0,          \
                     (hipStream_t)stream, (const bf16_t*)dO, (long)lddo, (const bf16_t*)O, (long)ldo, delta, scale, B, H, Sq, dk)
  if (dk <= 128) BMHRL_DELTA(4);
  else if (dk <= 256) BMHRL_DELTA(2);
  else BMHRL_DELTA(1);
#undef BMHRL_DELTA
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t lddp, void* dS, int64_t ldds,
                                      int64_t rows, int32_t cols, float scale, const uint8_t* mask, int64_t mask_sb,
                                      int64_t mask_sq, int32_t rows_per_query, int32_t queries, int32_t rows_per_group,
                                      int64_t group_stride, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(P && dP && dS && rows > 0 && cols > 0 && ldp >= cols && lddp >= cols && ldds >= cols);
  BMHRL_CHECK_ARG(rows_per_group >= 0 && (rows_per_group == 0 || (ldp == ldds && group_stride >= (int64_t)rows_per_group * ldp)));
  BMHRL_CHECK_ARG(!mask || (rows_per_query > 0 && queries > 0 && rows % ((int64_t)rows_per_query * queries) == 0));
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)P, (long)ldp, dP, (long)lddp,
                     (bf16_t*)dS, (long)ldds, (long)rows, cols, scale, mask, (long)mask_sb, (long)mask_sq,
                     mask ? rows_per_query : 1, mask ? queries : 1, rows_per_group, (long)group_stride);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int64_t rows, int32_t cols,
                                  int32_t rows_per_group, int64_t group_stride, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(S && P && rows > 0 && cols > 0);
  BMHRL_CHECK_ARG(rows_per_group >= 0 && (rows_per_group == 0 || group_stride >= (int64_t)rows_per_group * ldp));
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, (hipStream_t)stream, S, (long)lds, (bf16_t*)P, (long)ldp,
                     (long)rows, cols, rows_per_group, (long)group_stride);
  return hip_status(hipGetLastError());
}
