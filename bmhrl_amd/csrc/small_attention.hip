// Attention core for SHORT sequences (Sq, Sk <= 32) as ONE launch per direction -- the caption self attention of the fusion
// layers and the worker's goal attention (model/multihead_attention.py:7-31 on 30 caption positions; model/bm_hrl_agent.py:
// 73-117, 480-487).  On the batched-GEMM path these are score GEMM + row softmax + context GEMM forward and
// row term + dS GEMM + three gradient GEMMs backward: eight dependent launches of a few microseconds of work each on the
// step's serial chain.  Here a workgroup (4 waves) owns one (batch row, head):
//
//   forward : Q, K, V go to LDS row-major (V^T fragments come out of ds_read_b64_tr_b16); every wave forms S^T = K Q^T (32 x 32 x d_k, the key rows fed in a
//             permuted order so that the accumulator layout of S^T IS the B-operand layout of the next product), softmax over
//             the 16 registers + one cross-half exchange, P (bf16, the tensor the backward needs) stored by wave 0, then the
//             waves split the d_k columns of O^T = V^T P^T, apply the output dropout (same element ids as the GEMM epilogue
//             it replaces) and the rows leave through an LDS image as whole 16-byte pieces.
//   backward: dP^T = V dO^T in the same layout as P^T; delta = sum_k P dP from the SAME rounded P (attention.hip:
//             softmax_bwd_rows_kernel), dS^T = scale P (dP - delta), zero at masked keys (masked_fill passes no gradient);
//             dQ^T = K^T dS^T takes dS^T straight from the registers, dV^T = dO^T P and dK^T = Q^T dS take P^T / dS^T from small
//             LDS copies; the transposed operands (dO^T, K^T, Q^T) are transposed reads of the row-major images; optional
//             bias gradients (column sums of dQ / dK / dV) by one atomic per column.
//
// Roofline: none worth the name -- 128 workgroups of ~2 MFLOP; the launch is bound by its dependent load -> LDS -> MFMA chain
// (~5 us), which is the point: it replaces ~45 us of launches.
#include "../../include/bmhrl_hip.h"
#include "common.h"

namespace {

constexpr int SA_PAD = 8;        // row padding (elements) of the images read along their rows (ds_read_b128: conflict free)
constexpr int SA_TPAD = 32;      // ... of the images read transposed (the 4 rows of a ds_read_b64_tr_b16 block on 4 bank quarters)
constexpr int SA_TROW = 40;      // row length (elements) of the P^T / dS^T images: 32 queries + padding

struct SmallAttnArgs {
  const bf16_t* Q; long ldq;
  const bf16_t* K; long ldk;
  const bf16_t* V; long ldv;
  bf16_t* O; long ldo;
  bf16_t* P; int ldp;                       // (B, H, Sq, ldp)
  const uint8_t* mask; long mask_sb, mask_sq;
  int B, H, Sq, Sk;
  float scale, dropout_p;
  uint64_t seed; const uint64_t* seed_dev;
  // backward
  const bf16_t* dO; long lddo;
  bf16_t* dQ; long lddq;
  bf16_t* dK; long lddk;
  bf16_t* dV; long lddv;
  float* dbq; float* dbk; float* dbv;
};

// accumulator register r of lane half hh <-> key (forward) fed as row (r & 3) + 8 (r >> 2) + 4 hh of the A operand
__device__ __forceinline__ int key_of_row(int rho) {
  return 8 * ((rho >> 2) & 1) + (rho & 3) + 4 * ((rho >> 3) & 1) + 16 * ((rho >> 4) & 1);
}

// rows [0, nrows) x DK columns of a row-major bf16 matrix (row stride ld) -> LDS image with row stride DK + SA_PAD; rows
// >= nrows (up to 32) are zero
template <int DK, int PAD>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ g, long ld, int nrows, bf16_t* s) {
  constexpr int CPR = DK / 8;
  for (int c = threadIdx.x; c < 32 * CPR; c += 256) {
    const int row = c / CPR, ch = c % CPR;
    bf16x8 v = zero_bf16x8();
    if (row < nrows) v = *reinterpret_cast<const bf16x8*>(g + (long)row * ld + ch * 8);
    *reinterpret_cast<bf16x8*>(s + row * (DK + PAD) + ch * 8) = v;
  }
}
// A-operand fragment of X^T (rows d0 + (lane & 31), eight consecutive rows of X from row k0 + 8 (lane >> 5)) out of the
// row-major image X[row][d] with row stride LD: two transposed reads (gemm.hip's pattern)
template <int LD>
__device__ __forceinline__ bf16x8 frag_transposed(const bf16_t* img, int k0, int d0, int lane) {
  const int hh = lane >> 5, g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const bf16_t* base = img + (k0 + 8 * hh + q4) * LD + d0 + 16 * g1 + 4 * p4;
  return join8(lds_read_tr4(base), lds_read_tr4(base + 4 * LD));
}
// LDS image [row][DK + SA_PAD] -> rows [0, nrows) of a row-major global matrix, 16 bytes per thread and step
template <int DK>
__device__ __forceinline__ void store_rows(const bf16_t* s, bf16_t* __restrict__ g, long ld, int nrows) {
  constexpr int CPR = DK / 8;
  for (int c = threadIdx.x; c < 32 * CPR; c += 256) {
    const int row = c / CPR, ch = c % CPR;
    if (row < nrows) *reinterpret_cast<bf16x8*>(g + (long)row * ld + ch * 8) = *reinterpret_cast<const bf16x8*>(s + row * (DK + SA_PAD) + ch * 8);
  }
}
// column sums of the first nrows rows of an LDS image -> one atomic per column
template <int DK>
__device__ __forceinline__ void colsum_rows(const bf16_t* s, int nrows, float* __restrict__ out) {
  for (int d = threadIdx.x; d < DK; d += 256) {
    float a = 0.f;
    for (int r = 0; r < nrows; ++r) a += (float)s[r * (DK + SA_PAD) + d];
    atomicAdd(out + d, a);
  }
}
// accumulator tile (rows d0 + row(r, hh), column = lane & 31) -> image[column][d] as bf16
__device__ __forceinline__ void tile_to_image(const f32x16& acc, bf16_t* img, int ldimg, int d0, int col, int hh) {
#pragma unroll
  for (int r = 0; r < 16; ++r) img[col * ldimg + d0 + (r & 3) + 8 * (r >> 2) + 4 * hh] = (bf16_t)acc[r];
}

template <int DK>
__global__ __launch_bounds__(256) void small_attn_fwd_kernel(const SmallAttnArgs p) {
  constexpr int LDR = DK + SA_PAD;
  constexpr int LDT = DK + SA_TPAD;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[32 * LDR];      // Q rows; later the output image
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[32 * LDT];      // V rows, read transposed
  const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const long D = (long)p.H * DK;
  stage_rows<DK, SA_PAD>(p.Q + (long)b * p.Sq * p.ldq + hd * DK, p.ldq, p.Sq, Qs);
  stage_rows<DK, SA_PAD>(p.K + (long)b * p.Sk * p.ldk + hd * DK, p.ldk, p.Sk, Ks);
  stage_rows<DK, SA_TPAD>(p.V + (long)b * p.Sk * p.ldv + hd * DK, p.ldv, p.Sk, Vs);
  __syncthreads();

  // S^T[key][q]: A rows = keys in the permuted order, B rows = queries
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bf16_t* ka = Ks + key_of_row(r32) * LDR + 8 * hh;
  const bf16_t* qb = Qs + r32 * LDR + 8 * hh;
#pragma unroll 4
  for (int ks = 0; ks < DK / 16; ++ks)
    acc = BMHRL_MFMA16(*reinterpret_cast<const bf16x8*>(ka + 16 * ks), *reinterpret_cast<const bf16x8*>(qb + 16 * ks), acc, 0, 0, 0);
  // this lane: query r32, keys 8 hh + (r & 7) + 16 (r >> 3)
  const int q = r32;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_sb + (long)(q < p.Sq ? q : 0) * p.mask_sq : nullptr;
  float m = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = 8 * hh + (r & 7) + 16 * (r >> 3);
    float s = acc[r] * p.scale;
    if (key >= p.Sk) s = -INFINITY;
    else if (mrow && !mrow[key]) s = NEG_MASK;
    acc[r] = s;
    m = fmaxf(m, s);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[r] = __expf(acc[r] - m);
    l += acc[r];
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  bf16x8 pf[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (bf16_t)(acc[r] * inv);
  if (wave == 0 && q < p.Sq) {
    bf16_t* pr = p.P + (((long)b * p.H + hd) * p.Sq + q) * p.ldp;
    if (8 * hh < p.ldp) *reinterpret_cast<bf16x8*>(pr + 8 * hh) = pf[0];
    if (16 + 8 * hh < p.ldp) *reinterpret_cast<bf16x8*>(pr + 16 + 8 * hh) = pf[1];
  }
  __syncthreads();                                  // every wave is done with Qs: it becomes the output image

  // O^T = V^T P^T, the d-tiles dealt to the waves
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  for (int dt = wave; dt < DK / 32; dt += 4) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = BMHRL_MFMA16(frag_transposed<LDT>(Vs, 0, dt * 32, lane), pf[0], o, 0, 0, 0);
    o = BMHRL_MFMA16(frag_transposed<LDT>(Vs, 16, dt * 32, lane), pf[1], o, 0, 0, 0);
    if (p.dropout_p > 0.f) {
      const uint64_t base = ((uint64_t)b * p.Sq + q) * (uint64_t)D + (uint64_t)hd * DK + dt * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= dropout_scale(p.dropout_p, seed, base + (r & 3) + 8 * (r >> 2));
    }
    tile_to_image(o, Qs, LDR, dt * 32, q, hh);
  }
  __syncthreads();
  store_rows<DK>(Qs, p.O + (long)b * p.Sq * p.ldo + hd * DK, p.ldo, p.Sq);
}

template <int DK>
__global__ __launch_bounds__(256) void small_attn_bwd_kernel(const SmallAttnArgs p) {
  constexpr int LDR = DK + SA_PAD, LDT = DK + SA_TPAD;
  __shared__ __attribute__((aligned(16))) bf16_t Vs[32 * LDR];       // V rows (phase A); then the output image of a gradient
  __shared__ __attribute__((aligned(16))) bf16_t dOs[32 * LDT];      // dO rows: along the rows for dP^T, transposed for dV^T
  __shared__ __attribute__((aligned(16))) bf16_t Ks[32 * LDT];       // K rows, read transposed (dQ^T = K^T dS^T)
  __shared__ __attribute__((aligned(16))) bf16_t Qs[32 * LDT];       // Q rows, read transposed (dK^T = Q^T dS)
  __shared__ __attribute__((aligned(16))) bf16_t Pt[32 * SA_TROW];   // P^T [key][q]
  __shared__ __attribute__((aligned(16))) bf16_t dSt[32 * SA_TROW];  // dS^T [key][q]
  bf16_t* const Out = Vs;
  const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  stage_rows<DK, SA_PAD>(p.V + (long)b * p.Sk * p.ldv + hd * DK, p.ldv, p.Sk, Vs);
  stage_rows<DK, SA_TPAD>(p.dO + (long)b * p.Sq * p.lddo + hd * DK, p.lddo, p.Sq, dOs);
  stage_rows<DK, SA_TPAD>(p.K + (long)b * p.Sk * p.ldk + hd * DK, p.ldk, p.Sk, Ks);
  stage_rows<DK, SA_TPAD>(p.Q + (long)b * p.Sq * p.ldq + hd * DK, p.ldq, p.Sq, Qs);
  __syncthreads();

  // dP^T[key][q] = V dO^T, keys in the permuted order of the forward: same layout as P^T below
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const bf16_t* va = Vs + key_of_row(r32) * LDR + 8 * hh;
    const bf16_t* gb = dOs + r32 * LDT + 8 * hh;
#pragma unroll 4
    for (int ks = 0; ks < DK / 16; ++ks)
      acc = BMHRL_MFMA16(*reinterpret_cast<const bf16x8*>(va + 16 * ks), *reinterpret_cast<const bf16x8*>(gb + 16 * ks), acc, 0, 0, 0);
  }
  const int q = r32;
  bf16x8 pf[2] = {zero_bf16x8(), zero_bf16x8()};
  if (q < p.Sq) {
    const bf16_t* pr = p.P + (((long)b * p.H + hd) * p.Sq + q) * p.ldp;
    if (8 * hh < p.ldp) pf[0] = *reinterpret_cast<const bf16x8*>(pr + 8 * hh);
    if (16 + 8 * hh < p.ldp) pf[1] = *reinterpret_cast<const bf16x8*>(pr + 16 + 8 * hh);
  }
  float delta = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) delta = fmaf((float)pf[r >> 3][r & 7], acc[r], delta);
  delta += __shfl_xor(delta, 32, 64);
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_sb + (long)(q < p.Sq ? q : 0) * p.mask_sq : nullptr;
  bf16x8 sf[2];                                     // dS^T of this lane: the B operand of dQ^T = K^T dS^T
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = 8 * hh + (r & 7) + 16 * (r >> 3);
    float v = p.scale * (float)pf[r >> 3][r & 7] * (acc[r] - delta);
    if (key >= p.Sk || q >= p.Sq || (mrow && !mrow[key])) v = 0.f;
    sf[r >> 3][r & 7] = (bf16_t)v;
  }
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 8 * hh + (r & 7) + 16 * (r >> 3);
      Pt[key * SA_TROW + q] = pf[r >> 3][r & 7];
      dSt[key * SA_TROW + q] = sf[r >> 3][r & 7];
    }
  }
  __syncthreads();                                  // Pt / dSt visible; every wave is done with Vs: it becomes the output image

  // ---- dV^T = dO^T P  (rows d, columns key)
  for (int dt = wave; dt < DK / 32; dt += 4) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const bf16_t* bb = Pt + r32 * SA_TROW + 8 * hh;
    o = BMHRL_MFMA16(frag_transposed<LDT>(dOs, 0, dt * 32, lane), *reinterpret_cast<const bf16x8*>(bb), o, 0, 0, 0);
    o = BMHRL_MFMA16(frag_transposed<LDT>(dOs, 16, dt * 32, lane), *reinterpret_cast<const bf16x8*>(bb + 16), o, 0, 0, 0);
    tile_to_image(o, Out, LDR, dt * 32, r32, hh);
  }
  __syncthreads();
  store_rows<DK>(Out, p.dV + (long)b * p.Sk * p.lddv + hd * DK, p.lddv, p.Sk);
  if (p.dbv) colsum_rows<DK>(Out, p.Sk, p.dbv + hd * DK);
  __syncthreads();

  // ---- dQ^T = K^T dS^T  (rows d, columns q), dS^T from the registers
  for (int dt = wave; dt < DK / 32; dt += 4) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = BMHRL_MFMA16(frag_transposed<LDT>(Ks, 0, dt * 32, lane), sf[0], o, 0, 0, 0);
    o = BMHRL_MFMA16(frag_transposed<LDT>(Ks, 16, dt * 32, lane), sf[1], o, 0, 0, 0);
    tile_to_image(o, Out, LDR, dt * 32, r32, hh);
  }
  __syncthreads();
  store_rows<DK>(Out, p.dQ + (long)b * p.Sq * p.lddq + hd * DK, p.lddq, p.Sq);
  if (p.dbq) colsum_rows<DK>(Out, p.Sq, p.dbq + hd * DK);
  __syncthreads();

  // ---- dK^T = Q^T dS  (rows d, columns key)
  for (int dt = wave; dt < DK / 32; dt += 4) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const bf16_t* bb = dSt + r32 * SA_TROW + 8 * hh;
    o = BMHRL_MFMA16(frag_transposed<LDT>(Qs, 0, dt * 32, lane), *reinterpret_cast<const bf16x8*>(bb), o, 0, 0, 0);
    o = BMHRL_MFMA16(frag_transposed<LDT>(Qs, 16, dt * 32, lane), *reinterpret_cast<const bf16x8*>(bb + 16), o, 0, 0, 0);
    tile_to_image(o, Out, LDR, dt * 32, r32, hh);
  }
  __syncthreads();
  store_rows<DK>(Out, p.dK + (long)b * p.Sk * p.lddk + hd * DK, p.lddk, p.Sk);
  if (p.dbk) colsum_rows<DK>(Out, p.Sk, p.dbk + hd * DK);
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int bmhrl_small_attention_ok(int32_t Sq, int32_t Sk, int32_t dk) {
  return Sq >= 1 && Sq <= 32 && Sk >= 1 && Sk <= 32 && (dk == 64 || dk == 128 || dk == 256 || dk == 512);
}

extern "C" int bmhrl_small_attention_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                         void* O, int64_t ldo, void* P, int32_t ldp, const uint8_t* mask, int64_t mask_sb,
                                         int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t dk, float scale,
                                         float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(Q && K && V && O && P && B > 0 && H > 0 && bmhrl_small_attention_ok(Sq, Sk, dk));
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldp % 8 == 0 && ldp >= Sk && ldp <= 32);
  BMHRL_CHECK_ARG(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(O) && aligned16(P));
  BMHRL_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f && (long)B * H < (1l << 31));
  SmallAttnArgs a = {};
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.O = (bf16_t*)O; a.ldo = ldo; a.P = (bf16_t*)P; a.ldp = ldp;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev;
  dim3 grid((unsigned)(B * H)), block(256);
  switch (dk) {
    case 64: hipLaunchKernelGGL(small_attn_fwd_kernel<64>, grid, block, 0, (hipStream_t)stream, a); break;
    case 128: hipLaunchKernelGGL(small_attn_fwd_kernel<128>, grid, block, 0, (hipStream_t)stream, a); break;
    case 256: hipLaunchKernelGGL(small_attn_fwd_kernel<256>, grid, block, 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(small_attn_fwd_kernel<512>, grid, block, 0, (hipStream_t)stream, a); break;
  }
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_small_attention_bwd(const void* dO, int64_t lddo, const void* P, int32_t ldp, const void* Q, int64_t ldq,
                                         const void* K, int64_t ldk, const void* V, int64_t ldv, void* dQ, int64_t lddq,
                                         void* dK, int64_t lddk, void* dV, int64_t lddv, float* dbq, float* dbk, float* dbv,
                                         const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, int32_t B, int32_t H, int32_t Sq,
                                         int32_t Sk, int32_t dk, float scale, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dO && P && Q && K && V && dQ && dK && dV && B > 0 && H > 0 && bmhrl_small_attention_ok(Sq, Sk, dk));
  BMHRL_CHECK_ARG(lddo % 8 == 0 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0);
  BMHRL_CHECK_ARG(ldp % 8 == 0 && ldp >= Sk && ldp <= 32 && (long)B * H < (1l << 31));
  BMHRL_CHECK_ARG(aligned16(dO) && aligned16(P) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(dQ) && aligned16(dK) &&
                  aligned16(dV));
  SmallAttnArgs a = {};
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.P = (bf16_t*)const_cast<void*>(P); a.ldp = ldp;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
  a.dO = (const bf16_t*)dO; a.lddo = lddo; a.dQ = (bf16_t*)dQ; a.lddq = lddq; a.dK = (bf16_t*)dK; a.lddk = lddk;
  a.dV = (bf16_t*)dV; a.lddv = lddv; a.dbq = dbq; a.dbk = dbk; a.dbv = dbv;
  dim3 grid((unsigned)(B * H)), block(256);
  switch (dk) {
    case 64: hipLaunchKernelGGL(small_attn_bwd_kernel<64>, grid, block, 0, (hipStream_t)stream, a); break;
    case 128: hipLaunchKernelGGL(small_attn_bwd_kernel<128>, grid, block, 0, (hipStream_t)stream, a); break;
    case 256: hipLaunchKernelGGL(small_attn_bwd_kernel<256>, grid, block, 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(small_attn_bwd_kernel<512>, grid, block, 0, (hipStream_t)stream, a); break;
  }
  return hip_status(hipGetLastError());
}
