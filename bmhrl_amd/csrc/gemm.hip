// Batched bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// C[b](M,N) = epilogue( sum_k A[b](m,k) * B[b](k,n) ), fp32 accumulate in v_mfma_f32_32x32x16_bf16.
// One kernel serves nn.Linear forward (A row-major, B = weight [N][K]), its data gradient
// (B "transposed": weight read as [K][N]) and its weight gradient (both operands transposed), plus
// the batched Q.K^T / P.V style products of the attention backward.  Transposed operands are staged
// into LDS in their natural (coalesced) layout and transposed for free on the way to the MFMA by
// ds_read_b64_tr_b16; k-contiguous operands are read with ds_read_b128 from rows padded by 16 B.
//
// Block = 4 waves (2 x 2), each wave owns (32*TM) x (32*TN) of the (64*TM) x (64*TN) block tile, BK = 64.
// Global -> register -> LDS staging is software pipelined one K-tile ahead (two LDS buffers, one barrier
// per K-tile).  Roofline: MFMA-bound for the 1024-wide projections; HBM-bound for K <= 128.
#include <cstdlib>
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

struct GemmArgs {
  int M, N, K, batch2;
  const bf16_t* A; long lda, a_sb1, a_sb2;
  const bf16_t* B; long ldb, b_sb1, b_sb2;
  float* C; long ldc, c_sb1, c_sb2;
  bf16_t* Cb; long ldcb, cb_sb1, cb_sb2;
  int epilogue; float alpha; int relu; int accumulate;
  const float* bias;
  const float* residual; long ldr, r_sb1, r_sb2;
  const uint8_t* mask; long mask_sb1, mask_sm;
  const float* rowvec; const float* rowvec2; long rv_sb1, rv_sb2;
  const bf16_t* aux; long ldaux, aux_sb1, aux_sb2;
  float dropout_p; uint64_t seed; const uint64_t* seed_dev; long drop_sb1, drop_sb2, drop_sm;
  float* colsum; long cs_sb2, bias_sb2, cs_sb1, bias_sb1;
  int tiles_m, splits, k_per_split, vec_ok, dbg, fast_bf16, fast_pd, tiles_mn;
  int xcd_chunk, group_m;   // tile order of the direct-to-LDS kernel: see tile_of_block (0, 0: m fastest, the plain order)
  float* split_ws;          // ordered K split: the splits' partial tiles [batch][split][M][N] (summed by splitk_reduce_kernel)
};

// Which output tile a workgroup of the direct-to-LDS kernel owns.  Workgroups are dealt round-robin over the 8 XCDs, each with
// its own L2.  Plain order (m fastest): an XCD gets every eighth row panel and EVERY column panel -- a 16 x 16 grid of tiles puts
// 2 A panels and 16 B panels through each L2.  With xcd_chunk (= tiles / 8, when that divides) an XCD owns a contiguous run of
// the tile sequence, and the sequence itself walks groups of group_m row panels across all columns, so the run is a near-square
// patch of the output: 4 x 8 tiles = 4 + 8 panels instead of 2 + 16.
__device__ __forceinline__ void tile_of_block(const GemmArgs& p, int bx, int& tile_m, int& tile_n) {
  if (p.xcd_chunk) bx = (bx & 7) * p.xcd_chunk + (bx >> 3);
  if (p.group_m) {
    const int tiles_n = p.tiles_mn / p.tiles_m, per_group = p.group_m * tiles_n;
    const int g = bx / per_group, r = bx - g * per_group;
    const int gm = min(p.group_m, p.tiles_m - g * p.group_m);        // (the last group may be short)
    tile_n = r / gm;
    tile_m = g * p.group_m + (r - tile_n * gm);
  } else {
    tile_m = bx % p.tiles_m;
    tile_n = bx / p.tiles_m;
  }
}

constexpr int BK = 64;

// Tuning aid (-DBMHRL_GEMM_TRACE, tests/kbench/build.sh trace): cycle stamps of wave 0 of the first and the last workgroup of the
// direct-to-LDS kernel -- entry, first loads requested, first tile published, main loop done, epilogue done.
#ifdef BMHRL_GEMM_TRACE
__device__ long long g_gemm_trace[2][8];
#define BMHRL_GSTAMP(i)                                                                                   \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    g_gemm_trace[blockIdx.x != 0][i] = (long long)__builtin_readcyclecounter();
#else
#define BMHRL_GSTAMP(i)
#endif

// ---- epilogue, shared by the two main loops.  The accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) +
// 4*(lane>>5)) go through LDS (`smem`: SMEM_ELEMS bf16 elements, free once every wave is past the main loop).
template <int TM, int TN, int SMEM_ELEMS>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], bf16_t* smem, const int b1, const int b2,
                                              const int m0, const int n0, const int split) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) go through
  // LDS so that every thread then owns 4 consecutive columns of a row: residual / aux / mask loads and the C stores
  // are 8- or 16-byte, fully coalesced accesses instead of 2- or 4-byte ones at a 32-lane stride.
  if (p.dbg == 3) { if (acc[0][0][0] == 123.f) p.C[0] = 1.f; return; }
  // Fast path for the most common output: bf16 only, plain linear epilogue (bias / ReLU / dropout), aligned.  The values
  // are finished in registers (a lane owns ONE output column, so the bias is a scalar per MFMA tile), staged as bf16
  // (half the LDS bytes of the generic fp32 staging) and written with 16-byte stores: a wave covers whole 256-byte row
  // segments.  The projections with K = 128 (audio stream) are bound by exactly this output write.
  if (p.fast_bf16 && p.splits == 1) {
    constexpr int SCB = BN + 8;
    bf16_t* sCb = reinterpret_cast<bf16_t*>(smem);
    bf16_t* __restrict__ Cbf = p.Cb + b1 * p.cb_sb1 + b2 * p.cb_sb2;
    const float* __restrict__ bp = p.bias ? p.bias + b1 * p.bias_sb1 + b2 * p.bias_sb2 : nullptr;
    const uint64_t seedf = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
    const uint64_t dbase = (uint64_t)b1 * p.drop_sb1 + (uint64_t)b2 * p.drop_sb2;
    // (the bias values of the lane's TN columns are requested together, and the uniform relu / dropout switches are decided
    // ONCE: a scalar branch per element -- two per element, 128 per wave at 128 x 128 -- cost more than the staging itself)
    float biasv[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int n = n0 + wn * 32 * TN + ni * 32 + r32;
      biasv[ni] = (bp && n < p.N) ? bp[n] : 0.f;
    }
    auto stage_bf16 = [&](auto relu_, auto drop_) {
      constexpr bool RELU = decltype(relu_)::value, DROP = decltype(drop_)::value;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          const f32x16& av = acc[mi][ni];
          const int col = wn * 32 * TN + ni * 32 + r32, n = n0 + col;
          const int row0 = wm * 32 * TM + mi * 32 + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            float v = av[r] * p.alpha + biasv[ni];
            if constexpr (RELU) v = fmaxf(v, 0.f);
            if constexpr (DROP) v *= dropout_scale(p.dropout_p, seedf, dbase + (uint64_t)(m0 + row) * p.drop_sm + n);
            sCb[row * SCB + col] = (bf16_t)v;
          }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (p.dropout_p > 0.f) { if (p.relu) stage_bf16(T_{}, T_{}); else stage_bf16(F_{}, T_{}); }
    else { if (p.relu) stage_bf16(T_{}, F_{}); else stage_bf16(F_{}, F_{}); }
    __syncthreads();
    if (p.dbg == 5) return;                        // (tuning aid: staging only)
    constexpr int G8 = BM * BN / 8 / 256;          // 16-byte groups per thread
#pragma unroll
    for (int i = 0; i < G8; ++i) {
      const int g = tid + i * 256;
      const int row = g / (BN / 8), c8 = (g % (BN / 8)) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(sCb + row * SCB + c8);
        bf16_t* dst = Cbf + (long)m * p.ldcb + n;
        if (p.dbg == 4) { if (v[0] == (bf16_t)123.25f) dst[0] = v[0]; continue; }     // (tuning aid: the epilogue without its stores)
        if (n + 8 <= p.N) *reinterpret_cast<bf16x8*>(dst) = v;
        else for (int j = 0; j < 8 && n + j < p.N; ++j) dst[j] = v[j];
      }
    }
    return;
  }
  // Same idea for the two softmax epilogues of the attention backward (P recomputation, dS): the row vectors (row max /
  // 1 / row sum, or delta) are staged in LDS once per tile (the generic path re-loads and re-divides them per 4-column
  // group), the P tile that dS needs is brought in with 16-byte loads and updated in place, the result leaves as
  // 16-byte stores.  These GEMMs have K = d_k = 256, i.e. four k-steps: they are all epilogue.
  if (p.fast_pd && p.splits == 1) {
    constexpr int SCB = BN + 8;
    bf16_t* sCb = reinterpret_cast<bf16_t*>(smem);
    float* sRV = reinterpret_cast<float*>(smem + BM * SCB);
    float* sRV2 = sRV + BM;
    static_assert((BM * SCB) * 2 + 2 * BM * 4 <= SMEM_ELEMS * 2, "softmax epilogue staging must fit");
    bf16_t* __restrict__ Cbf = p.Cb + b1 * p.cb_sb1 + b2 * p.cb_sb2;
    const bool prob = p.epilogue == BMHRL_EPI_PROB;
    const float* __restrict__ rv = p.rowvec + b1 * p.rv_sb1 + b2 * p.rv_sb2;
    const float* __restrict__ rv2 = prob ? p.rowvec2 + b1 * p.rv_sb1 + b2 * p.rv_sb2 : nullptr;
    if (tid < BM) {
      const int m = min(m0 + tid, p.M - 1);
      sRV[tid] = rv[m];
      sRV2[tid] = prob ? __builtin_amdgcn_rcpf(rv2[m]) : 0.f;
    }
    constexpr int G8 = BM * BN / 8 / 256;
    if (!prob) {   // P tile -> LDS
      const bf16_t* __restrict__ Pg = p.aux + b1 * p.aux_sb1 + b2 * p.aux_sb2;
#pragma unroll
      for (int i = 0; i < G8; ++i) {
        const int g = tid + i * 256;
        const int row = g / (BN / 8), c8 = (g % (BN / 8)) * 8;
        const int m = m0 + row, n = n0 + c8;
        bf16x8 v = zero_bf16x8();
        if (m < p.M && n < p.N) {
          const bf16_t* src = Pg + (long)m * p.ldaux + n;
          if (n + 8 <= p.N) v = *reinterpret_cast<const bf16x8*>(src);
          else for (int j = 0; j < 8 && n + j < p.N; ++j) v[j] = src[j];
        }
        *reinterpret_cast<bf16x8*>(sCb + row * SCB + c8) = v;
      }
    }
    __syncthreads();
    // key mask (mask_sm == 0).  PROB: masked scores are the -1e9 fill; DSCORE: the fill is a constant, so no gradient reaches
    // the score of a masked key (masked_fill, model/multihead_attention.py:22 -- it only matters for a fully masked row,
    // whose probabilities are uniform instead of zero)
    const uint8_t* __restrict__ Mk = p.mask ? p.mask + b1 * p.mask_sb1 : nullptr;
    // a lane's 16 accumulator rows are 4 groups of 4 consecutive rows: the row vectors come in as 16-byte LDS reads, once
    // per 32-row tile (not once per element and output tile)
    auto finish = [&](const f32x16& av, const int mi, const int ni, const f32x4 (&r1)[4], const f32x4 (&r2)[4]) {
      const int col = wn * 32 * TN + ni * 32 + r32, n = n0 + col;
      const bool keep = !Mk || n >= p.N || Mk[n] != 0;
      const int row0 = wm * 32 * TM + mi * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        bf16_t* q = sCb + row * SCB + col;
        float v;
        if (prob) {
          const float x = keep ? av[r] * p.alpha : NEG_MASK;
          v = __expf(x - r1[r >> 2][r & 3]) * r2[r >> 2][r & 3];
        } else {
          v = keep ? (float)*q * (av[r] - r1[r >> 2][r & 3]) * p.alpha : 0.f;
        }
        *q = (bf16_t)v;
      }
    };
    auto finish_rows = [&](const int mi) {
      const int row0 = wm * 32 * TM + mi * 32 + 4 * h;
      f32x4 r1[4], r2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        r1[g] = *reinterpret_cast<const f32x4*>(sRV + row0 + 8 * g);
        r2[g] = *reinterpret_cast<const f32x4*>(sRV2 + row0 + 8 * g);
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) finish(acc[mi][ni], mi, ni, r1, r2);
    };
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) finish_rows(mi);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G8; ++i) {
      const int g = tid + i * 256;
      const int row = g / (BN / 8), c8 = (g % (BN / 8)) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(sCb + row * SCB + c8);
        bf16_t* dst = Cbf + (long)m * p.ldcb + n;
        if (n + 8 <= p.N) *reinterpret_cast<bf16x8*>(dst) = v;
        else for (int j = 0; j < 8 && n + j < p.N; ++j) dst[j] = v[j];
      }
    }
    return;
  }
  constexpr int SC = BN + 4;
  static_assert(BM * SC * 2 <= SMEM_ELEMS, "C tile must fit in the staging buffers");
  float* sC = reinterpret_cast<float*>(smem);
  auto stage = [&](const f32x16& av, const int mi, const int ni) {
    float* base = sC + (wm * 32 * TM + mi * 32 + 4 * h) * SC + wn * 32 * TN + ni * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * SC] = av[r];
  };
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) stage(acc[mi][ni], mi, ni);
  __syncthreads();
  if (p.dbg == 5) return;                          // (tuning aid: staging only)

  float* __restrict__ Cg = p.C ? p.C + b1 * p.c_sb1 + b2 * p.c_sb2 : nullptr;
  bf16_t* __restrict__ Cbg = p.Cb ? p.Cb + b1 * p.cb_sb1 + b2 * p.cb_sb2 : nullptr;
  const float* __restrict__ Rg = p.residual ? p.residual + b1 * p.r_sb1 + b2 * p.r_sb2 : nullptr;
  const uint8_t* __restrict__ Mg = p.mask ? p.mask + b1 * p.mask_sb1 : nullptr;
  const float* __restrict__ RVg = p.rowvec ? p.rowvec + b1 * p.rv_sb1 + b2 * p.rv_sb2 : nullptr;
  const float* __restrict__ RV2g = p.rowvec2 ? p.rowvec2 + b1 * p.rv_sb1 + b2 * p.rv_sb2 : nullptr;
  const bf16_t* __restrict__ AUXg = p.aux ? p.aux + b1 * p.aux_sb1 + b2 * p.aux_sb2 : nullptr;
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  const uint64_t drop_base = (uint64_t)b1 * p.drop_sb1 + (uint64_t)b2 * p.drop_sb2;
  const bool first_split = split == 0;
  const float* __restrict__ biasp = (first_split && p.bias) ? p.bias + b1 * p.bias_sb1 + b2 * p.bias_sb2 : nullptr;
  if (!first_split) Rg = nullptr;

  // element-wise part, one variant per epilogue kind (KIND: 0 plain linear, 1 linear with mask / dropout, 2 PROB,
  // 3 DSCORE, 4 RELU_BWD); x = accumulator, returns the output value
  auto elem = [&](auto kind, float x, float bias, float res, float aux, float rv, float rv2, int m, int n, bool keep) -> float {
    constexpr int KIND = decltype(kind)::value;
    if constexpr (KIND == 0) {
      x = x * p.alpha + bias;
      if (p.relu) x = fmaxf(x, 0.f);
      return x + res;
    } else if constexpr (KIND == 1) {
      x = x * p.alpha + bias;
      if (!keep) x = NEG_MASK;
      if (p.relu) x = fmaxf(x, 0.f);
      if (p.dropout_p > 0.f) x *= dropout_scale(p.dropout_p, seed, drop_base + (uint64_t)m * p.drop_sm + n);
      return x + res;
    } else if constexpr (KIND == 2) {
      x = x * p.alpha;
      if (!keep) x = NEG_MASK;
      return __expf(x - rv) * rv2;
    } else if constexpr (KIND == 3) {
      return keep ? aux * (x - rv) * p.alpha : 0.f;
    } else {
      return aux > 0.f ? x * p.alpha : 0.f;
    }
  };

  constexpr int GROUPS = BM * BN / 4 / 256;
  // optional column sums of the OUTPUT tile (the bias gradient of the layer whose dY this GEMM produces): a thread
  // always works on the same 4 columns (256 % (BN/4) == 0), so it keeps a private partial and the 256 / (BN/4) threads
  // sharing a column group are combined through LDS: one atomic per column per tile
  float* __restrict__ CSg = p.colsum ? p.colsum + b1 * p.cs_sb1 + b2 * p.cs_sb2 : nullptr;
  f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};
  // (r04: a two-pass form of this loop -- every global operand of a thread's 16 row pieces requested first, then compute + stores
  // -- was built against the suspicion that the per-piece conditional loads make every piece wait for the stores of the one
  // before (vmcnt counts stores): 28.8 vs 26.2 us at 4096 x 1024 x 1024 with bias + dropout + residual, 22.3 vs 19.8 plain,
  // and the token-step GEMMs of the incremental decoder slower too -- dropped.)
  auto run = [&](auto kind) {
#pragma unroll 4
    for (int i = 0; i < GROUPS; ++i) {
      const int g = tid + i * 256;
      const int row = g / (BN / 4), c4 = (g % (BN / 4)) * 4;
      const int m = m0 + row, n = n0 + c4;
      if (m < p.M && n < p.N) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sC + row * SC + c4);
        const float rv = RVg ? RVg[m] : 0.f;
        const float rv2 = RV2g ? __builtin_amdgcn_rcpf(RV2g[m]) : 1.f;      // 1 ulp reciprocal: P is rounded to bf16 anyway
        if (p.vec_ok && n + 4 <= p.N) {
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, r4 = {0.f, 0.f, 0.f, 0.f}, x4 = {0.f, 0.f, 0.f, 0.f};
          if (biasp) b4 = *reinterpret_cast<const f32x4*>(biasp + n);
          if (Rg) r4 = *reinterpret_cast<const f32x4*>(Rg + (long)m * p.ldr + n);
          if (AUXg) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(AUXg + (long)m * p.ldaux + n);
            x4[0] = (float)t[0]; x4[1] = (float)t[1]; x4[2] = (float)t[2]; x4[3] = (float)t[3];
          }
          uint32_t keep4 = 0x01010101u;        // mask bytes of the 4 columns (one 4-byte load when aligned)
          if constexpr (decltype(kind)::value == 1 || decltype(kind)::value == 2 || decltype(kind)::value == 3) {
            if (Mg) {
              const uint8_t* mp = Mg + (long)m * p.mask_sm + n;
              if (((uintptr_t)mp & 3) == 0) keep4 = *reinterpret_cast<const uint32_t*>(mp);
              else keep4 = (uint32_t)mp[0] | ((uint32_t)mp[1] << 8) | ((uint32_t)mp[2] << 16) | ((uint32_t)mp[3] << 24);
            }
          }
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = elem(kind, a4[j], b4[j], r4[j], x4[j], rv, rv2, m, n + j, ((keep4 >> (8 * j)) & 0xffu) != 0);
          cs4 += o;
          if (p.dbg == 4) { if (o[0] == 123.25f && Cg) Cg[0] = o[0]; continue; }      // (tuning aid: the epilogue without its stores)
          if (Cg) {
            float* dst = Cg + (long)m * p.ldc + n;
            if (p.splits > 1) {
#pragma unroll
              for (int j = 0; j < 4; ++j) atomicAdd(dst + j, o[j]);
            } else {
              if (p.accumulate) o += *reinterpret_cast<const f32x4*>(dst);
              *reinterpret_cast<f32x4*>(dst) = o;
            }
          }
          if (Cbg) {
            bf16x4 ob;
            ob[0] = (bf16_t)o[0]; ob[1] = (bf16_t)o[1]; ob[2] = (bf16_t)o[2]; ob[3] = (bf16_t)o[3];
            *reinterpret_cast<bf16x4*>(Cbg + (long)m * p.ldcb + n) = ob;
          }
        } else {   // ragged right edge or unaligned operands: scalar accesses
          for (int j = 0; j < 4 && n + j < p.N; ++j) {
            const float b = biasp ? biasp[n + j] : 0.f;
            const float r = Rg ? Rg[(long)m * p.ldr + n + j] : 0.f;
            const float ax = AUXg ? (float)AUXg[(long)m * p.ldaux + n + j] : 0.f;
            const bool keep = !Mg || Mg[(long)m * p.mask_sm + n + j] != 0;
            const float o = elem(kind, a4[j], b, r, ax, rv, rv2, m, n + j, keep);
            cs4[j] += o;
            if (Cg) {
              float* dst = Cg + (long)m * p.ldc + n + j;
              if (p.splits > 1) atomicAdd(dst, o);
              else *dst = p.accumulate ? *dst + o : o;
            }
            if (Cbg) Cbg[(long)m * p.ldcb + n + j] = (bf16_t)o;
          }
        }
      }
    }
  };
  if (p.splits > 1 && p.split_ws) {
    // ordered K split: the raw partial tile goes to this split's slab with plain stores; splitk_reduce_kernel adds the slabs in
    // split order (and applies alpha / bias / residual / accumulate), so the result does not depend on who arrives when
    float* __restrict__ slab = p.split_ws + (((long)(b1 * p.batch2 + b2) * p.splits + split) * p.M) * p.N;
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int row = idx / BN, col = idx % BN;
      const int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) slab[(long)m * p.N + n] = sC[row * SC + col];
    }
    return;
  }
  if (p.splits > 1) {
    // split-K partial sums: lane l adds column l of a row, so one wave instruction covers 256 contiguous bytes (the
    // shape float atomics run at full rate with); bias / residual are added by the first split only.
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int row = idx / BN, col = idx % BN;
      const int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) {
        float x = sC[row * SC + col] * p.alpha;
        if (biasp) x += biasp[n];
        if (Rg) x += Rg[(long)m * p.ldr + n];
        atomicAdd(Cg + (long)m * p.ldc + n, x);
      }
    }
    return;
  }
  if (p.epilogue == BMHRL_EPI_LINEAR) {
    if (!Mg && p.dropout_p == 0.f) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  } else if (p.epilogue == BMHRL_EPI_PROB) run(std::integral_constant<int, 2>{});
  else if (p.epilogue == BMHRL_EPI_DSCORE) run(std::integral_constant<int, 3>{});
  else run(std::integral_constant<int, 4>{});
  if (CSg) {
    constexpr int CG = BN / 4, SHARE = 256 / CG;        // column groups per tile, threads per group
    __syncthreads();                                    // everyone is done reading the staged accumulators
    *reinterpret_cast<f32x4*>(sC + (tid / CG) * BN + (tid % CG) * 4) = cs4;
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < SHARE; ++k) t += sC[k * BN + tid];
      atomicAdd(CSg + n0 + tid, t);
    }
  }
}

template <int TM, int TN, bool AT, bool BT>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bx, const int by, const int bz) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  // LDS row strides (elements).  k-contiguous tiles: +8 (16 B) keeps ds_read_b128 conflict free;
  // transposed tiles ([k][m]): +32 (64 B) puts the 4 rows of a tr-read block on disjoint bank quarters.
  constexpr int SA = AT ? (BM + 32) : (BK + 8);
  constexpr int SB = BT ? (BN + 32) : (BK + 8);
  constexpr int ROWS_A = AT ? BK : BM, ROWS_B = BT ? BK : BN;
  constexpr int A_ELEMS = ROWS_A * SA, B_ELEMS = ROWS_B * SB;
  constexpr int CH_A = BM * BK / 8 / 256, CH_B = BN * BK / 8 / 256;  // 16-byte chunks per thread
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (A_ELEMS + B_ELEMS)];

  if (p.dbg == 1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;

  const int b1 = bz / p.batch2, b2 = bz % p.batch2;
  const int tile_m = bx % p.tiles_m, tile_n = bx / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ Ag = p.A + b1 * p.a_sb1 + b2 * p.a_sb2;
  const bf16_t* __restrict__ Bg = p.B + b1 * p.b_sb1 + b2 * p.b_sb2;
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  // split-K: this block reduces k in [k_begin, k_end)
  const int k_begin = by * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int K8 = (k_end == p.K) ? ((p.K + 7) & ~7) : k_end;

  // Staging: global -> registers -> LDS, one K-tile ahead (two LDS buffers, one barrier per tile; a second
  // co-resident block per CU covers this block's load latency).  The hot loop carries no bounds logic: each chunk
  // has a constant 32-bit element offset from a uniform base that advances by one K-tile per iteration (rows /
  // columns past the matrix edge are clamped to valid memory and only feed outputs that are never stored); only
  // the last, ragged K-tile of a reduction is loaded through the masked path that zero-fills k >= K.
  bf16x8 ra[2][CH_A], rb[2][CH_B];   // two staging sets: the tile two steps ahead is requested while the previous one lands
  int a_off[CH_A], b_off[CH_B], a_lds[CH_A], b_lds[CH_B], a_kk[CH_A], b_kk[CH_B];
#pragma unroll
  for (int i = 0; i < CH_A; ++i) {
    const int c = tid + i * 256;
    const int row = AT ? c / (BM / 8) : c / (BK / 8), col = AT ? c % (BM / 8) : c % (BK / 8);
    a_lds[i] = row * SA + col * 8;
    a_kk[i] = AT ? row : col * 8;                                   // k index of this chunk inside a tile
    a_off[i] = AT ? min(m0 + col * 8, M8 - 8) : min(m0 + row, p.M - 1) * (int)p.lda;
  }
#pragma unroll
  for (int i = 0; i < CH_B; ++i) {
    const int c = tid + i * 256;
    const int row = BT ? c / (BN / 8) : c / (BK / 8), col = BT ? c % (BN / 8) : c % (BK / 8);
    b_lds[i] = row * SB + col * 8;
    b_kk[i] = BT ? row : col * 8;
    b_off[i] = BT ? min(n0 + col * 8, N8 - 8) : min(n0 + row, p.N - 1) * (int)p.ldb;
  }
  const int Klast8 = ((p.K + 7) & ~7) - 8;   // last valid 8-wide chunk along a k-contiguous row

  auto load_fast = [&](int k0, auto set_) {
    constexpr int ST = decltype(set_)::value;   // every k of the tile is < K
    const bf16_t* Ak = Ag + (AT ? (long)k0 * p.lda : (long)k0);
    const bf16_t* Bk = Bg + (BT ? (long)k0 * p.ldb : (long)k0);
#pragma unroll
    for (int i = 0; i < CH_A; ++i) ra[ST][i] = *reinterpret_cast<const bf16x8*>(Ak + a_off[i] + (AT ? a_kk[i] * (int)p.lda : a_kk[i]));
#pragma unroll
    for (int i = 0; i < CH_B; ++i) rb[ST][i] = *reinterpret_cast<const bf16x8*>(Bk + b_off[i] + (BT ? b_kk[i] * (int)p.ldb : b_kk[i]));
  };
  auto load_tail = [&](int k0, auto set_) {
    constexpr int ST = decltype(set_)::value;   // ragged last tile: clamp, then zero what lies at k >= K
#pragma unroll
    for (int i = 0; i < CH_A; ++i) {
      const int k = k0 + a_kk[i];
      const bool ok = AT ? (k < k_end) : (k < K8);
      const bf16_t* q = AT ? Ag + (long)min(k, p.K - 1) * p.lda + a_off[i] : Ag + a_off[i] + min(k, Klast8);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(q);
      ra[ST][i] = ok ? v : zero_bf16x8();
    }
#pragma unroll
    for (int i = 0; i < CH_B; ++i) {
      const int k = k0 + b_kk[i];
      const bool ok = BT ? (k < k_end) : (k < K8);
      const bf16_t* q = BT ? Bg + (long)min(k, p.K - 1) * p.ldb + b_off[i] : Bg + b_off[i] + min(k, Klast8);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(q);
      rb[ST][i] = ok ? v : zero_bf16x8();
    }
  };
  auto store_tiles = [&](int buf, auto set_) {
    constexpr int ST = decltype(set_)::value;
    bf16_t* sA = smem + buf * (A_ELEMS + B_ELEMS);
    bf16_t* sB = sA + A_ELEMS;
#pragma unroll
    for (int i = 0; i < CH_A; ++i) *reinterpret_cast<bf16x8*>(sA + a_lds[i]) = ra[ST][i];
#pragma unroll
    for (int i = 0; i < CH_B; ++i) *reinterpret_cast<bf16x8*>(sB + b_lds[i]) = rb[ST][i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int cur) {
    const bf16_t* sA = smem + cur * (A_ELEMS + B_ELEMS);
    const bf16_t* sB = sA + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int mb = wm * 32 * TM + mi * 32;
        if (AT) {
          const bf16_t* base = sA + (ks * 16 + 8 * h + q4) * SA + mb + 16 * g1 + 4 * p4;
          af[mi] = join8(lds_read_tr4(base), lds_read_tr4(base + 4 * SA));
        } else {
          af[mi] = *reinterpret_cast<const bf16x8*>(sA + (mb + r32) * SA + ks * 16 + 8 * h);
        }
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int nb = wn * 32 * TN + ni * 32;
        if (BT) {
          const bf16_t* base = sB + (ks * 16 + 8 * h + q4) * SB + nb + 16 * g1 + 4 * p4;
          bfr[ni] = join8(lds_read_tr4(base), lds_read_tr4(base + 4 * SB));
        } else {
          bfr[ni] = *reinterpret_cast<const bf16x8*>(sB + (nb + r32) * SB + ks * 16 + 8 * h);
        }
      }
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
  };

  const int nk = (k_end - k_begin + BK - 1) / BK;
  if (nk <= 0 || p.dbg == 2) return;   // empty split (uniform for the whole block)
  const bool ragged = (k_end - k_begin) % BK != 0;         // the last tile holds k >= K (zero-filled)
  auto load_any = [&](int t, auto set_) {
    if (ragged && t == nk - 1) load_tail(k_begin + t * BK, set_);
    else load_fast(k_begin + t * BK, set_);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // Invariant at the top of step kt (parity q = kt & 1): LDS buffer q holds tile kt, register set 1-q holds (or is
  // receiving) tile kt+1, register set q is free.  A step requests tile kt+2 into set q, multiplies tile kt, then moves
  // set 1-q into LDS buffer 1-q: that store waits only for loads issued a whole step earlier.
  load_any(0, S0{});
  if (nk > 1) load_any(1, S1{});
  store_tiles(0, S0{});
  __syncthreads();
  int kt = 0;
  // steady state, two steps per trip, no branches (so the compiler's wait counts stay exact: vmcnt = the loads of the
  // newer tile): every tile requested here is a full one
  // (split-K blocks keep the plain one-tile-ahead steps below: measured 20-30 % slower with the deep prefetch)
  const int nk_full = p.splits > 1 ? 0 : (ragged ? nk - 1 : nk);      // tiles [0, nk_full) have every k < K
  for (; kt + 3 < nk_full; kt += 2) {
    load_fast(k_begin + (kt + 2) * BK, S0{});
    __builtin_amdgcn_sched_barrier(0);     // the requests go out first (the scheduler would sink them behind the stores)
    compute(0);
    store_tiles(1, S1{});
    __syncthreads();
    load_fast(k_begin + (kt + 3) * BK, S1{});
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    store_tiles(0, S0{});
    __syncthreads();
  }
  // the last (up to three or four) tiles, incl. a ragged one: same steps with their conditions
  for (; kt < nk; kt += 2) {
    if (kt + 2 < nk) load_any(kt + 2, S0{});
    compute(0);
    if (kt + 1 < nk) store_tiles(1, S1{});
    __syncthreads();
    if (kt + 1 < nk) {
      if (kt + 3 < nk) load_any(kt + 3, S1{});
      compute(1);
      if (kt + 2 < nk) store_tiles(0, S0{});
      __syncthreads();
    }
  }

  gemm_epilogue<TM, TN, 2 * (A_ELEMS + B_ELEMS)>(p, acc, smem, b1, b2, m0, n0, by);    // (every step ends in a barrier)
}

template <int TM, int TN, bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
  gemm_body<TM, TN, AT, BT>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Up to four independent problems of the same kernel variant in ONE launch (the leaf weight-gradient products of a fusion
// layer's backward: each is a few microseconds of work behind a launch, and nothing but the optimizer waits for them).
// Block b belongs to the problem whose range [first[i], first[i + 1]) holds it; inside its problem the blocks are ordered
// (batch, split, tile).  The problem's arguments are picked field by field with uniform selects (an indexed copy of a
// kernel-argument struct would live in scratch memory).
struct GemmGroup {
  GemmArgs p0, p1, p2, p3;
  int first[5];
  int n;
};
template <int TM, int TN, bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_group_kernel(const GemmGroup g) {
  const int b = (int)blockIdx.x;
  int i = 0;
  if (g.n > 1 && b >= g.first[1]) i = 1;
  if (g.n > 2 && b >= g.first[2]) i = 2;
  if (g.n > 3 && b >= g.first[3]) i = 3;
  GemmArgs p = g.p0;
  if (i == 1) p = g.p1;
  if (i == 2) p = g.p2;
  if (i == 3) p = g.p3;
  int r = b - g.first[i];
  const int bx = r % p.tiles_mn;
  r /= p.tiles_mn;
  gemm_body<TM, TN, AT, BT>(p, bx, r % p.splits, r / p.splits);
}


// ---- Main loop on direct-to-LDS loads (every reduction whose k range is a multiple of 64: all the large projections).
// global_load_lds moves 1 KiB per wave instruction straight into the tile image (no staging registers, no ds_write); the
// image is lane-linear, so bank conflicts are avoided by XOR-swizzling the SOURCE chunk and reading through the same XOR:
//   k-contiguous operand   [rows][64 k], 128-byte rows : 16-byte chunk ^= (row >> 1) & 7   (ds_read_b128: the 16 rows of a
//                                                        lane group fall on 16 different slots of the 256-byte bank row)
//   transposed operand     [64 k][R],  R*2-byte rows   : chunk ^= (k & 3) << 2 (R = 128) or ((k >> 1) & 1) << 2 (R = 64):
//                                                        the 4 k-rows of a ds_read_b64_tr_b16 block sit on 4 bank quarters
// Two stages; a step issues the loads of tile t+1, multiplies tile t and ends in vmcnt(0) + barrier.  The fragment reads
// are inline asm with counted lgkmcnt waits: the compiler would put a vmcnt(0) in front of every LDS read it can see while
// a direct-to-LDS load is in flight (it cannot tell the stages apart), serialising load and multiply; all LDS addresses
// are one VGPR per operand fragment row + immediates (the k-loop is unrolled over the two stages).
template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F>
__device__ __forceinline__ void gsfor(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    gsfor<I + 1, N>(f);
  }
}
template <int IMM>
__device__ __forceinline__ void gemm_glds16(const char* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, IMM, 0);
}

template <int OFF>
__device__ __forceinline__ bf16x8 gemm_lds128(unsigned addr) {     // (asm: see the header of the kernel below)
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x4 gemm_ldstr(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <int TM, int TN, bool AT, bool BT, int NS>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const GemmArgs p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int EPI_ELEMS = BM * (BN + 4) * 2;                       // fp32 C staging of the epilogue, in bf16 elements
  constexpr int SMEM_ELEMS = (NS * STAGE_BYTES / 2) > EPI_ELEMS ? (NS * STAGE_BYTES / 2) : EPI_ELEMS;

  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM_ELEMS];
  char* const sbase = reinterpret_cast<char*>(smem);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sbase;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int bz = blockIdx.z, b1 = bz / p.batch2, b2 = bz % p.batch2;
  int tile_m, tile_n;
  tile_of_block(p, (int)blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A + b1 * p.a_sb1 + b2 * p.a_sb2);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B + b1 * p.b_sb1 + b2 * p.b_sb2);
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  const int k_begin = blockIdx.y * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin) / BK;                             // whole tiles only (host-checked)
  if (nk <= 0) return;
  BMHRL_GSTAMP(0)

  // ---- staging: per-lane byte offsets from the operand base, fixed for the launch (rows / columns past the matrix edge
  // are clamped to valid memory: they only feed outputs that are never stored); the uniform base advances by one k-tile
  constexpr int PA = A_BYTES / 1024 / 4, PB = B_BYTES / 1024 / 4;     // 1 KiB pieces per wave and operand (4 / 2)
  unsigned a_off[PA], b_off[PB];
  auto piece_off = [&](const bool trans, const int R, const int i, const int r0, const int rmax, const int rmax8, const long ld) {
    if (!trans) {                       // [R rows][64 k]: 8 rows per piece
      const int row = wave * (R / 4) + 8 * i + (lane >> 3), pc = lane & 7;
      const int sc = pc ^ ((row >> 1) & 7);
      return (unsigned)((long)min(r0 + row, rmax - 1) * ld * 2 + sc * 16);
    }
    const int cpr = R / 8, rpp = 64 / cpr;                             // chunks per row, k-rows per piece
    const int krow = wave * 16 + rpp * i + lane / cpr, pc = lane % cpr;
    const int sc = pc ^ (R == 128 ? ((krow & 3) << 2) : (((krow >> 1) & 1) << 2));
    return (unsigned)((long)krow * ld * 2 + (long)min(r0 + sc * 8, rmax8 - 8) * 2);
  };
#pragma unroll
  for (int i = 0; i < PA; ++i) a_off[i] = piece_off(AT, BM, i, m0, p.M, M8, p.lda);
#pragma unroll
  for (int i = 0; i < PB; ++i) b_off[i] = piece_off(BT, BN, i, n0, p.N, N8, p.ldb);
  const long a_step = AT ? (long)BK * p.lda * 2 : (long)BK * 2, b_step = BT ? (long)BK * p.ldb * 2 : (long)BK * 2;
  const char* const a_k0 = Ag + (AT ? (long)k_begin * p.lda * 2 : (long)k_begin * 2);
  const char* const b_k0 = Bg + (BT ? (long)k_begin * p.ldb * 2 : (long)k_begin * 2);
  // LDS destination of this wave's pieces inside a stage: consecutive 1 KiB pieces (wave w: pieces [w*P, w*P + P))
  char* const a_dst = sbase + wave * PA * 1024;
  char* const b_dst = sbase + A_BYTES + wave * PB * 1024;
  auto stage = [&](auto st_, const int t) {        // tile t of this block's reduction -> stage ST
    constexpr int ST = decltype(st_)::value;
    // uniform (SGPR) base + the lane's fixed 32-bit offset: the readfirstlane keeps the sum in scalar registers (the
    // compiler otherwise re-associates it into two 64-bit vector adds per load)
    auto sgpr_ptr = [](const char* q) {
      const uint64_t u = reinterpret_cast<uint64_t>(q);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
      return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* ab = sgpr_ptr(a_k0 + (long)t * a_step);
    const char* bb = sgpr_ptr(b_k0 + (long)t * b_step);
    gsfor<0, PA>([&](auto i) {
      constexpr int I = decltype(i)::value;
      unsigned o = a_off[I];
      asm volatile("" : "+v"(o));
      gemm_glds16<0>(ab + o, a_dst + ST * STAGE_BYTES + I * 1024);
    });
    gsfor<0, PB>([&](auto i) {
      constexpr int I = decltype(i)::value;
      unsigned o = b_off[I];
      asm volatile("" : "+v"(o));
      gemm_glds16<0>(bb + o, b_dst + ST * STAGE_BYTES + I * 1024);
    });
  };

  // ---- fragment addresses, one set per stage (the k-steps / fragment rows are 16-bit immediates)
  //   k-contiguous: row = wX*32*T + 32*i + r32, chunk (2 ks + h) ^ ((r32 >> 1) & 7): one address per ks
  //   transposed  : k-row 16 ks + 8 h + q4 (+4), chunk (wX*4*T + 4 i + 2 g1 + (p4 >> 1)) ^ swizzle(q4): one address per i
  constexpr int NA = AT ? TM : 4, NB = BT ? TN : 4;
  unsigned a_addr0[NA], b_addr0[NB];              // stage 0; a step adds its stage's (uniform) byte offset
  if constexpr (!AT) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_addr0[ks] = lds0 + (wm * 32 * TM + r32) * 128 + (((2 * ks + h) ^ ((r32 >> 1) & 7)) << 4);
  } else {
    constexpr int RB = BM * 2;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int c = wm * 4 * TM + 4 * i + 2 * g1 + (p4 >> 1);
      const int sw = BM == 128 ? (q4 << 2) : (((q4 >> 1) & 1) << 2);
      a_addr0[i] = lds0 + (8 * h + q4) * RB + ((c ^ sw) << 4) + ((p4 & 1) << 3);
    }
  }
  if constexpr (!BT) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_addr0[ks] = lds0 + A_BYTES + (wn * 32 * TN + r32) * 128 + (((2 * ks + h) ^ ((r32 >> 1) & 7)) << 4);
  } else {
    constexpr int RB = BN * 2;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int c = wn * 4 * TN + 4 * i + 2 * g1 + (p4 >> 1);
      const int sw = BN == 128 ? (q4 << 2) : (((q4 >> 1) & 1) << 2);
      b_addr0[i] = lds0 + A_BYTES + (8 * h + q4) * RB + ((c ^ sw) << 4) + ((p4 & 1) << 3);
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one k-tile out of stage ST: the fragment reads run one k-step ahead of the MFMAs (LDS returns in order: "at most N
  // outstanding" = everything older has arrived; the counter has 4 bits, so no more than two k-steps are in flight)
  constexpr int RA = AT ? 2 : 1, RBn = BT ? 2 : 1;            // LDS reads per fragment
  constexpr int PER_KS = TM * RA + TN * RBn;                  // reads per k-step (<= 8)
  auto compute = [&](auto st_) {
    constexpr int ST = decltype(st_)::value;
    bf16x8 af[4][TM], bfr[4][TN];
    unsigned a_addr[1][NA], b_addr[1][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_addr[0][i] = a_addr0[i] + ST * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NB; ++i) b_addr[0][i] = b_addr0[i] + ST * STAGE_BYTES;
    auto read_ks = [&](auto ks_) {
      constexpr int KS = decltype(ks_)::value;
      gsfor<0, TM>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        if constexpr (!AT) {
          af[KS][I] = gemm_lds128<I * 32 * 128>(a_addr[0][KS]);
        } else {
          constexpr int RB = BM * 2;
          af[KS][I] = join8(gemm_ldstr<KS * 16 * RB>(a_addr[0][I]), gemm_ldstr<(KS * 16 + 4) * RB>(a_addr[0][I]));
        }
      });
      gsfor<0, TN>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        if constexpr (!BT) {
          bfr[KS][I] = gemm_lds128<I * 32 * 128>(b_addr[0][KS]);
        } else {
          constexpr int RB = BN * 2;
          bfr[KS][I] = join8(gemm_ldstr<KS * 16 * RB>(b_addr[0][I]), gemm_ldstr<(KS * 16 + 4) * RB>(b_addr[0][I]));
        }
      });
    };
    read_ks(IC<0>{});
    read_ks(IC<1>{});
    gsfor<0, 4>([&](auto ks_) {
      constexpr int KS = decltype(ks_)::value;
      constexpr int LEFT = KS < 3 ? PER_KS : 0;                // the reads of the next k-step may still be in flight
      __builtin_amdgcn_sched_barrier(0);
      bf16x8& a0 = af[KS][0];
      bf16x8& b0 = bfr[KS][0];
      bf16x8& a1 = af[KS][TM - 1];
      bf16x8& b1r = bfr[KS][TN - 1];
      if constexpr (TM == 2 && TN == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1r) : "n"(LEFT));
      else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a0), "+v"(b0) : "n"(LEFT));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[KS][mi], bfr[KS][ni], acc[mi][ni], 0, 0, 0);
      if constexpr (KS + 2 < 4) read_ks(IC<KS + 2>{});
    });
  };

  // NS stages: tiles t+1 .. t+NS-2 stay in flight across the barrier of step t (counted vmcnt: a wave's loads complete in
  // order, so "at most (NS-2) tiles' worth outstanding" = tile t+1 has landed).  Two stages suit shapes with >= 2
  // workgroups per CU (the other workgroup covers the wait); with one workgroup per CU -- the 4096 x 1024 projections: 256
  // tiles -- a load issued at the top of a 0.25 us step would be awaited at its end, and NS = 4 hides the latency instead.
  constexpr int PT = PA + PB;                                   // this wave's pieces per tile
  auto step = [&](auto st_, const int t) {                      // multiply tile t (stage ST = t % NS)
    constexpr int ST = decltype(st_)::value;
    constexpr int NXT = (ST + NS - 1) % NS;                     // the stage read in step t-1
    const bool more = t + NS - 1 < nk;
    if (more) stage(IC<NXT>{}, t + NS - 1);
    compute(st_);
    if (t + 1 < nk) {
      if (more && NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the tail drains completely: simple and at most NS-2 steps)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  gsfor<0, NS - 1>([&](auto i_) {
    if (decltype(i_)::value < nk) stage(i_, decltype(i_)::value);
  });
  BMHRL_GSTAMP(1)
  if (NS > 2 && nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PT) : "memory");   // tile 0 (conservative when nk < NS-1: see below)
  if (NS == 2 || nk < NS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_GSTAMP(2)
  for (int t0 = 0; t0 < nk; t0 += NS) {
    gsfor<0, NS>([&](auto i_) {
      if (t0 + decltype(i_)::value < nk) step(i_, t0 + decltype(i_)::value);
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();             // every wave is done with the stages: the epilogue reuses them
  asm volatile("" ::: "memory");
  BMHRL_GSTAMP(3)
  gemm_epilogue<TM, TN, SMEM_ELEMS>(p, acc, smem, b1, b2, m0, n0, (int)blockIdx.y);
  BMHRL_GSTAMP(4)
}

// ---- 128 x 128 tiles with EIGHT waves: two wave groups split every k-tile between them (intra-workgroup K split).
// The shapes that give a CU exactly one 128 x 128 tile (4096 x 1024 outputs: 256 tiles) leave the four waves of the kernel above
// alone on their SIMDs: a wave issues its 8 direct-to-LDS pieces, its 16 fragment reads and its 16 MFMAs of a k-tile in
// order, the barrier at the end of the step exposes whatever the two-stage ring did not cover, and prologue and epilogue
// have nothing beside them.  Their forward epilogues carry dropout and the residual, so a K split over WORKGROUPS (fp32
// atomics) is not available.  Here waves 0-3 and 4-7 both own the whole tile (the same 2 x 2 wave layout) and take k-steps
// {0, 1} and {2, 3} of every 64-deep k-tile: 8 MFMAs, 8 fragment reads and 4 pieces per wave and tile, two waves per SIMD that
// cover each other's waits, three stages with the loads of tile t + 2 in flight across the barrier of step t.  The two
// partial tiles meet in LDS (group 1 stores, group 0 adds: lane-private 16-byte slots, no conflicts), then group 0 runs the
// unchanged epilogue while group 1 has ended (a barrier counts the waves that are still there).
template <bool AT, bool BT>
__global__ __launch_bounds__(512, 2) void gemm_glds8_kernel(const GemmArgs p) {
  constexpr int TM = 2, TN = 2, NS = 3, NW = 8;
  constexpr int BM = 128, BN = 128;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int EPI_ELEMS = BM * (BN + 4) * 2;
  constexpr int SMEM_ELEMS = (NS * STAGE_BYTES / 2) > EPI_ELEMS ? (NS * STAGE_BYTES / 2) : EPI_ELEMS;
  static_assert(4 * 64 * 256 <= SMEM_ELEMS * 2, "the partial tile of wave group 1 fits the stages");

  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM_ELEMS];
  char* const sbase = reinterpret_cast<char*>(smem);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sbase;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int bz = blockIdx.z, b1 = bz / p.batch2, b2 = bz % p.batch2;
  const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const char* __restrict__ Ag = reinterpret_cast<const char*>(p.A + b1 * p.a_sb1 + b2 * p.a_sb2);
  const char* __restrict__ Bg = reinterpret_cast<const char*>(p.B + b1 * p.b_sb1 + b2 * p.b_sb2);
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  const int nk = p.K / BK;                                           // no K split over workgroups; whole tiles (host-checked)
  BMHRL_GSTAMP(0)

  // ---- staging: the stage images of gemm_glds_kernel, filled by eight waves (2 + 2 pieces each)
  constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW;
  unsigned a_off[PA], b_off[PB];
  auto piece_off = [&](const bool trans, const int R, const int i, const int r0, const int rmax, const int rmax8, const long ld) {
    if (!trans) {                       // [R rows][64 k]: 8 rows per piece
      const int row = wave * (R / NW) + 8 * i + (lane >> 3), pc = lane & 7;
      const int sc = pc ^ ((row >> 1) & 7);
      return (unsigned)((long)min(r0 + row, rmax - 1) * ld * 2 + sc * 16);
    }
    const int cpr = R / 8, rpp = 64 / cpr;                             // chunks per row, k-rows per piece
    const int krow = wave * (64 / NW) + rpp * i + lane / cpr, pc = lane % cpr;
    const int sc = pc ^ ((krow & 3) << 2);
    return (unsigned)((long)krow * ld * 2 + (long)min(r0 + sc * 8, rmax8 - 8) * 2);
  };
#pragma unroll
  for (int i = 0; i < PA; ++i) a_off[i] = piece_off(AT, BM, i, m0, p.M, M8, p.lda);
#pragma unroll
  for (int i = 0; i < PB; ++i) b_off[i] = piece_off(BT, BN, i, n0, p.N, N8, p.ldb);
  const long a_step = AT ? (long)BK * p.lda * 2 : (long)BK * 2, b_step = BT ? (long)BK * p.ldb * 2 : (long)BK * 2;
  char* const a_dst = sbase + wave * PA * 1024;
  char* const b_dst = sbase + A_BYTES + wave * PB * 1024;
  auto stage = [&](auto st_, const int t) {
    constexpr int ST = decltype(st_)::value;
    auto sgpr_ptr = [](const char* q) {
      const uint64_t u = reinterpret_cast<uint64_t>(q);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
      return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* ab = sgpr_ptr(Ag + (long)t * a_step);
    const char* bb = sgpr_ptr(Bg + (long)t * b_step);
    gsfor<0, PA>([&](auto i) {
      constexpr int I = decltype(i)::value;
      unsigned o = a_off[I];
      asm volatile("" : "+v"(o));
      gemm_glds16<0>(ab + o, a_dst + ST * STAGE_BYTES + I * 1024);
    });
    gsfor<0, PB>([&](auto i) {
      constexpr int I = decltype(i)::value;
      unsigned o = b_off[I];
      asm volatile("" : "+v"(o));
      gemm_glds16<0>(bb + o, b_dst + ST * STAGE_BYTES + I * 1024);
    });
  };

  // ---- fragment addresses (stage 0), as gemm_glds_kernel; this group's k-steps are 2 grp and 2 grp + 1
  constexpr int NA = AT ? TM : 2, NB = BT ? TN : 2;
  unsigned a_addr0[NA], b_addr0[NB];
  if constexpr (!AT) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
      a_addr0[k2] = lds0 + (wm * 64 + r32) * 128 + (((2 * (2 * grp + k2) + h) ^ ((r32 >> 1) & 7)) << 4);
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int c = wm * 8 + 4 * i + 2 * g1 + (p4 >> 1);
      a_addr0[i] = lds0 + (8 * h + q4 + 32 * grp) * (BM * 2) + ((c ^ (q4 << 2)) << 4) + ((p4 & 1) << 3);
    }
  }
  if constexpr (!BT) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
      b_addr0[k2] = lds0 + A_BYTES + (wn * 64 + r32) * 128 + (((2 * (2 * grp + k2) + h) ^ ((r32 >> 1) & 7)) << 4);
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int c = wn * 8 + 4 * i + 2 * g1 + (p4 >> 1);
      b_addr0[i] = lds0 + A_BYTES + (8 * h + q4 + 32 * grp) * (BN * 2) + ((c ^ (q4 << 2)) << 4) + ((p4 & 1) << 3);
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int RA = AT ? 2 : 1, RBn = BT ? 2 : 1;
  constexpr int PER_KS = TM * RA + TN * RBn;
  auto compute = [&](auto st_) {
    constexpr int ST = decltype(st_)::value;
    bf16x8 af[2][TM], bfr[2][TN];
    unsigned a_addr[NA], b_addr[NB];                         // (the stage offset does not fit the 16-bit immediate for stage 2)
#pragma unroll
    for (int i = 0; i < NA; ++i) a_addr[i] = a_addr0[i] + ST * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NB; ++i) b_addr[i] = b_addr0[i] + ST * STAGE_BYTES;
    auto read_ks = [&](auto k2_) {                           // k-step 2 grp + K2 of the tile
      constexpr int K2 = decltype(k2_)::value;
      gsfor<0, TM>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        if constexpr (!AT) {
          af[K2][I] = gemm_lds128<I * 32 * 128>(a_addr[K2]);
        } else {
          constexpr int RB = BM * 2;
          af[K2][I] = join8(gemm_ldstr<K2 * 16 * RB>(a_addr[I]), gemm_ldstr<(K2 * 16 + 4) * RB>(a_addr[I]));
        }
      });
      gsfor<0, TN>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        if constexpr (!BT) {
          bfr[K2][I] = gemm_lds128<I * 32 * 128>(b_addr[K2]);
        } else {
          constexpr int RB = BN * 2;
          bfr[K2][I] = join8(gemm_ldstr<K2 * 16 * RB>(b_addr[I]), gemm_ldstr<(K2 * 16 + 4) * RB>(b_addr[I]));
        }
      });
    };
    read_ks(IC<0>{});
    read_ks(IC<1>{});
    gsfor<0, 2>([&](auto k2_) {
      constexpr int K2 = decltype(k2_)::value;
      constexpr int LEFT = K2 == 0 ? PER_KS : 0;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(af[K2][0]), "+v"(af[K2][1]), "+v"(bfr[K2][0]), "+v"(bfr[K2][1]) : "n"(LEFT));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[K2][mi], bfr[K2][ni], acc[mi][ni], 0, 0, 0);
    });
  };

  constexpr int PT = PA + PB;
  auto step = [&](auto st_, const int t) {
    constexpr int ST = decltype(st_)::value;
    constexpr int NXT = (ST + NS - 1) % NS;
    const bool more = t + NS - 1 < nk;
    if (more) stage(IC<NXT>{}, t + NS - 1);
    compute(st_);
    if (t + 1 < nk) {
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  gsfor<0, NS - 1>([&](auto i_) {
    if (decltype(i_)::value < nk) stage(i_, decltype(i_)::value);
  });
  BMHRL_GSTAMP(1)
  if (nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_GSTAMP(2)
  for (int t0 = 0; t0 < nk; t0 += NS) {
    gsfor<0, NS>([&](auto i_) {
      if (t0 + decltype(i_)::value < nk) step(i_, t0 + decltype(i_)::value);
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();             // every wave is done with the stages
  asm volatile("" ::: "memory");
  BMHRL_GSTAMP(3)
  // ---- the two partial tiles: group 1 -> LDS (16 bytes per lane and slot: wave-private, conflict free) -> group 0
  f32x4* xch = reinterpret_cast<f32x4*>(sbase) + w4 * 16 * 64 + lane;
  if (grp == 1) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * g + e];
          xch[((mi * TN + ni) * 4 + g) * 64] = v;
        }
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = xch[((mi * TN + ni) * 4 + g) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][4 * g + e] += v[e];
      }
  __syncthreads();                          // (the four remaining waves: the epilogue stages through the same memory)
  gemm_epilogue<TM, TN, SMEM_ELEMS>(p, acc, smem, b1, b2, m0, n0, 0);
  BMHRL_GSTAMP(4)
}

// direct-to-LDS main loop: whole k-tiles only (no zero fill of a ragged reduction tail) and 32-bit lane offsets
bool uses_glds(const GemmArgs& a, int a_trans, int b_trans) {
  static const int no_glds = getenv("BMHRL_GEMM_NOGLDS") ? atoi(getenv("BMHRL_GEMM_NOGLDS")) : 0;       // (tuning aid: A/B)
  const long a_span = (a_trans ? 64 : (long)a.M) * a.lda * 2, b_span = (b_trans ? 64 : (long)a.N) * a.ldb * 2;
  return !no_glds && a.K % BK == 0 && a_span < (1l << 31) && b_span < (1l << 31) && a.M >= 8 && a.N >= 8;
}

template <int TM, int TN>
hipError_t launch(const GemmArgs& a, int a_trans, int b_trans, int batch, int splits, hipStream_t s) {
  GemmArgs p = a;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.tiles_m = (a.M + BM - 1) / BM;
  const int tiles_n = (a.N + BN - 1) / BN;
  p.tiles_mn = p.tiles_m * tiles_n;
  p.xcd_chunk = p.group_m = 0;
  {
    // XCD-aware tile order (BMHRL_GEMM_XCD=1, tuning switch, off): patches of about sqrt(tiles / 8) row panels.  Measured r04
    // (tests/kbench/gemm_xcd_ab.sh): alone +-5 % either way (V qkv 36.8 -> 34.8 us, A qkv 32.8 -> 30.9, V qkv dX 45.8 -> 43.8,
    // V dW 21.1 -> 20.3; V qkv dW 48.1 -> 50.3, 8192^3 998 -> 1050 us); the captured step 4.873 / 4.899 ms without, 4.882 /
    // 4.868 with: the L2 fill traffic these GEMMs cause is not what limits them.
    static const int xcd = getenv("BMHRL_GEMM_XCD") ? atoi(getenv("BMHRL_GEMM_XCD")) : 0;
    if (xcd && p.tiles_mn % 8 == 0 && p.tiles_mn >= 64 && tiles_n >= 2) {
      const int chunk = p.tiles_mn / 8;
      int g = 1;
      while ((g + 1) * (g + 1) <= chunk && g + 1 <= p.tiles_m) ++g;     // floor(sqrt(chunk)), at most tiles_m
      while (g > 1 && chunk % g != 0) --g;                               // whole patches: g rows x chunk / g columns
      if (xcd >= 2) g = xcd <= p.tiles_m ? xcd : p.tiles_m;              // (forced group height)
      p.xcd_chunk = chunk;
      p.group_m = g;
    }
  }
  p.splits = splits;
  p.dbg = getenv("BMHRL_GEMM_DBG") ? atoi(getenv("BMHRL_GEMM_DBG")) : 0;
  const int ktiles = (a.K + BK - 1) / BK;
  p.k_per_split = ((ktiles + splits - 1) / splits) * BK;
  dim3 grid(p.tiles_m * tiles_n, splits, batch), block(256);
  const bool glds = uses_glds(a, a_trans, b_trans);
  if (glds) {
    // stages: 2 for the 128 x 128 tiles (two workgroups per CU cover each other's waits; four stages at one workgroup per
    // CU measured equal or slower on every shape of the step: these sizes are bound by the ~70 GB/s a CU gets from L2 into
    // LDS, not by exposed latency); 4 for the 64 x 64 tiles when the grid gives a CU at most two of them (16 KiB stages:
    // a step is only 4 MFMAs per wave, the loads run three steps ahead -- 1024 x 1024 x 4096 dW 34 -> 21 us; with more
    // workgroups per CU, or a K split, the smaller footprint of two stages wins: 3072 x 1024 x 4096 dW 42 vs 49 us)
    static const int force_ns = getenv("BMHRL_GEMM_STAGES") ? atoi(getenv("BMHRL_GEMM_STAGES")) : 0;   // (tuning aid)
    const long blocks = (long)p.tiles_m * tiles_n * batch;
    // (Eight stages for the caption-side GEMMs -- 480 rows, 40 .. 128 tiles -- measured no better than four: 8.2 vs 7.3 us
    // at 480 x 1024 x 1024; those launches sit on their fixed costs, not on the depth of the ring.)
    int ns = TM == 1 && splits == 1 && blocks <= 448 ? 4 : 2;
    if (force_ns) ns = TM == 1 ? (force_ns >= 4 ? 4 : 2) : 2;
    // eight waves per 128 x 128 tile (gemm_glds8_kernel) when a CU gets at most one tile: BMHRL_GEMM_W8 = 0 (default) off, 1 on for
    // grids of up to W8_MAX workgroups, 2 always.  Alone it is the faster kernel on exactly those shapes (4096 x 1024 x 3072 dX
    // 45.0 -> 38.3 us, 4096 x 1024 x 1024 19.8 -> 18.8 us); inside the captured step, where the audio branch's kernels run beside
    // these GEMMs, the step measured 5.24 ms with it against 5.02 ms without: its 512-thread workgroups with 96 KiB of LDS leave
    // a CU no room for a workgroup of the other stream, which is worth more than the kernel's own time.  Off by default.
    static const int w8 = getenv("BMHRL_GEMM_W8") ? atoi(getenv("BMHRL_GEMM_W8")) : 0;
    static const long w8_max = getenv("BMHRL_GEMM_W8MAX") ? atol(getenv("BMHRL_GEMM_W8MAX")) : 256;
    if constexpr (TM == 2 && TN == 2) {
      if (w8 && splits == 1 && a.K >= 2 * BK && (w8 == 2 || blocks <= w8_max)) {
        dim3 block8(512);
        if (!a_trans && !b_trans) hipLaunchKernelGGL((gemm_glds8_kernel<false, false>), grid, block8, 0, s, p);
        else if (!a_trans && b_trans) hipLaunchKernelGGL((gemm_glds8_kernel<false, true>), grid, block8, 0, s, p);
        else if (a_trans && !b_trans) hipLaunchKernelGGL((gemm_glds8_kernel<true, false>), grid, block8, 0, s, p);
        else hipLaunchKernelGGL((gemm_glds8_kernel<true, true>), grid, block8, 0, s, p);
        return hipGetLastError();
      }
    }
#define BMHRL_GLDS(AT_, BT_)                                                                                    \
    do {                                                                                                          \
      if (TM == 1 && ns == 4) hipLaunchKernelGGL((gemm_glds_kernel<TM, TN, AT_, BT_, TM == 1 ? 4 : 2>), grid, block, 0, s, p); \
      else hipLaunchKernelGGL((gemm_glds_kernel<TM, TN, AT_, BT_, 2>), grid, block, 0, s, p);                   \
    } while (0)
    if (!a_trans && !b_trans) BMHRL_GLDS(false, false);
    else if (!a_trans && b_trans) BMHRL_GLDS(false, true);
    else if (a_trans && !b_trans) BMHRL_GLDS(true, false);
    else BMHRL_GLDS(true, true);
#undef BMHRL_GLDS
    return hipGetLastError();
  }
  if (!a_trans && !b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false>), grid, block, 0, s, p);
  else if (!a_trans && b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true>), grid, block, 0, s, p);
  else if (a_trans && !b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true>), grid, block, 0, s, p);
  return hipGetLastError();
}

// Tile size and K split of a problem -- ONE function for the launcher and for bmhrl_gemm_splits (callers that hand a weight
// gradient GEMM uninitialised memory must know for certain that it will not be accumulated into with atomics).
struct TilePlan { bool big, mid; int splits; };
TilePlan tile_plan(int M, int N, int K, int batch, bool can_split) {
  TilePlan t;
  const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  const long small_tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
  // split-K (fp32 atomics into a ZEROED C) for reductions much longer than the output is wide -- the weight
  // gradients dW = dY^T X.  Only plain fp32 outputs qualify and the caller must opt in (C zero-initialised).
  int splits = 1;
  static const int force_tile = getenv("BMHRL_GEMM_TILE") ? atoi(getenv("BMHRL_GEMM_TILE")) : 0;  // 1 = 64x64, 2 = 128x128 (tuning aid)
  static const long big_min = getenv("BMHRL_GEMM_BIGMIN") ? atol(getenv("BMHRL_GEMM_BIGMIN")) : 256;
  bool big = force_tile ? force_tile >= 2 : big_tiles >= big_min;
  // an output with at most `small_dim` rows or columns (the 30 caption positions against a memory: scores, contexts and
  // their gradients, batched over samples x heads) wastes 3/4 of a 128-wide tile where a 64-wide one wastes half
  static const int small_dim = getenv("BMHRL_GEMM_SMALLDIM") ? atoi(getenv("BMHRL_GEMM_SMALLDIM")) : 64;
  if (!force_tile && (M <= small_dim || N <= small_dim)) big = false;
  // (128x128 tiles + K split for small outputs: re-measured slower than 64x64 tiles + K split since the deep-prefetch /
  //  epilogue changes -- V dW 30 vs 23 us, A-out dW 25 vs 18 us; kept behind BMHRL_GEMM_BIGSPLIT=1 as a tuning aid)
  static const int big_split = getenv("BMHRL_GEMM_BIGSPLIT") ? atoi(getenv("BMHRL_GEMM_BIGSPLIT")) : 0;
  if (big_split && can_split && !force_tile && M >= 128 && N >= 128 && big_tiles <= 96 && K >= 1024) {
    // weight gradients with a small output and a long reduction: 128x128 tiles (about 3x the rate of 64x64 ones),
    // the chip is filled through the K split
    const int ktiles = (K + BK - 1) / BK;
    int s2 = (int)((256 + big_tiles - 1) / big_tiles);
    if (s2 > ktiles / 4) s2 = ktiles / 4;
    if (s2 >= 2) { big = true; splits = s2; }
  }
  if (splits == 1 && can_split && !big && small_tiles < 192) {
    const int ktiles = (K + BK - 1) / BK;
    splits = (int)((512 + small_tiles - 1) / small_tiles);
    if (splits > ktiles / 4) splits = ktiles / 4;   // >= 256 of K per split
    if (splits < 1) splits = 1;
  }
  // 128x128 tiles only when they still give every CU (256) a block; otherwise 64x64 tiles fill the chip better.
  // (256 x 128 tiles, one workgroup of 4 waves x 128 x 64 per CU with 512 registers per wave -- 48 KiB of operands per k-step for
  // twice the FLOPs of a square tile -- were built and measured SLOWER on every shape: 4096 x 3072 x 1024 48.5 vs 38.5 us,
  // 8192^3 1025 vs 1097 TF/s: what two co-resident workgroups hide for each other outweighs the lower traffic per FLOP.)
  // 128 x 64 tiles when 128 x 128 ones would give a CU at most one workgroup (the 4096 x 1024 projections of the video stream:
  // 256 tiles): two workgroups per CU cover each other's waits -- 4096 x 1024 x 1024 19.3 -> 17.0 us, its dX 16.1 -> 13.7 us;
  // with more columns (2048, 3072) the square tile's lower traffic per FLOP wins (26.5 vs 31.5 us).  Alone, that is: inside
  // the captured step, where the audio branch runs next to these GEMMs, the step measured 6.06 - 6.23 ms with them against
  // 5.95 ms without, so the option is off by default (BMHRL_GEMM_MIDMAX=256 turns it on, BMHRL_GEMM_TILE=3 forces it).
  static const int mid_max = getenv("BMHRL_GEMM_MIDMAX") ? atoi(getenv("BMHRL_GEMM_MIDMAX")) : 0;
  if (bmhrl_deterministic()) splits = 1;          // (atomics of a K split land in scheduling order)
  t.mid = force_tile ? force_tile == 3 : (big && splits == 1 && big_tiles <= mid_max);
  t.big = big;
  t.splits = splits;
  return t;
}

// BMHRL_DETERMINISTIC: the column sums the epilogue would add with one atomic per tile -- one thread per column walks the
// batches and rows of the OUTPUT in order instead (sums of the stored, rounded values).
__global__ __launch_bounds__(1024) void colsum_ordered_kernel(const GemmArgs p, int batch1) {
  // block = 64 columns x 16 row lanes; a thread takes every 16th row, eight loads in flight; the partial sums meet in a fixed order
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  for (int b1 = 0; b1 < batch1; ++b1)
    for (int b2 = 0; b2 < p.batch2; ++b2) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n < p.N) {
        if (p.Cb) {
          const bf16_t* c = p.Cb + b1 * p.cb_sb1 + b2 * p.cb_sb2 + n;
          for (int m = rl; m < p.M; m += 128)
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (m + 16 * j < p.M) a[j] += (float)c[(long)(m + 16 * j) * p.ldcb];
        } else {
          const float* c = p.C + b1 * p.c_sb1 + b2 * p.c_sb2 + n;
          for (int m = rl; m < p.M; m += 128)
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (m + 16 * j < p.M) a[j] += c[(long)(m + 16 * j) * p.ldc];
        }
      }
      __syncthreads();
      red[rl][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      __syncthreads();
      if (rl == 0 && n < p.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        p.colsum[b1 * p.cs_sb1 + b2 * p.cs_sb2 + n] += t;
      }
    }
}

// Second pass of the ordered K split: C = [C +] alpha * sum_s slab_s (+ bias) (+ residual), slabs added in split order.
// (n_live: the splits that got at least one k-tile -- ceil(k-tiles / splits) tiles each, so the last ones can be empty and
// their slabs unwritten)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p, int batch, int n_live) {
  const long per = (long)p.M * p.N, total = per * batch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long bz = i / per, r = i - bz * per;
    const int m = (int)(r / p.N), n = (int)(r - (long)m * p.N);
    const int b1 = (int)(bz / p.batch2), b2 = (int)(bz % p.batch2);
    const float* slab = p.split_ws + bz * p.splits * per + r;
    float x = 0.f;
    for (int sidx = 0; sidx < n_live; ++sidx) x += slab[sidx * per];
    x *= p.alpha;
    if (p.bias) x += p.bias[b1 * p.bias_sb1 + b2 * p.bias_sb2 + n];
    if (p.residual) x += p.residual[b1 * p.r_sb1 + b2 * p.r_sb2 + (long)m * p.ldr + n];
    float* dst = p.C + b1 * p.c_sb1 + b2 * p.c_sb2 + (long)m * p.ldc + n;
    *dst = p.accumulate ? *dst + x : x;
  }
}

}  // namespace

// how many K splits bmhrl_gemm uses for a plain fp32 (M, N) = A^T-style product with allow_split_k set: 1 means every element
// of C is written exactly once (C may be uninitialised), > 1 means fp32 atomics into a C that must be zero
extern "C" int bmhrl_gemm_splits(int32_t M, int32_t N, int32_t K, int32_t batch) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return -1;
  return tile_plan(M, N, K, batch, true).splits;
}

namespace {
// argument checks + the kernel's argument block + tile / split decision of one problem
int prepare(const bmhrl_gemm_desc* d, GemmArgs& a, TilePlan& tp, int& batch) {
  BMHRL_CHECK_ARG(d && d->A && d->B && (d->C || d->Cb));
  BMHRL_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->batch1 > 0 && d->batch2 > 0);
  BMHRL_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0);
  BMHRL_CHECK_ARG(d->a_sb1 % 8 == 0 && d->a_sb2 % 8 == 0 && d->b_sb1 % 8 == 0 && d->b_sb2 % 8 == 0);
  BMHRL_CHECK_ARG(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0);
  BMHRL_CHECK_ARG(d->lda >= (d->a_trans ? ((d->M + 7) & ~7) : ((d->K + 7) & ~7)));
  BMHRL_CHECK_ARG(d->ldb >= (d->b_trans ? ((d->N + 7) & ~7) : ((d->K + 7) & ~7)));
  BMHRL_CHECK_ARG(d->epilogue >= 0 && d->epilogue <= 3);
  if (d->epilogue == BMHRL_EPI_PROB || d->epilogue == BMHRL_EPI_DSCORE) BMHRL_CHECK_ARG(d->rowvec != nullptr);
  if (d->epilogue == BMHRL_EPI_DSCORE || d->epilogue == BMHRL_EPI_RELU_BWD) BMHRL_CHECK_ARG(d->aux != nullptr);
  if (d->epilogue == BMHRL_EPI_PROB) BMHRL_CHECK_ARG(d->rowvec2 != nullptr);
  BMHRL_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f);
  a.M = d->M; a.N = d->N; a.K = d->K; a.batch2 = d->batch2;
  a.A = (const bf16_t*)d->A; a.lda = d->lda; a.a_sb1 = d->a_sb1; a.a_sb2 = d->a_sb2;
  a.B = (const bf16_t*)d->B; a.ldb = d->ldb; a.b_sb1 = d->b_sb1; a.b_sb2 = d->b_sb2;
  a.C = d->C; a.ldc = d->ldc; a.c_sb1 = d->c_sb1; a.c_sb2 = d->c_sb2;
  a.Cb = (bf16_t*)d->Cb; a.ldcb = d->ldcb; a.cb_sb1 = d->cb_sb1; a.cb_sb2 = d->cb_sb2;
  a.epilogue = d->epilogue; a.alpha = d->alpha; a.relu = d->relu; a.accumulate = d->accumulate;
  a.bias = d->bias;
  a.residual = d->residual; a.ldr = d->ldr; a.r_sb1 = d->r_sb1; a.r_sb2 = d->r_sb2;
  a.mask = d->mask; a.mask_sb1 = d->mask_sb1; a.mask_sm = d->mask_sm;
  a.rowvec = d->rowvec; a.rowvec2 = d->rowvec2; a.rv_sb1 = d->rv_sb1; a.rv_sb2 = d->rv_sb2;
  a.aux = (const bf16_t*)d->aux; a.ldaux = d->ldaux; a.aux_sb1 = d->aux_sb1; a.aux_sb2 = d->aux_sb2;
  a.dropout_p = d->dropout_p; a.seed = d->seed; a.seed_dev = d->seed_dev; a.tiles_m = 0;
  a.drop_sb1 = d->drop_sb1; a.drop_sb2 = d->drop_sb2; a.drop_sm = d->drop_sm;
  a.colsum = d->colsum; a.cs_sb2 = d->colsum_sb2; a.bias_sb2 = d->bias_sb2; a.cs_sb1 = d->colsum_sb1; a.bias_sb1 = d->bias_sb1;
  if (a.drop_sb1 == 0 && a.drop_sb2 == 0 && a.drop_sm == 0) {
    a.drop_sm = d->N; a.drop_sb2 = (long)d->M * d->N; a.drop_sb1 = a.drop_sb2 * d->batch2;
  }
  batch = d->batch1 * d->batch2;
  // vector (8/16-byte) epilogue accesses need aligned bases and leading dimensions
  auto al = [](const void* q, uintptr_t a) { return q == nullptr || ((uintptr_t)q % a) == 0; };
  a.vec_ok = al(d->C, 16) && d->ldc % 4 == 0 && d->c_sb1 % 4 == 0 && d->c_sb2 % 4 == 0 && al(d->Cb, 8) && d->ldcb % 4 == 0 &&
             d->cb_sb1 % 4 == 0 && d->cb_sb2 % 4 == 0 && al(d->residual, 16) && d->ldr % 4 == 0 && d->r_sb1 % 4 == 0 &&
             d->r_sb2 % 4 == 0 && al(d->aux, 8) && d->ldaux % 4 == 0 && d->aux_sb1 % 4 == 0 && d->aux_sb2 % 4 == 0 &&
             al(d->bias, 16) && d->bias_sb2 % 4 == 0 && d->bias_sb1 % 4 == 0;
  a.fast_bf16 = d->Cb && !d->C && d->epilogue == BMHRL_EPI_LINEAR && !d->mask && !d->residual && !d->aux && !d->colsum &&
                !d->accumulate && al(d->Cb, 16) && d->ldcb % 8 == 0 && d->cb_sb1 % 8 == 0 && d->cb_sb2 % 8 == 0;
  const bool out_ok = d->Cb && !d->C && !d->residual && !d->colsum && !d->accumulate && !d->bias && al(d->Cb, 16) &&
                      d->ldcb % 8 == 0 && d->cb_sb1 % 8 == 0 && d->cb_sb2 % 8 == 0 && d->rowvec;
  a.fast_pd = (d->epilogue == BMHRL_EPI_PROB && out_ok && d->rowvec2 && (!d->mask || d->mask_sm == 0) && !d->aux) ||
              (d->epilogue == BMHRL_EPI_DSCORE && out_ok && d->aux && al(d->aux, 16) && d->ldaux % 8 == 0 &&
               d->aux_sb1 % 8 == 0 && d->aux_sb2 % 8 == 0 && (!d->mask || d->mask_sm == 0));
  const bool can_split = d->allow_split_k && d->C && !d->Cb && d->epilogue == BMHRL_EPI_LINEAR && !d->relu && !d->mask &&
                         d->dropout_p == 0.f && (!d->accumulate || d->split_ws) && !d->colsum;
  tp = tile_plan(d->M, d->N, d->K, batch, can_split);
  // ordered K split: only with a workspace that holds every split's partial tile
  a.split_ws = (tp.splits > 1 && d->split_ws && d->split_ws_elems >= (int64_t)tp.splits * batch * d->M * d->N) ? d->split_ws : nullptr;
  if (tp.splits > 1 && d->accumulate && !a.split_ws) return -22;
  return 0;
}
}  // namespace

void gemm_trace_dump(const GemmArgs& a, hipStream_t stream) {
#ifdef BMHRL_GEMM_TRACE
  if (getenv("BMHRL_GEMM_TRACE")) {
    long long hh[2][8];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_gemm_trace), sizeof(hh));
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "gemm trace (%s block, M %d N %d K %d): setup %lld  first tile %lld  loop %lld  epilogue %lld  total %lld\n",
              w ? "last" : "first", a.M, a.N, a.K, hh[w][1] - hh[w][0], hh[w][2] - hh[w][1], hh[w][3] - hh[w][2], hh[w][4] - hh[w][3],
              hh[w][4] - hh[w][0]);
  }
#else
  (void)a; (void)stream;
#endif
}

extern "C" int bmhrl_gemm(const bmhrl_gemm_desc* d, bmhrl_stream_t stream) {
  GemmArgs a;
  TilePlan tp;
  int batch;
  if (const int rc = prepare(d, a, tp, batch)) return rc;
  const bool ordered_colsum = bmhrl_deterministic() && a.colsum != nullptr;
  GemmArgs full = a;
  if (ordered_colsum) a.colsum = nullptr;
  hipError_t e;
  if (tp.mid) e = launch<2, 1>(a, d->a_trans, d->b_trans, batch, tp.splits, (hipStream_t)stream);
  else if (tp.big) e = launch<2, 2>(a, d->a_trans, d->b_trans, batch, tp.splits, (hipStream_t)stream);
  else e = launch<1, 1>(a, d->a_trans, d->b_trans, batch, tp.splits, (hipStream_t)stream);
  if (e == hipSuccess && ordered_colsum) {
    hipLaunchKernelGGL(colsum_ordered_kernel, dim3((unsigned)((full.N + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, full, d->batch1);
    e = hipGetLastError();
  }
  if (e == hipSuccess && a.split_ws) {
    GemmArgs r = a;
    r.splits = tp.splits;
    const int ktiles = (a.K + BK - 1) / BK, tiles_per_split = (ktiles + tp.splits - 1) / tp.splits;
    const int n_live = (ktiles + tiles_per_split - 1) / tiles_per_split;
    const long total = (long)batch * a.M * a.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, r, batch,
                       n_live);
    e = hipGetLastError();
  }
  gemm_trace_dump(a, (hipStream_t)stream);
  return hip_status(e);
}

// n <= 4 independent problems as ONE launch when all of them are 64 x 64-tile problems of the same operand layout on the
// register-staged main loop (reductions that are not a multiple of 64: the caption-side weight gradients, K = B L); any other
// mix is launched one by one -- same results either way.
extern "C" int bmhrl_gemm_group(const bmhrl_gemm_desc* d, int32_t n, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(d && n >= 1);
  if (n > 4) {
    for (int i = 0; i < n; i += 4)
      if (const int rc = bmhrl_gemm_group(d + i, n - i < 4 ? n - i : 4, stream)) return rc;
    return 0;
  }
  GemmGroup g;
  GemmArgs* ps[4] = {&g.p0, &g.p1, &g.p2, &g.p3};
  bool same = n > 1;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    TilePlan tp;
    int batch;
    if (const int rc = prepare(d + i, *ps[i], tp, batch)) return rc;
    GemmArgs& p = *ps[i];
    same = same && !tp.big && !tp.mid && d[i].a_trans == d[0].a_trans && d[i].b_trans == d[0].b_trans &&
           !uses_glds(p, d[i].a_trans, d[i].b_trans) && p.split_ws == nullptr &&
           !(bmhrl_deterministic() && p.colsum != nullptr);      // (ordered column sums are a pass of bmhrl_gemm)
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_mn = p.tiles_m * ((p.N + 63) / 64);
    p.splits = tp.splits;
    p.dbg = 0;
    const int ktiles = (p.K + BK - 1) / BK;
    p.k_per_split = ((ktiles + tp.splits - 1) / tp.splits) * BK;
    g.first[i] = total;
    total += p.tiles_mn * tp.splits * batch;
  }
  if (!same) {
    for (int i = 0; i < n; ++i)
      if (const int rc = bmhrl_gemm(d + i, stream)) return rc;
    return 0;
  }
  for (int i = n; i < 5; ++i) g.first[i] = total;
  for (int i = n; i < 4; ++i) *ps[i] = g.p0;
  g.n = n;
  dim3 grid((unsigned)total), block(256);
  const hipStream_t s = (hipStream_t)stream;
  if (!d[0].a_trans && !d[0].b_trans) hipLaunchKernelGGL((gemm_group_kernel<1, 1, false, false>), grid, block, 0, s, g);
  else if (!d[0].a_trans && d[0].b_trans) hipLaunchKernelGGL((gemm_group_kernel<1, 1, false, true>), grid, block, 0, s, g);
  else if (d[0].a_trans && !d[0].b_trans) hipLaunchKernelGGL((gemm_group_kernel<1, 1, true, false>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_group_kernel<1, 1, true, true>), grid, block, 0, s, g);
  return hip_status(hipGetLastError());
}
