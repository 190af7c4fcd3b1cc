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
  float* colsum; long cs_sb2, bias_sb2;
  int tiles_m, splits, k_per_split, vec_ok, dbg, fast_bf16, fast_pd;
};

constexpr int BK = 64;

template <int TM, int TN, bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
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

  const int bz = blockIdx.z, b1 = bz / p.batch2, b2 = bz % p.batch2;
  const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ Ag = p.A + b1 * p.a_sb1 + b2 * p.a_sb2;
  const bf16_t* __restrict__ Bg = p.B + b1 * p.b_sb1 + b2 * p.b_sb2;
  const int M8 = (p.M + 7) & ~7, N8 = (p.N + 7) & ~7;
  // split-K: this block reduces k in [k_begin, k_end)
  const int k_begin = blockIdx.y * p.k_per_split;
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
    const float* __restrict__ bp = p.bias ? p.bias + b2 * p.bias_sb2 : nullptr;
    const uint64_t seedf = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
    const uint64_t dbase = (uint64_t)b1 * p.drop_sb1 + (uint64_t)b2 * p.drop_sb2;
    auto stage_bf16 = [&](const f32x16& av, const int mi, const int ni) {
      const int col = wn * 32 * TN + ni * 32 + r32, n = n0 + col;
      const float bias = (bp && n < p.N) ? bp[n] : 0.f;
      const int row0 = wm * 32 * TM + mi * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        float v = av[r] * p.alpha + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.dropout_p > 0.f) v *= dropout_scale(p.dropout_p, seedf, dbase + (uint64_t)(m0 + row) * p.drop_sm + n);
        sCb[row * SCB + col] = (bf16_t)v;
      }
    };
    stage_bf16(acc[0][0], 0, 0);
    if constexpr (TN > 1) stage_bf16(acc[0][1], 0, 1);
    if constexpr (TM > 1) {
      stage_bf16(acc[1][0], 1, 0);
      if constexpr (TN > 1) stage_bf16(acc[1][1], 1, 1);
    }
    __syncthreads();
    constexpr int G8 = BM * BN / 8 / 256;          // 16-byte groups per thread
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
  // Same idea for the two softmax epilogues of the attention backward (P recomputation, dS): the row vectors (row max /
  // 1 / row sum, or delta) are staged in LDS once per tile (the generic path re-loads and re-divides them per 4-column
  // group), the P tile that dS needs is brought in with 16-byte loads and updated in place, the result leaves as
  // 16-byte stores.  These GEMMs have K = d_k = 256, i.e. four k-steps: they are all epilogue.
  if (p.fast_pd && p.splits == 1) {
    constexpr int SCB = BN + 8;
    bf16_t* sCb = reinterpret_cast<bf16_t*>(smem);
    float* sRV = reinterpret_cast<float*>(smem + BM * SCB);
    float* sRV2 = sRV + BM;
    static_assert((BM * SCB) * 2 + 2 * BM * 4 <= 2 * (A_ELEMS + B_ELEMS) * 2, "softmax epilogue staging must fit");
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
    const uint8_t* __restrict__ Mk = (prob && p.mask) ? p.mask + b1 * p.mask_sb1 : nullptr;     // key mask (mask_sm == 0)
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
          v = (float)*q * (av[r] - r1[r >> 2][r & 3]) * p.alpha;
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
      finish(acc[mi][0], mi, 0, r1, r2);
      if constexpr (TN > 1) finish(acc[mi][1], mi, 1, r1, r2);
    };
    finish_rows(0);
    if constexpr (TM > 1) finish_rows(1);
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
  static_assert(BM * SC * 2 <= 2 * (A_ELEMS + B_ELEMS), "C tile must fit in the staging buffers");
  float* sC = reinterpret_cast<float*>(smem);
  auto stage = [&](const f32x16& av, const int mi, const int ni) {
    float* base = sC + (wm * 32 * TM + mi * 32 + 4 * h) * SC + wn * 32 * TN + ni * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * SC] = av[r];
  };
  stage(acc[0][0], 0, 0);
  if constexpr (TN > 1) stage(acc[0][1], 0, 1);
  if constexpr (TM > 1) {
    stage(acc[1][0], 1, 0);
    if constexpr (TN > 1) stage(acc[1][1], 1, 1);
  }
  __syncthreads();

  float* __restrict__ Cg = p.C ? p.C + b1 * p.c_sb1 + b2 * p.c_sb2 : nullptr;
  bf16_t* __restrict__ Cbg = p.Cb ? p.Cb + b1 * p.cb_sb1 + b2 * p.cb_sb2 : nullptr;
  const float* __restrict__ Rg = p.residual ? p.residual + b1 * p.r_sb1 + b2 * p.r_sb2 : nullptr;
  const uint8_t* __restrict__ Mg = p.mask ? p.mask + b1 * p.mask_sb1 : nullptr;
  const float* __restrict__ RVg = p.rowvec ? p.rowvec + b1 * p.rv_sb1 + b2 * p.rv_sb2 : nullptr;
  const float* __restrict__ RV2g = p.rowvec2 ? p.rowvec2 + b1 * p.rv_sb1 + b2 * p.rv_sb2 : nullptr;
  const bf16_t* __restrict__ AUXg = p.aux ? p.aux + b1 * p.aux_sb1 + b2 * p.aux_sb2 : nullptr;
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  const uint64_t drop_base = (uint64_t)b1 * p.drop_sb1 + (uint64_t)b2 * p.drop_sb2;
  const bool first_split = blockIdx.y == 0;
  const float* __restrict__ biasp = (first_split && p.bias) ? p.bias + b2 * p.bias_sb2 : nullptr;
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
      return aux * (x - rv) * p.alpha;
    } else {
      return aux > 0.f ? x * p.alpha : 0.f;
    }
  };

  constexpr int GROUPS = BM * BN / 4 / 256;
  // optional column sums of the OUTPUT tile (the bias gradient of the layer whose dY this GEMM produces): a thread
  // always works on the same 4 columns (256 % (BN/4) == 0), so it keeps a private partial and the 256 / (BN/4) threads
  // sharing a column group are combined through LDS: one atomic per column per tile
  float* __restrict__ CSg = p.colsum ? p.colsum + b2 * p.cs_sb2 : nullptr;
  f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};
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
          if constexpr (decltype(kind)::value == 1 || decltype(kind)::value == 2) {
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

template <int TM, int TN>
hipError_t launch(const GemmArgs& a, int a_trans, int b_trans, int batch, int splits, hipStream_t s) {
  GemmArgs p = a;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.tiles_m = (a.M + BM - 1) / BM;
  const int tiles_n = (a.N + BN - 1) / BN;
  p.splits = splits;
  p.dbg = getenv("BMHRL_GEMM_DBG") ? atoi(getenv("BMHRL_GEMM_DBG")) : 0;
  const int ktiles = (a.K + BK - 1) / BK;
  p.k_per_split = ((ktiles + splits - 1) / splits) * BK;
  dim3 grid(p.tiles_m * tiles_n, splits, batch), block(256);
  if (!a_trans && !b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false>), grid, block, 0, s, p);
  else if (!a_trans && b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true>), grid, block, 0, s, p);
  else if (a_trans && !b_trans) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true>), grid, block, 0, s, p);
  return hipGetLastError();
}

}  // namespace

extern "C" int bmhrl_gemm(const bmhrl_gemm_desc* d, bmhrl_stream_t stream) {
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
  GemmArgs a;
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
  a.colsum = d->colsum; a.cs_sb2 = d->colsum_sb2; a.bias_sb2 = d->bias_sb2;
  if (a.drop_sb1 == 0 && a.drop_sb2 == 0 && a.drop_sm == 0) {
    a.drop_sm = d->N; a.drop_sb2 = (long)d->M * d->N; a.drop_sb1 = a.drop_sb2 * d->batch2;
  }
  const int batch = d->batch1 * d->batch2;
  // vector (8/16-byte) epilogue accesses need aligned bases and leading dimensions
  auto al = [](const void* q, uintptr_t a) { return q == nullptr || ((uintptr_t)q % a) == 0; };
  a.vec_ok = al(d->C, 16) && d->ldc % 4 == 0 && d->c_sb1 % 4 == 0 && d->c_sb2 % 4 == 0 && al(d->Cb, 8) && d->ldcb % 4 == 0 &&
             d->cb_sb1 % 4 == 0 && d->cb_sb2 % 4 == 0 && al(d->residual, 16) && d->ldr % 4 == 0 && d->r_sb1 % 4 == 0 &&
             d->r_sb2 % 4 == 0 && al(d->aux, 8) && d->ldaux % 4 == 0 && d->aux_sb1 % 4 == 0 && d->aux_sb2 % 4 == 0 &&
             al(d->bias, 16) && d->bias_sb2 % 4 == 0;
  a.fast_bf16 = d->Cb && !d->C && d->epilogue == BMHRL_EPI_LINEAR && !d->mask && !d->residual && !d->aux && !d->colsum &&
                !d->accumulate && al(d->Cb, 16) && d->ldcb % 8 == 0 && d->cb_sb1 % 8 == 0 && d->cb_sb2 % 8 == 0;
  const bool out_ok = d->Cb && !d->C && !d->residual && !d->colsum && !d->accumulate && !d->bias && al(d->Cb, 16) &&
                      d->ldcb % 8 == 0 && d->cb_sb1 % 8 == 0 && d->cb_sb2 % 8 == 0 && d->rowvec;
  a.fast_pd = (d->epilogue == BMHRL_EPI_PROB && out_ok && d->rowvec2 && (!d->mask || d->mask_sm == 0) && !d->aux) ||
              (d->epilogue == BMHRL_EPI_DSCORE && out_ok && d->aux && al(d->aux, 16) && d->ldaux % 8 == 0 &&
               d->aux_sb1 % 8 == 0 && d->aux_sb2 % 8 == 0);
  const long big_tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch;
  const long small_tiles = (long)((d->M + 63) / 64) * ((d->N + 63) / 64) * batch;
  // split-K (fp32 atomics into a ZEROED C) for reductions much longer than the output is wide -- the weight
  // gradients dW = dY^T X.  Only plain fp32 outputs qualify and the caller must opt in (C zero-initialised).
  int splits = 1;
  const bool can_split = d->allow_split_k && d->C && !d->Cb && d->epilogue == BMHRL_EPI_LINEAR && !d->relu && !d->mask &&
                         d->dropout_p == 0.f && !d->accumulate && !d->colsum;
  static const int force_tile = getenv("BMHRL_GEMM_TILE") ? atoi(getenv("BMHRL_GEMM_TILE")) : 0;  // 1 = 64x64, 2 = 128x128 (tuning aid)
  static const long big_min = getenv("BMHRL_GEMM_BIGMIN") ? atol(getenv("BMHRL_GEMM_BIGMIN")) : 256;
  bool big = force_tile ? force_tile == 2 : big_tiles >= big_min;
  // (128x128 tiles + K split for small outputs: re-measured slower than 64x64 tiles + K split since the deep-prefetch /
  //  epilogue changes -- V dW 30 vs 23 us, A-out dW 25 vs 18 us; kept behind BMHRL_GEMM_BIGSPLIT=1 as a tuning aid)
  static const int big_split = getenv("BMHRL_GEMM_BIGSPLIT") ? atoi(getenv("BMHRL_GEMM_BIGSPLIT")) : 0;
  if (big_split && can_split && !force_tile && d->M >= 128 && d->N >= 128 && big_tiles <= 96 && d->K >= 1024) {
    // weight gradients with a small output and a long reduction: 128x128 tiles (about 3x the rate of 64x64 ones),
    // the chip is filled through the K split
    const int ktiles = (d->K + BK - 1) / BK;
    int s2 = (int)((256 + big_tiles - 1) / big_tiles);
    if (s2 > ktiles / 4) s2 = ktiles / 4;
    if (s2 >= 2) { big = true; splits = s2; }
  }
  if (splits == 1 && can_split && !big && small_tiles < 192) {
    const int ktiles = (d->K + BK - 1) / BK;
    splits = (int)((512 + small_tiles - 1) / small_tiles);
    if (splits > ktiles / 4) splits = ktiles / 4;   // >= 256 of K per split
    if (splits < 1) splits = 1;
  }
  hipError_t e;
  // 128x128 tiles only when they still give every CU (256) a block; otherwise 64x64 tiles fill the chip better.
  if (big) e = launch<2, 2>(a, d->a_trans, d->b_trans, batch, splits, (hipStream_t)stream);
  else e = launch<1, 1>(a, d->a_trans, d->b_trans, batch, splits, (hipStream_t)stream);
  return hip_status(e);
}
