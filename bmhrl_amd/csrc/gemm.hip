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
  int tiles_m, splits, k_per_split, vec_ok;
};

constexpr int BK = 64;

template <int TM, int TN, bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  // LDS row strides (elements).  k-contiguous tiles: +8 (16 B) keeps ds_read_b128 conflict free;
  // transposed tiles ([k][m]): +32 (64 B) puts the 4 rows of a tr-read block on disjoint bank quarters.
  constexpr int SA = AT ? (BM + 32) : (BK + 8);
  constexpr int SB = BT ? (BN + 32) : (BK + 8);
  constexpr int ROWS_A = AT ? BK : BM, ROWS_B = BT ? BK : BN;
  constexpr int A_ELEMS = ROWS_A * SA, B_ELEMS = ROWS_B * SB;
  constexpr int CH_A = BM * BK / 8 / 256, CH_B = BN * BK / 8 / 256;  // 16-byte chunks per thread
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (A_ELEMS + B_ELEMS)];

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

  bf16x8 ra[CH_A], rb[CH_B];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < CH_A; ++i) {
      const int c = tid + i * 256;
      if (AT) {
        const int row = c / (BM / 8), mc = c % (BM / 8);
        const bool ok = (k0 + row < k_end) && (m0 + mc * 8 < M8);
        ra[i] = ok ? *reinterpret_cast<const bf16x8*>(Ag + (long)(k0 + row) * p.lda + m0 + mc * 8) : zero_bf16x8();
      } else {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        const bool ok = (m0 + row < p.M) && (k0 + kc * 8 < K8);
        ra[i] = ok ? *reinterpret_cast<const bf16x8*>(Ag + (long)(m0 + row) * p.lda + k0 + kc * 8) : zero_bf16x8();
      }
    }
#pragma unroll
    for (int i = 0; i < CH_B; ++i) {
      const int c = tid + i * 256;
      if (BT) {
        const int row = c / (BN / 8), nc = c % (BN / 8);
        const bool ok = (k0 + row < k_end) && (n0 + nc * 8 < N8);
        rb[i] = ok ? *reinterpret_cast<const bf16x8*>(Bg + (long)(k0 + row) * p.ldb + n0 + nc * 8) : zero_bf16x8();
      } else {
        const int row = c / (BK / 8), kc = c % (BK / 8);
        const bool ok = (n0 + row < p.N) && (k0 + kc * 8 < K8);
        rb[i] = ok ? *reinterpret_cast<const bf16x8*>(Bg + (long)(n0 + row) * p.ldb + k0 + kc * 8) : zero_bf16x8();
      }
    }
  };
  auto store_tiles = [&](int buf) {
    bf16_t* sA = smem + buf * (A_ELEMS + B_ELEMS);
    bf16_t* sB = sA + A_ELEMS;
#pragma unroll
    for (int i = 0; i < CH_A; ++i) {
      const int c = tid + i * 256;
      const int row = AT ? c / (BM / 8) : c / (BK / 8);
      const int col = AT ? c % (BM / 8) : c % (BK / 8);
      *reinterpret_cast<bf16x8*>(sA + row * SA + col * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CH_B; ++i) {
      const int c = tid + i * 256;
      const int row = BT ? c / (BN / 8) : c / (BK / 8);
      const int col = BT ? c % (BN / 8) : c % (BK / 8);
      *reinterpret_cast<bf16x8*>(sB + row * SB + col * 8) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (k_end - k_begin + BK - 1) / BK;
  if (nk <= 0) return;   // empty split (uniform for the whole block)
  load_tiles(k_begin);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(k_begin + (kt + 1) * BK);
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
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  The accumulators (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) go through
  // LDS so that every thread then owns 4 consecutive columns of a row: residual / aux / mask loads and the C stores
  // are 8- or 16-byte, fully coalesced accesses instead of 2- or 4-byte ones at a 32-lane stride.
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

  constexpr int GROUPS = BM * BN / 4 / 256;
#pragma unroll 4
  for (int i = 0; i < GROUPS; ++i) {
    const int g = tid + i * 256;
    const int row = g / (BN / 4), c4 = (g % (BN / 4)) * 4;
    const int m = m0 + row, n = n0 + c4;
    if (m >= p.M || n >= p.N) continue;
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(sC + row * SC + c4);
    const int nv = min(4, p.N - n);
    const bool vec = p.vec_ok && nv == 4;
    float v[4] = {a4[0], a4[1], a4[2], a4[3]};
    float bias[4] = {0.f, 0.f, 0.f, 0.f}, res[4] = {0.f, 0.f, 0.f, 0.f}, aux[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && first_split) {
      if (vec) { const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + n); bias[0] = t[0]; bias[1] = t[1]; bias[2] = t[2]; bias[3] = t[3]; }
      else for (int j = 0; j < nv; ++j) bias[j] = p.bias[n + j];
    }
    if (Rg && first_split) {
      const float* rp = Rg + (long)m * p.ldr + n;
      if (vec) { const f32x4 t = *reinterpret_cast<const f32x4*>(rp); res[0] = t[0]; res[1] = t[1]; res[2] = t[2]; res[3] = t[3]; }
      else for (int j = 0; j < nv; ++j) res[j] = rp[j];
    }
    if (AUXg) {
      const bf16_t* ap = AUXg + (long)m * p.ldaux + n;
      if (vec) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(ap); aux[0] = (float)t[0]; aux[1] = (float)t[1]; aux[2] = (float)t[2]; aux[3] = (float)t[3]; }
      else for (int j = 0; j < nv; ++j) aux[j] = (float)ap[j];
    }
    const float rv = RVg ? RVg[m] : 0.f;
    const float rv2 = RV2g ? 1.f / RV2g[m] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nv) break;
      float x = v[j];
      if (p.epilogue == BMHRL_EPI_LINEAR) {
        x = x * p.alpha + bias[j];
        if (Mg && !Mg[(long)m * p.mask_sm + n + j]) x = NEG_MASK;
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.dropout_p > 0.f) x *= dropout_scale(p.dropout_p, seed, drop_base + (uint64_t)m * p.drop_sm + n + j);
        x += res[j];
      } else if (p.epilogue == BMHRL_EPI_PROB) {
        x = x * p.alpha;
        if (Mg && !Mg[(long)m * p.mask_sm + n + j]) x = NEG_MASK;
        x = __expf(x - rv) * rv2;
      } else if (p.epilogue == BMHRL_EPI_DSCORE) {
        x = aux[j] * (x - rv) * p.alpha;
      } else {  // BMHRL_EPI_RELU_BWD
        x = aux[j] > 0.f ? x * p.alpha : 0.f;
      }
      v[j] = x;
    }
    if (Cg) {
      float* dst = Cg + (long)m * p.ldc + n;
      if (p.splits > 1) {
        for (int j = 0; j < nv; ++j) atomicAdd(dst + j, v[j]);
      } else if (vec) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if (p.accumulate) { const f32x4 t = *reinterpret_cast<const f32x4*>(dst); o += t; }
        *reinterpret_cast<f32x4*>(dst) = o;
      } else {
        for (int j = 0; j < nv; ++j) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
      }
    }
    if (Cbg) {
      bf16_t* dst = Cbg + (long)m * p.ldcb + n;
      if (vec) {
        bf16x4 o;
        o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
        *reinterpret_cast<bf16x4*>(dst) = o;
      } else {
        for (int j = 0; j < nv; ++j) dst[j] = (bf16_t)v[j];
      }
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
  if (a.drop_sb1 == 0 && a.drop_sb2 == 0 && a.drop_sm == 0) {
    a.drop_sm = d->N; a.drop_sb2 = (long)d->M * d->N; a.drop_sb1 = a.drop_sb2 * d->batch2;
  }
  const int batch = d->batch1 * d->batch2;
  // vector (8/16-byte) epilogue accesses need aligned bases and leading dimensions
  auto al = [](const void* q, uintptr_t a) { return q == nullptr || ((uintptr_t)q % a) == 0; };
  a.vec_ok = al(d->C, 16) && d->ldc % 4 == 0 && d->c_sb1 % 4 == 0 && d->c_sb2 % 4 == 0 && al(d->Cb, 8) && d->ldcb % 4 == 0 &&
             d->cb_sb1 % 4 == 0 && d->cb_sb2 % 4 == 0 && al(d->residual, 16) && d->ldr % 4 == 0 && d->r_sb1 % 4 == 0 &&
             d->r_sb2 % 4 == 0 && al(d->aux, 8) && d->ldaux % 4 == 0 && d->aux_sb1 % 4 == 0 && d->aux_sb2 % 4 == 0 &&
             al(d->bias, 16);
  const long big_tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch;
  const long small_tiles = (long)((d->M + 63) / 64) * ((d->N + 63) / 64) * batch;
  // split-K (fp32 atomics into a ZEROED C) for reductions much longer than the output is wide -- the weight
  // gradients dW = dY^T X.  Only plain fp32 outputs qualify and the caller must opt in (C zero-initialised).
  int splits = 1;
  const bool can_split = d->allow_split_k && d->C && !d->Cb && d->epilogue == BMHRL_EPI_LINEAR && !d->relu && !d->mask &&
                         d->dropout_p == 0.f && !d->accumulate;
  const bool big = big_tiles >= 256;
  if (can_split && !big && small_tiles < 384) {
    const int ktiles = (d->K + BK - 1) / BK;
    splits = (int)((512 + small_tiles - 1) / small_tiles);
    if (splits > ktiles / 4) splits = ktiles / 4;   // >= 256 of K per split
    if (splits < 1) splits = 1;
  }
  hipError_t e;
  // 128x128 tiles only when they still give every CU (256) a block; otherwise 64x64 tiles fill the chip better.
  if (big) e = launch<2, 2>(a, d->a_trans, d->b_trans, batch, 1, (hipStream_t)stream);
  else e = launch<1, 1>(a, d->a_trans, d->b_trans, batch, splits, (hipStream_t)stream);
  return hip_status(e);
}
