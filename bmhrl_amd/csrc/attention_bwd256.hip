// Backward of the head-dimension-256 attentions (video self attention, A<-V cross attention: model/multihead_attention.py:7-31
// under autograd, as used at model/bm_hrl_agent.py:358-377): the softmax part as ONE kernel.
//
// The GEMM path forms, per (sample, head), P = softmax-recompute(Q K^T) (a batched GEMM with the PROB epilogue), the row term
// delta (a pass over dO and O) and dS = P (dO V^T - delta) scale (a second batched GEMM with the DSCORE epilogue that reads P
// back): three launches around two products with d_k = 256 -- four k-steps each, "all epilogue" --, P read twice.  Both users
// have at most 256 keys, so a wave that owns 32 query rows can hold the WHOLE score row of its queries: S^T = K Q^T and
// dP^T = V dO^T for all (up to eight) 32-key tiles are 2 x 8 accumulator tiles = 256 registers, the query row sits on one lane
// (the forward kernel's transposed formulation, attention_fwd.h), and everything the softmax backward needs is then a lane's
// own data: P from the forward's statistics (row max, row sum: kept separately so that a fully masked row -- uniform over every
// key -- stays exact), delta = sum_k P dP (of the bf16-rounded P the dV product will use: the sum that makes the rows of dS
// cancel) with one lane^32 exchange, dS = P (dP - delta) scale with masked keys at exactly 0.  P and dS
// leave as bf16 (B, H, Sq, pad8(Sk)) through an LDS image (16-byte row stores); dQ / dK / dV stay the three GEMMs they were.
//
// Workgroup = 4 waves x 32 query rows of one (sample, head); K and V tiles (32 keys x 512 bytes each) go global -> LDS by
// direct-to-LDS loads, two stages, both read as row fragments (ds_read_b128, rows XOR-swizzled on the source address as the
// forward's K tiles); Q^T and dO^T fragments stay in registers for the whole launch.
#include "attention_fwd.h"

namespace {

template <unsigned OFF>
__device__ __forceinline__ bf16x8 asm_lds_b128(unsigned addr) {   // (asm: the compiler drains direct-to-LDS loads before LDS reads it sees)
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

struct PsArgs {
  const bf16_t* Q; long ldq;      // (B*Sq, ldq), head h at columns [h*256, h*256 + 256)
  const bf16_t* K; long ldk;      // (B*Sk, ldk)
  const bf16_t* V; long ldv;
  const bf16_t* dO; long lddo;    // gradient w.r.t. the pre-dropout attention output, (B*Sq, lddo)
  const float* row_max; const float* row_sum;   // (B, H, Sq), natural-log units (the forward's statistics)
  const uint8_t* mask; long mask_sb;            // key mask (B, Sk) or nullptr
  bf16_t* P; bf16_t* dS; long ldp;              // (B, H, Sq, ldp), ldp = pad8(Sk)
  int B, H, Sq, Sk;
  float scale;
  int q_tiles, xcd_map;
};

constexpr int PS_STAGE = 2 * 32 * 512;            // K tile + V tile
constexpr int PS_ROWB = 512 + 16;                 // bytes per row of a wave's output image (256 keys bf16, padded)
constexpr int PS_IMG = 4 * 32 * PS_ROWB;
constexpr int PS_LDS = 2 * PS_STAGE > PS_IMG ? 2 * PS_STAGE : PS_IMG;

template <int NTILES>
__global__ __launch_bounds__(256, 1) void attn_bwd_ps256_kernel(const PsArgs p) {
  constexpr int DK = 256;
  __shared__ __attribute__((aligned(16))) char smem_raw[PS_LDS];
  __shared__ unsigned s_keep[8];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  // the q-tiles of one (sample, head) stream the same K / V: block ids that differ by multiples of 8 share an XCD (one L2)
  int qt, bh;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q1 = idx / p.q_tiles;
    bh = xcd + 8 * q1;
    qt = idx - q1 * p.q_tiles;
  } else {
    qt = blockIdx.x % p.q_tiles;
    bh = blockIdx.x / p.q_tiles;
  }
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q_row = qt * 128 + wave * 32 + r32;
  const bool q_ok = q_row < p.Sq;
  const int qr = q_ok ? q_row : p.Sq - 1;
  constexpr int nt = NTILES;                                          // key tiles: ceil(Sk / 32) <= 8

  // ---- K / V staging: wave w fills rows [8 w, 8 w + 8) of both tiles, 1 KiB (2 rows) per instruction; lane l writes chunk
  // l % 32 of row 2 i + l / 32, which holds the row's logical chunk (l % 32) ^ (row & 15)
  const char* __restrict__ Kb = reinterpret_cast<const char*>(p.K + (long)b * p.Sk * p.ldk + hd * DK);
  const char* __restrict__ Vb = reinterpret_cast<const char*>(p.V + (long)b * p.Sk * p.ldv + hd * DK);
  const int hi = lane >> 5, pch = lane & 31;
  // per-lane byte offsets of the four pieces of a tile, fixed for the launch (a row's swizzle depends on row & 15 only, which
  // a tile step of 32 rows leaves alone): tile t adds t * 32 rows; rows behind the last key re-read key Sk - 1 (P is 0 there)
  const unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  int srow[4];
  unsigned sch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    srow[i] = wave * 8 + 2 * i + hi;
    sch[i] = (unsigned)((pch ^ (srow[i] & 15)) << 4);
  }
  auto stage = [&](const int t, const int st) {
    bf16_t* kd = reinterpret_cast<bf16_t*>(smem_raw + st * PS_STAGE) + wave * 8 * DK;
    bf16_t* vd = kd + 32 * DK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned key = (unsigned)min(32 * t + srow[i], p.Sk - 1);
      glds16<0>(Kb + (key * ldk2 + sch[i]), kd + i * 2 * DK);
      glds16<0>(Vb + (key * ldv2 + sch[i]), vd + i * 2 * DK);
    }
  };
  stage(0, 0);
  // keep bits of the (at most 256) keys: thread = key; balloted into LDS behind the last tile's products
  const bool keep_t = tid < p.Sk && (p.mask == nullptr || p.mask[(long)b * p.mask_sb + min(tid, p.Sk - 1)] != 0);

  // Q^T and dO^T fragments: lane (q = r32, h) holds row q, columns 16 st + 8 h .. + 8
  bf16x8 qf[16], dof[16];
  {
    const bf16_t* qp = p.Q + ((long)b * p.Sq + qr) * p.ldq + hd * DK + 8 * h;
    const bf16_t* dp = p.dO + ((long)b * p.Sq + qr) * p.lddo + hd * DK + 8 * h;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      qf[st] = *reinterpret_cast<const bf16x8*>(qp + 16 * st);
      dof[st] = *reinterpret_cast<const bf16x8*>(dp + 16 * st);
    }
  }
  const long si = ((long)b * p.H + hd) * p.Sq + qr;
  const float m_row = p.row_max[si], il_row = 1.f / p.row_sum[si];

  unsigned k_addr[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) k_addr[st] = lds0 + r32 * 512 + (((2 * st + h) ^ (r32 & 15)) << 4);

  f32x16 s[nt], dpv[nt];
#pragma unroll
  for (int t = 0; t < nt; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[t][r] = 0.f; dpv[t][r] = 0.f; }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<0, nt>([&](auto t_) {
    constexpr int T = decltype(t_)::value;
    {
      constexpr unsigned SOFF = (T & 1) * PS_STAGE;
      if constexpr (T + 1 < nt) stage(T + 1, (T + 1) & 1);
      // S^T(T) = K(T) Q^T, dP^T(T) = V(T) dO^T: 16 k-steps, the row fragments of both tiles read four steps at a time
      static_for<0, 4>([&](auto g_) {
        constexpr int G = decltype(g_)::value;
        bf16x8 kf[4], vf[4];
        static_for<0, 4>([&](auto u_) {
          constexpr int U = decltype(u_)::value, ST = 4 * G + U;
          kf[U] = asm_lds_b128<SOFF + (ST >= 8 ? 256 : 0)>(k_addr[ST & 7]);
          vf[U] = asm_lds_b128<SOFF + 32 * 512 + (ST >= 8 ? 256 : 0)>(k_addr[ST & 7]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
        BMHRL_SB();
        static_for<0, 4>([&](auto u_) {
          constexpr int U = decltype(u_)::value, ST = 4 * G + U;
          s[T] = BMHRL_MFMA16(kf[U], qf[ST], s[T], 0, 0, 0);
          dpv[T] = BMHRL_MFMA16(vf[U], dof[ST], dpv[T], 0, 0, 0);
        });
      });
      if constexpr (T == nt - 1) {
        const unsigned long long bal = __ballot(keep_t);
        if (lane == 0) { s_keep[2 * wave] = (unsigned)bal; s_keep[2 * wave + 1] = (unsigned)(bal >> 32); }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // tile T + 1 has landed everywhere; every wave is done with stage T
      asm volatile("" ::: "memory");
    }
  });

  // ---- softmax backward on the lane's own query row.  Key of register r of tile t: 32 t + 4 h + (r & 3) + 8 (r >> 2).
  // The per-element work is kept to: one fma (score -> exponent), one bit-field extract + one bit select (masked keys take the
  // exponent of the fill value), exp2, one multiply, the bf16 rounding, one fma for the row term; then subtract + two multiplies
  // for dS.  No per-key test is needed for dS: a masked key of a row that has any unmasked key has P = exp2(-1.4e9) = 0
  // exactly, and a fully masked row (row max == the fill value: the forward keeps it exact) gets the factor 0 as a whole.
  constexpr float LOG2E = 1.4426950408889634f;
  const float c1 = p.scale * LOG2E, mneg = -m_row * LOG2E;
  const float arg_masked = (NEG_MASK - m_row) * LOG2E;              // 0 for a fully masked row: uniform attention
  const float ds_scale = m_row < -5e8f ? 0.f : p.scale;             // masked_fill passes no gradient
  const unsigned tail_valid = (p.Sk & 31) ? ((1u << (p.Sk & 31)) - 1u) >> (4 * h) : 0xffffffffu;   // keys < Sk of the last tile
  float delta = 0.f;
#pragma unroll
  for (int t = 0; t < nt; ++t) {
    const unsigned kw = (unsigned)__builtin_amdgcn_readfirstlane(s_keep[t]) >> (4 * h);     // this lane's keys from bit 0
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2);
      const int keep = -(int)((kw >> c) & 1u);                                       // 0 / -1 (v_bfe_i32)
      const float a_k = fmaf(s[t][r], c1, mneg);
      const float arg = __int_as_float((__float_as_int(a_k) & keep) | (__float_as_int(arg_masked) & ~keep));
      float pr = __builtin_amdgcn_exp2f(arg) * il_row;
      if (t == nt - 1) pr = ((tail_valid >> c) & 1u) ? pr : 0.f;                     // rows behind the last key (clamped loads)
      pr = (float)(bf16_t)pr;                                                        // the value the dV product reads
      s[t][r] = pr;
      delta = fmaf(pr, dpv[t][r], delta);
    }
  }
  delta += __shfl_xor(delta, 32, 64);       // the two 32-lane halves hold disjoint keys of the same query row
#pragma unroll
  for (int t = 0; t < nt; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dpv[t][r] = (s[t][r] * ds_scale) * (dpv[t][r] - delta);

  // ---- P, then dS: bf16 rows through the wave's padded LDS image, whole 16-byte pieces of contiguous rows to global
  char* img = smem_raw + wave * 32 * PS_ROWB;
  const int q0 = qt * 128 + wave * 32;
  const int chunks = (int)(p.ldp >> 3);                                 // 16-byte pieces per output row (ldp = pad8(Sk) <= 256)
  auto emit = [&](f32x16 (&val)[nt], bf16_t* out) {
#pragma unroll
    for (int t = 0; t < nt; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (bf16_t)val[t][4 * g + e];
        *reinterpret_cast<bf16x4*>(img + r32 * PS_ROWB + (32 * t + 8 * g + 4 * h) * 2) = w;
      }
    // (a wave reads back what it wrote itself: the compiler's own lgkmcnt wait orders the two)
    bf16_t* ob = out + (((long)b * p.H + hd) * p.Sq + q0) * p.ldp;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * i + hi, ch = pch;
      if (ch < chunks && ch < 4 * nt && q0 + row < p.Sq) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(img + row * PS_ROWB + ch * 16);
        *reinterpret_cast<bf16x8*>(ob + (long)row * p.ldp + ch * 8) = v;
      }
    }
  };
  emit(s, p.P);
  emit(dpv, p.dS);
}

}  // namespace

extern "C" int bmhrl_attention_bwd_scores256_ok(int32_t Sq, int32_t Sk, int32_t dk, int64_t mask_sq) {
  return dk == 256 && Sk >= 1 && Sk <= 256 && Sq >= 1 && mask_sq == 0;
}

extern "C" int bmhrl_attention_bwd_scores256(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                             const void* dO, int64_t lddo, const float* row_max, const float* row_sum,
                                             const uint8_t* mask, int64_t mask_sb, void* P, void* dS, int64_t ldp, int32_t B,
                                             int32_t H, int32_t Sq, int32_t Sk, float scale, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(Q && K && V && dO && row_max && row_sum && P && dS);
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0 && Sk <= 256);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && ldp % 8 == 0 && ldp >= Sk && ldp <= 256);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * 256 && ldk >= (int64_t)H * 256 && ldv >= (int64_t)H * 256 && lddo >= (int64_t)H * 256);
  BMHRL_CHECK_ARG((int64_t)Sk * ldk * 2 < (1ll << 31) && (int64_t)Sk * ldv * 2 < (1ll << 31));   // 32-bit lane offsets
  BMHRL_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)dO | (uintptr_t)P | (uintptr_t)dS) & 15) == 0);
  PsArgs a;
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.dO = (const bf16_t*)dO; a.lddo = lddo; a.row_max = row_max; a.row_sum = row_sum; a.mask = mask; a.mask_sb = mask_sb;
  a.P = (bf16_t*)P; a.dS = (bf16_t*)dS; a.ldp = ldp; a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
  a.q_tiles = (Sq + 127) / 128;
  a.xcd_map = ((B * H) % 8 == 0 && getenv("BMHRL_PS256_PLAINMAP") == nullptr) ? 1 : 0;
  const dim3 grid((unsigned)(B * H * a.q_tiles)), block(256);
  switch ((Sk + 31) / 32) {
    case 1: hipLaunchKernelGGL(attn_bwd_ps256_kernel<1>, grid, block, 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(attn_bwd_ps256_kernel<2>, grid, block, 0, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL(attn_bwd_ps256_kernel<3>, grid, block, 0, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL(attn_bwd_ps256_kernel<4>, grid, block, 0, (hipStream_t)stream, a); break;
    case 5: hipLaunchKernelGGL(attn_bwd_ps256_kernel<5>, grid, block, 0, (hipStream_t)stream, a); break;
    case 6: hipLaunchKernelGGL(attn_bwd_ps256_kernel<6>, grid, block, 0, (hipStream_t)stream, a); break;
    case 7: hipLaunchKernelGGL(attn_bwd_ps256_kernel<7>, grid, block, 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(attn_bwd_ps256_kernel<8>, grid, block, 0, (hipStream_t)stream, a); break;
  }
  return hip_status(hipGetLastError());
}
