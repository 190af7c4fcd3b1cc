// Head-dimension-256 attention forward for memories of at most 256 keys (the A<-V cross attention: 800 audio queries against the
// 256 video keys; model/multihead_attention.py:7-31 as called at model/bm_hrl_agent.py:362-377) in the two-phase form of the
// score backward (attention_bwd256.hip): the whole score row of a query fits one lane, so the softmax is EXACT and lane-local --
// no running maximum, no lazy rescale, no per-tile ballots (the generic online-softmax loop of attention_fwd.h spends ≈ 280 VALU
// instructions per 32-key tile on them, 2.6 k cycles per tile against 1 k of MFMA).
//
//   phase A   S^T(t) = K(t) Q^T for every key tile t (<= 8) into 128 accumulator registers; Q^T fragments in registers; ALL K tiles
//             are requested in the prologue (one LDS stage per tile, direct-to-LDS loads, rows XOR-swizzled as the generic
//             kernel's K tiles): with one workgroup per CU a tile of look-ahead left every tile waiting for its load
//   softmax   p = exp2(score - row max) on the lane's own 128 values (masked keys take the fill value -1e9, keys behind the
//             memory are 0), one lane^32 exchange each for the maximum and the sum; p -> bf16 = the B operands of phase B
//             (registers 0-7 / 8-15 of a tile are the two 16-key steps, as in the generic kernel)
//   phase B   O^T += V^T(t) P^T(t): V(t) takes the stage of K(t) as soon as every wave is done with it (all V tiles are in flight
//             before the softmax starts), transposed reads (ds_read_b64_tr_b16) with the generic kernel's V swizzle and
//             addressing; O^T is scaled by 1 / row sum at the end (P stays un-normalised in bf16, as there)
//   epilogue  output dropout with the generic kernel's element numbering (same mask for the same seed), bf16 rows through a
//             padded LDS image, statistics in natural-log units (a fully masked row keeps the exact fill value)
//
// Workgroup = 4 waves x 64 query rows (TWO 32-row blocks per wave) of one (sample, head): every K / V fragment read from LDS feeds
// two MFMAs (the four waves of a workgroup each read every fragment: with one block per wave the LDS port was as busy as the
// matrix pipe), and the 800 audio queries of the A<-V attention become 4 workgroups per (sample, head) = 256 workgroups = ONE
// round of the chip instead of 448 = 1.75.  Both halves of the register file are used: S^T (2 x 8 tiles) and later O^T (2 x 8
// tiles) in the 256 accumulator registers, Q^T fragments / P^T in the 256 vector registers.  The q-tiles of a (sample, head)
// share an XCD.  Compiled per number of key tiles (runtime tile loops around 256 accumulator registers spill).
#include "attention_fwd.h"

namespace {

template <unsigned OFF>
__device__ __forceinline__ bf16x8 fs_lds_b128(unsigned addr) {   // (asm: the compiler drains direct-to-LDS loads before LDS reads it sees)
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

struct FsArgs {
  const bf16_t* Q; long ldq;
  const bf16_t* K; long ldk;
  const bf16_t* V; long ldv;
  bf16_t* O; long ldo;
  float* row_max; float* row_sum;
  const uint8_t* mask; long mask_sb;            // key mask (B, Sk) or nullptr
  int B, H, Sq, Sk;
  float scale, dropout_p; uint64_t seed; const uint64_t* seed_dev;
  int q_tiles, xcd_map;
};

constexpr int FS_TILE = 32 * 512;                 // one K or V tile: 32 keys x 512 bytes
constexpr int FS_VBASE = 0;                       // stage t holds K(t), later V(t)
constexpr int FS_ROWB = 512 + 16;                 // bytes per row of a wave's output image (256 columns bf16, padded)
constexpr int FS_IMG = 4 * 32 * FS_ROWB;
constexpr int FS_LDS = 8 * FS_TILE > FS_IMG ? 8 * FS_TILE : FS_IMG;

template <int NTILES>
__global__ __launch_bounds__(256, 1) void attn_fwd_sk256_kernel(const FsArgs p) {
  constexpr int DK = 256, nt = NTILES;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  __shared__ __attribute__((aligned(16))) char smem_raw[FS_LDS];
  __shared__ unsigned s_keep[8];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  BMHRL_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
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
  constexpr int QB = 2;                               // 32-row query blocks per wave
  const int q_base = qt * (128 * QB) + wave * (32 * QB);

  // ---- staging: wave w fills rows [8 w, 8 w + 8) of a tile, 1 KiB (2 rows) per instruction; K rows swizzled chunk ^= row & 15
  // (row fragments by ds_read_b128), V rows chunk ^= (row & 3) << 2 (transposed reads); rows behind the last key re-read key Sk - 1
  const char* __restrict__ Kb = reinterpret_cast<const char*>(p.K + (long)b * p.Sk * p.ldk + hd * DK);
  const char* __restrict__ Vb = reinterpret_cast<const char*>(p.V + (long)b * p.Sk * p.ldv + hd * DK);
  const int hi = lane >> 5, pch = lane & 31;
  const unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  int srow[4];
  unsigned kch[4], vch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    srow[i] = wave * 8 + 2 * i + hi;
    kch[i] = (unsigned)((pch ^ (srow[i] & 15)) << 4);
    vch[i] = (unsigned)((pch ^ ((srow[i] & 3) << 2)) << 4);
  }
  auto stage_k = [&](const int t, const int st) {
    bf16_t* kd = reinterpret_cast<bf16_t*>(smem_raw + st * FS_TILE) + wave * 8 * DK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16<0>(Kb + ((unsigned)min(32 * t + srow[i], p.Sk - 1) * ldk2 + kch[i]), kd + i * 2 * DK);
  };
  auto stage_v = [&](const int t, const int st) {
    bf16_t* vd = reinterpret_cast<bf16_t*>(smem_raw + FS_VBASE + st * FS_TILE) + wave * 8 * DK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16<0>(Vb + ((unsigned)min(32 * t + srow[i], p.Sk - 1) * ldv2 + vch[i]), vd + i * 2 * DK);
  };
  // keep bits of the (at most 256) keys: thread = key; balloted into LDS behind the last score tile
  const bool keep_t = tid < p.Sk && (p.mask == nullptr || p.mask[(long)b * p.mask_sb + min(tid, p.Sk - 1)] != 0);

  // Q^T fragments: lane (q = r32, h) holds row q, columns 16 st + 8 h .. + 8.  Read straight from global memory a fragment load
  // touches 32 rows x 32 bytes; instead the wave's 64 rows go through the (still empty) stage area by direct-to-LDS loads -- whole
  // 1 KiB pieces, rows swizzled as the K tiles -- and are read back as row fragments.  Wave w uses stages 2 w, 2 w + 1 for its
  // own rows only, so no barrier is needed before the reads; one is needed before the K tiles overwrite the area.
  bf16x8 qf[QB][16];
  {
    const char* __restrict__ Qb = reinterpret_cast<const char*>(p.Q + (long)b * p.Sq * p.ldq + hd * DK);
    const unsigned ldq2 = (unsigned)p.ldq * 2u;
    bf16_t* qd = reinterpret_cast<bf16_t*>(smem_raw) + wave * 64 * DK;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int row = 2 * i + hi;                                      // 0 .. 63 of the wave's rows
      const unsigned qrow = (unsigned)min(q_base + row, p.Sq - 1);
      glds16<0>(Qb + (qrow * ldq2 + (unsigned)((pch ^ (row & 15)) << 4)), qd + i * 2 * DK);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned qa = lds0 + wave * 64 * 512 + r32 * 512;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const unsigned a = qa + qb * 32 * 512 + (((2 * st + h) ^ (r32 & 15)) << 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(qf[qb][st]) : "v"(a));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int st = 0; st < 16; ++st) asm volatile("" : "+v"(qf[qb][st]));
    __builtin_amdgcn_s_barrier();                        // every wave has its fragments: the stage area is free for the K tiles
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int t = 0; t < nt; ++t) stage_k(t, t);
  unsigned k_addr[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) k_addr[st] = lds0 + r32 * 512 + (((2 * st + h) ^ (r32 & 15)) << 4);

  // ---- phase A: the score tiles
  f32x16 s[QB][nt];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int t = 0; t < nt; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qb][t][r] = 0.f;
  // Groups of four k-steps (4 LDS fragment reads -> 8 MFMAs, both query blocks), software-pipelined: the reads of group i + 1
  // are issued before the MFMAs of group i (LDS reads return in order: lgkmcnt(4) = "group i is here").  Entering tile T needs
  // K(T) in LDS for every wave: loads complete in order, issued behind K(T) are K(T + 1 ..) and V(0 .. T - 3), four pieces each;
  // the barrier also says every wave is past tile T - 2, so V(T - 2) may take that stage.
  auto enter_tile = [&](auto t_) {
    constexpr int T = decltype(t_)::value;
    constexpr int BEHIND = 4 * (nt - 1 - T) + 4 * (T >= 2 ? T - 2 : 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BEHIND) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (T >= 2) stage_v(T - 2, T - 2);
  };
  auto kread = [&](auto i_, bf16x8 (&kf)[4]) {
    constexpr int I = decltype(i_)::value, T = I / 4, G = I % 4;
    static_for<0, 4>([&](auto u_) {
      constexpr int U = decltype(u_)::value, ST = 4 * G + U;
      kf[U] = fs_lds_b128<(ST >= 8 ? 256 : 0)>(k_addr[ST & 7] + T * FS_TILE);      // (stage offsets pass the 16-bit immediate)
    });
  };
  auto kmfma = [&](auto i_, bf16x8 (&kf)[4]) {
    constexpr int I = decltype(i_)::value, T = I / 4, G = I % 4;
    static_for<0, 4>([&](auto u_) {
      constexpr int U = decltype(u_)::value, ST = 4 * G + U;
      s[0][T] = BMHRL_MFMA16(kf[U], qf[0][ST], s[0][T], 0, 0, 0);
      s[1][T] = BMHRL_MFMA16(kf[U], qf[1][ST], s[1][T], 0, 0, 0);
    });
  };
  bf16x8 kfa[4], kfb[4];
  BMHRL_STAMP(1)
  enter_tile(std::integral_constant<int, 0>{});
  BMHRL_STAMP(2)
  kread(std::integral_constant<int, 0>{}, kfa);
  static_for<0, 4 * nt>([&](auto i_) {
    constexpr int I = decltype(i_)::value;
    if constexpr (I + 1 < 4 * nt) {
      if constexpr ((I + 1) % 4 == 0) enter_tile(std::integral_constant<int, (I + 1) / 4>{});
      if constexpr (I % 2 == 0) {
        kread(std::integral_constant<int, I + 1>{}, kfb);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kfa[0]), "+v"(kfa[1]), "+v"(kfa[2]), "+v"(kfa[3]));
      } else {
        kread(std::integral_constant<int, I + 1>{}, kfa);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kfb[0]), "+v"(kfb[1]), "+v"(kfb[2]), "+v"(kfb[3]));
      }
    } else {
      if constexpr (I % 2 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kfa[0]), "+v"(kfa[1]), "+v"(kfa[2]), "+v"(kfa[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kfb[0]), "+v"(kfb[1]), "+v"(kfb[2]), "+v"(kfb[3]));
    }
    BMHRL_SB();
    if constexpr (I % 2 == 0) kmfma(i_, kfa);
    else kmfma(i_, kfb);
    BMHRL_SB();
  });
  BMHRL_STAMP(3)
  {
    const unsigned long long bal = __ballot(keep_t);
    if (lane == 0) { s_keep[2 * wave] = (unsigned)bal; s_keep[2 * wave + 1] = (unsigned)(bal >> 32); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // every wave is past the last score tile; the keep bits are in LDS
  asm volatile("" ::: "memory");
  if constexpr (nt >= 2) stage_v(nt - 2, nt - 2);
  stage_v(nt - 1, nt - 1);

  // ---- exact softmax on the lane's own row.  Key of register r of tile t: 32 t + 4 h + (r & 3) + 8 (r >> 2).
  const float c1 = p.scale * LOG2E;
  constexpr float FILL2 = NEG_MASK * LOG2E;
  const unsigned tail_valid = (p.Sk & 31) ? ((1u << (p.Sk & 31)) - 1u) >> (4 * h) : 0xffffffffu;
  float inv_l[QB];
  bf16x8 pf[QB][nt][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < nt; ++t) {
      const unsigned kw_all = (unsigned)__builtin_amdgcn_readfirstlane(s_keep[t]);
      if (kw_all == 0xffffffffu) {                          // (wave-uniform) no masked key in this tile: the usual case
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[qb][t][r] *= c1;
          m = fmaxf(m, s[qb][t][r]);
        }
      } else {
        const unsigned kw = kw_all >> (4 * h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const int keep = -(int)((kw >> c) & 1u);
          float a = __int_as_float((__float_as_int(s[qb][t][r] * c1) & keep) | (__float_as_int(FILL2) & ~keep));
          if (t == nt - 1) a = ((tail_valid >> c) & 1u) ? a : -INFINITY;        // rows behind the last key
          s[qb][t][r] = a;
          m = fmaxf(m, a);
        }
      }
    }
    m = pair_max(m);                                   // the other half of the row sits on lane ^ 32
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < nt; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[qb][t][r] - m);
        l += e;
        pf[qb][t][r >> 3][r & 7] = (bf16_t)e;
      }
    l += __shfl_xor(l, 32, 64);
    inv_l[qb] = __builtin_amdgcn_rcpf(l);
    // statistics in natural-log units (a fully masked row keeps the exact fill value); written here, not behind phase B: two
    // values fewer to carry through it
    const int q_row = q_base + 32 * qb + r32;
    if (h == 0 && q_row < p.Sq) {
      const long si = ((long)b * p.H + hd) * p.Sq + q_row;
      p.row_max[si] = (m <= FILL2) ? NEG_MASK : m * LN2;
      p.row_sum[si] = l;
    }
  }

  BMHRL_STAMP(4)
  // ---- phase B: O^T += V^T(t) P^T(t)
  f32x16 o[QB][8];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.f;
  unsigned v_addr[4];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const int lc = 4 * dd + 2 * g1 + (p4 >> 1), pc = lc ^ (q4 << 2);
    v_addr[dd] = lds0 + FS_VBASE + 2 * ((4 * h + q4) * DK) + (pc << 4) + ((p4 & 1) << 3);
  }
  // Quarter tiles (two d-tiles: 8 transposed reads -> 8 MFMAs, both query blocks), pipelined like phase A.  Entering tile T needs
  // V(T) in LDS for every wave: issued behind it are V(T + 1 ..).
  auto enter_v = [&](auto t_) {
    constexpr int T = decltype(t_)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (nt - 1 - T)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto vread = [&](auto j_, bf16x8 (&vf)[2][2]) {
    constexpr int J = decltype(j_)::value, T = J / 4, QTR = J % 4;
    constexpr unsigned HOFF = (QTR / 2) * 256;
    static_for<0, 2>([&](auto e_) {
      constexpr int DD = 2 * (QTR % 2) + decltype(e_)::value;
      const unsigned a0 = v_addr[DD] + T * FS_TILE;
      vf[decltype(e_)::value][0] = join8(asm_tr4<HOFF>(a0), asm_tr4<HOFF + 8 * DK * 2>(a0));
      vf[decltype(e_)::value][1] = join8(asm_tr4<HOFF + 16 * DK * 2>(a0), asm_tr4<HOFF + 24 * DK * 2>(a0));
    });
  };
  auto vmfma = [&](auto j_, bf16x8 (&vf)[2][2]) {
    constexpr int J = decltype(j_)::value, T = J / 4, QTR = J % 4;
    static_for<0, 4>([&](auto i_) {
      constexpr int I = decltype(i_)::value, DT = 4 * (QTR / 2) + 2 * (QTR % 2) + I / 2;
      o[0][DT] = BMHRL_MFMA16(vf[I / 2][I % 2], pf[0][T][I % 2], o[0][DT], 0, 0, 0);
      o[1][DT] = BMHRL_MFMA16(vf[I / 2][I % 2], pf[1][T][I % 2], o[1][DT], 0, 0, 0);
    });
  };
#define FS_VWAIT(N, V) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(V[0][0]), "+v"(V[0][1]), "+v"(V[1][0]), "+v"(V[1][1]))
  bf16x8 vfa[2][2], vfb[2][2];
  enter_v(std::integral_constant<int, 0>{});
  vread(std::integral_constant<int, 0>{}, vfa);
  static_for<0, 4 * nt>([&](auto j_) {
    constexpr int J = decltype(j_)::value;
    if constexpr (J + 1 < 4 * nt) {
      if constexpr ((J + 1) % 4 == 0) enter_v(std::integral_constant<int, (J + 1) / 4>{});
      if constexpr (J % 2 == 0) { vread(std::integral_constant<int, J + 1>{}, vfb); FS_VWAIT(8, vfa); }
      else { vread(std::integral_constant<int, J + 1>{}, vfa); FS_VWAIT(8, vfb); }
    } else {
      if constexpr (J % 2 == 0) FS_VWAIT(0, vfa);
      else FS_VWAIT(0, vfb);
    }
    BMHRL_SB();
    if constexpr (J % 2 == 0) vmfma(j_, vfa);
    else vmfma(j_, vfb);
    BMHRL_SB();
  });
#undef FS_VWAIT
  BMHRL_STAMP(5)
  __builtin_amdgcn_s_barrier();                        // every wave is done with the stages: the output image takes their place
  asm volatile("" ::: "memory");

  // ---- epilogue: 1 / row sum, output dropout, bf16 rows through the wave's padded image -- one 32-row block after the other
  {
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  char* img = smem_raw + wave * 32 * FS_ROWB;
  const bool drop = p.dropout_p > 0.f;
  // (the lane id is taken through an opaque asm here: otherwise the compiler forms the epilogue's 64-bit row addresses in the
  //  prologue and carries them -- spilled -- across both phases)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int r32 = lane_e & 31, h = lane_e >> 5, hi = lane_e >> 5, pch = lane_e & 31;
  // (DROP is a compile-time flag of the block: a uniform run-time test inside the 256-element loops is a scalar branch per element)
  auto finish = [&](auto drop_) {
    constexpr bool DROP = decltype(drop_)::value;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q0 = q_base + 32 * qb, q_row = q0 + r32;
      const float inv = inv_l[qb];
      const uint64_t ebase = ((uint64_t)b * p.Sq + q_row) * (uint64_t)(p.H * DK) + hd * DK + 4 * h;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = o[qb][dt][4 * g + j] * inv;
            if constexpr (DROP) x *= dropout_scale(p.dropout_p, seed, ebase + 32 * dt + 8 * g + j);
            w[j] = (bf16_t)x;
          }
          *reinterpret_cast<bf16x4*>(img + r32 * FS_ROWB + (dt * 32 + 8 * g + 4 * h) * 2) = w;
        }
      // (a wave reads back what it wrote itself: the compiler's own lgkmcnt wait orders the two)
      bf16_t* ob = p.O + ((long)b * p.Sq + q0) * p.ldo + hd * DK;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 2 * i + hi;
        if (q0 + row < p.Sq)
          *reinterpret_cast<bf16x8*>(ob + (long)row * p.ldo + pch * 8) = *reinterpret_cast<const bf16x8*>(img + row * FS_ROWB + pch * 16);
      }
    }
  };
  if (drop) finish(std::integral_constant<bool, true>{});
  else finish(std::integral_constant<bool, false>{});
  }
  BMHRL_STAMP(6)
}

}  // namespace

// (C++ linkage: called from attention.hip's bmhrl_attention_fwd only)
bool bmhrl_attn256_sk_ok(int B, int H, int Sq, int Sk, long mask_sq, bool force) {
  static const int mode = getenv("BMHRL_ATTN_SK256") ? atoi(getenv("BMHRL_ATTN_SK256")) : 1;     // 0 off, 1 automatic, 2 whenever it can
  if (Sk > 256 || Sk < 1 || mask_sq != 0) return false;
  if (force) return true;
  // one 256-row workgroup per CU: worth it once the grid fills most of the chip (the A<-V shape: 256 workgroups)
  return mode == 2 || (mode == 1 && (long)B * H * ((Sq + 255) / 256) >= 200);
}

int bmhrl_attn256_sk_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo,
                         float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb, int32_t B, int32_t H, int32_t Sq,
                         int32_t Sk, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, hipStream_t stream) {
  BMHRL_CHECK_ARG(Q && K && V && O && row_max && row_sum && B > 0 && H > 0 && Sq > 0 && Sk > 0 && Sk <= 256);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * 256 && ldk >= (int64_t)H * 256 && ldv >= (int64_t)H * 256 && ldo >= (int64_t)H * 256);
  BMHRL_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0);
  BMHRL_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  BMHRL_CHECK_ARG((int64_t)Sk * ldk * 2 < (1ll << 31) && (int64_t)Sk * ldv * 2 < (1ll << 31) && (int64_t)Sq * ldq * 2 < (1ll << 31));   // 32-bit lane offsets
  FsArgs a;
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.O = (bf16_t*)O; a.ldo = ldo; a.row_max = row_max; a.row_sum = row_sum; a.mask = mask; a.mask_sb = mask_sb;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev;
  a.q_tiles = (Sq + 255) / 256;
  a.xcd_map = ((B * H) % 8 == 0) ? 1 : 0;
  const dim3 grid((unsigned)(B * H * a.q_tiles)), block(256);
  switch ((Sk + 31) / 32) {
    case 1: hipLaunchKernelGGL(attn_fwd_sk256_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(attn_fwd_sk256_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(attn_fwd_sk256_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(attn_fwd_sk256_kernel<4>, grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL(attn_fwd_sk256_kernel<5>, grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL(attn_fwd_sk256_kernel<6>, grid, block, 0, stream, a); break;
    case 7: hipLaunchKernelGGL(attn_fwd_sk256_kernel<7>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(attn_fwd_sk256_kernel<8>, grid, block, 0, stream, a); break;
  }
  attn_trace_dump("attn256 two-phase", Sq, Sk, stream, 6, 6);
  return hip_status(hipGetLastError());
}
