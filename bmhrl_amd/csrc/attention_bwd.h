// Fused attention BACKWARD for the absorbed-projection (head dimension 128, shared keys / values) form: the attentions
// whose keys / values are the 128-wide audio stream (V<-A cross attention, audio self attention).  Autograd of
//   S_h = scale * Qp_h X^T (masked_fill(mask == 0, -1e9)) ; P_h = softmax(S_h) ; Cx_h = P_h X
// (model/multihead_attention.py:7-31 in the form of DESIGN.md section 9), given dCx:
//   dP_h = dCx_h X^T ; delta = rowsum(dCx_h * Cx_h) ; dS_h = P_h * (dP_h - delta) * scale, 0 at masked keys
//   dQp_h = dS_h X                                    (bmhrl_attention_shared128_bwd_dq : attn_bwd_dq128_kernel)
//   dX    = sum_h ( P_h^T dCx_h + dS_h^T Qp_h )       (bmhrl_attention_shared128_bwd_dx : attn_bwd_dx128_kernel + reduce)
// P is recomputed per tile from the forward's softmax statistics (row max, row sum: kept separately so that a fully masked
// row -- uniform over every key -- stays exact) and neither P nor dS ever reaches HBM.  Two kernels instead of one with
// atomics: each product is reduced inside ONE workgroup's registers (dQp over the keys by the query-block workgroup, dX over
// the queries by the key-block workgroup), so the results are deterministic; S and dP are formed twice (7 instead of 5
// products), which is cheaper here than 26 .. 180 MB of fp32 atomic traffic at the ~1.3 TB/s the chip adds at.
//
// Both kernels keep the layout conventions of the forward kernel (attention_fwd.h): the reduced index of a product is an
// MFMA k index, S / dP accumulators are turned into the next product's B operand in registers (k order of the accumulator:
// 16 s + 8 (j >> 2) + 4 h + (j & 3)), tiles go global -> LDS by direct-to-LDS loads into ONE image per tile that serves
// row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16): chunk ^= ((row & 3) << 2) | ((row >> 2) & 3).
#pragma once
#include "attention_fwd.h"

namespace {

template <int OFF>
__device__ __forceinline__ f32x4 asm_ldsf4(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

struct AttnBwdArgs {
  const bf16_t* Q; long ldq;        // Qp  (B, L, H, 128)
  const bf16_t* X; long ldx;        // X   (B, Sk, 128)
  const bf16_t* dO; long lddo;      // dCx (B, L, H, 128)
  bf16_t* dQ; long lddq;            // out (dq kernel): dQp (B, L, H, 128)
  float* dXp;                       // out (dx kernel): per-head partials (B, H, Sk, 128)
  const float* row_max; const float* row_sum; const float* delta;     // (B, H, L); delta = sum_d dCx * Cx
  const uint8_t* mask; long mask_sb;
  int B, H, Sq, Sk;
  float scale;
  int q_tiles, map_mode, per_b;
  unsigned magic_perb, magic_qt, magic_h;
};

// ------------------------------------------------------------------------------------------------------------------ dQ
// Query-centric, the forward kernel's structure: workgroup = QW x KW waves, wave (qi, ki) owns 32 query rows of one head and
// the keys [32 ki, 32 ki + 32) of every (32 KW)-key tile of X.  Per tile: S^T = X Qp^T and dP^T = X dCx^T (the SAME X row
// fragments feed both chains), dS^T in registers, dQp^T += X^T dS^T (transposed reads of the same image).  The key splits'
// partial sums are added through LDS at the end.
template <int QW, int KW, int NS>
__global__ __launch_bounds__(64 * QW * KW, 2) void attn_bwd_dq128_kernel(const AttnBwdArgs p) {
  constexpr int DK = 128;
  using C = AttnCfg<DK, QW, KW, NS, true>;
  constexpr int NW = C::NW, NT = C::NT, BN = C::BN, VST = C::VST, ND = C::ND, NH = C::NH;
  static_assert(NS >= 3, "tile t+1 has landed at the end of iteration t while tile t+NS-1 is in flight");
  __shared__ __attribute__((aligned(16))) char smem_raw[C::LDS_BYTES];
  uint64_t* s_slow = reinterpret_cast<uint64_t*>(smem_raw + C::MAIN_BYTES);
  uint64_t* s_valid = s_slow + C::WORDS;
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  constexpr float LOG2E = 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = wave_s / KW, ki = wave_s % KW;
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  int bh, qt;
  if (p.map_mode == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q1 = fast_div(idx, p.per_b, p.magic_perb), rem = idx - q1 * p.per_b;
    const int hq = fast_div(rem, p.q_tiles, p.magic_qt);
    bh = (xcd + 8 * q1) * p.H + hq;
    qt = rem - hq * p.q_tiles;
  } else {
    bh = fast_div((int)blockIdx.x, p.q_tiles, p.magic_qt);
    qt = (int)blockIdx.x - bh * p.q_tiles;
  }
  const int b = fast_div(bh, p.H, p.magic_h), hd = bh - b * p.H;
  const int q_row = qt * (32 * QW) + qi * 32 + r32;
  const bool q_ok = q_row < p.Sq;
  const char* __restrict__ Xgb = reinterpret_cast<const char*>(p.X + (long)b * p.Sk * p.ldx);
  const int nt_all = (p.Sk + BN - 1) / BN;
  const uint8_t* __restrict__ mrow_b = p.mask ? p.mask + (long)b * p.mask_sb : nullptr;
  const bool mask_al4 = (reinterpret_cast<uintptr_t>(mrow_b) & 3) == 0;

  // ---- staging of X tiles (as the forward kernel's shared image)
  constexpr int CPR = DK / 8, RPI = 64 / CPR, RW = BN / NW, GL = RW / RPI;
  static_assert(RW % RPI == 0 && GL >= 1, "a wave stages whole 1 KiB pieces");
  const int hi = lane / CPR, pch = lane % CPR;
  const int wrow = wave_s * RW;
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  auto issue_x = [&](const int t) {
    bf16_t* sdst = smem + (t % NS) * VST + wrow * DK;
    const int k0 = t * BN;
    if (k0 + BN <= p.Sk) {
      const char* base = Xgb + ((long)k0 + wrow) * p.ldx * 2;
      int hi_ = hi, pch_ = pch;
      asm volatile("" : "+v"(hi_), "+v"(pch_));     // the lane offsets are re-derived per tile (registers are scarce here)
      static_for<0, GL>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const int r = RPI * I + hi_;
        unsigned o = (unsigned)(r * (int)p.ldx * 2 + ((pch_ ^ swz(wrow + r)) << 4) - 1024 * (I & 3));
        asm volatile("" : "+v"(o));
        glds16<1024 * (I & 3)>(base + o, sdst + (I / 4) * 4 * RPI * DK);
      });
    } else {   // ragged last tile: clamp the key row (its dS is 0)
      int hi_ = hi, pch_ = pch;
      asm volatile("" : "+v"(hi_), "+v"(pch_));     // recomputed here: hoisted out of the loop these lane constants get spilled
      static_for<0, GL>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const int r = RPI * I + hi_;
        const int gr = min(k0 + wrow + r, p.Sk - 1);
        glds16<0>(Xgb + (unsigned)(gr * (int)p.ldx * 2 + ((pch_ ^ swz(wrow + r)) << 4)), sdst + RPI * I * DK);
      });
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nt_all) issue_x(s);

  // Qp^T / dCx^T fragments: lane (q = r32, h) holds row q, columns [16 step + 8 h, +8)
  bf16x8 qf[8], dof[8];
  {
    const long ro = ((long)b * p.Sq + (q_ok ? q_row : 0));
    const bf16_t* qp = p.Q + ro * p.ldq + hd * DK + 8 * h;
    const bf16_t* dp = p.dO + ro * p.lddo + hd * DK + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      qf[s] = q_ok ? *reinterpret_cast<const bf16x8*>(qp + 16 * s) : zero_bf16x8();
      dof[s] = q_ok ? *reinterpret_cast<const bf16x8*>(dp + 16 * s) : zero_bf16x8();
    }
  }
  // softmax statistics of this lane's query row: P = exp2(c s - m2) / l ; dS = P (dP - delta) scale
  float m2 = 0.f, invl_s = 0.f, dl_s = 0.f;
  if (q_ok) {
    const long si = ((long)b * p.H + hd) * p.Sq + q_row;
    m2 = p.row_max[si] * LOG2E;
    invl_s = p.scale / p.row_sum[si];
    dl_s = p.delta[si] * invl_s;
  }
  // key mask ballots (as the forward kernel): slow = 4-key groups with a masked / padding key, valid = with a valid key
  const int n_words = (nt_all * BN + 255) >> 8;
  for (int j = 0; j * NW < n_words; ++j) {
    const int i0 = 4 * (tid + NT * j);
    uint32_t v = 0x01010101u;
    if (mrow_b != nullptr && i0 < p.Sk) {
      if (mask_al4 && i0 + 4 <= p.Sk) {
        v = *reinterpret_cast<const uint32_t*>(mrow_b + i0);
      } else {
        v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < p.Sk) v |= (uint32_t)mrow_b[i0 + e] << (8 * e);
      }
    }
    bool any_slow = false, any_valid = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool keep = i0 + e < p.Sk && ((v >> (8 * e)) & 0xffu) != 0;
      any_slow |= !keep;
      any_valid |= keep;
    }
    const uint64_t bs = __ballot(any_slow), bv = __ballot(any_valid);
    if (lane == 0 && NW * j + wave_s < C::WORDS) {
      s_slow[NW * j + wave_s] = bs;
      s_valid[NW * j + wave_s] = bv;
    }
  }

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  // LDS addresses (stage 0): X^T fragments by transposed reads, X row fragments by ds_read_b128 (as the forward kernel).
  // Registers are the scarce resource here (dQp^T 64 + Qp^T 32 + dCx^T 32 stay resident): the 16 per-fragment addresses
  // are re-derived from two bases per tile (one XOR + shift each) instead of being kept.
  const unsigned vbase = lds0 + 2 * ((32 * ki + 4 * h + q4) * DK) + ((p4 & 1) << 3);
  const int vlc = 2 * g1 + (p4 >> 1), vsw0 = (q4 << 2) | (h & 3), vsw1 = (q4 << 2) | ((h + 2) & 3);
  const unsigned kbase = lds0 + 2 * ((32 * ki + r32) * DK);
  const int ksw = swz(r32);

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // tiles behind the last valid key contribute nothing (dS = 0 there); a batch row without any valid key: dS = 0 everywhere
  int nt = nt_all;
  if (mrow_b != nullptr) {
    int last_group = -1;
    for (int w = 0; w < n_words; ++w) {
      const uint64_t bv = s_valid[w];
      if (bv != 0ull) last_group = 64 * w + 63 - __builtin_clzll(bv);
    }
    nt = last_group >= 0 ? (4 * last_group) / BN + 1 : 0;
  }
  nt = __builtin_amdgcn_readfirstlane(nt);
  auto slow_window = [&](const int base) -> uint64_t {
    const int t = base + lane;
    const int g = t * KW + ki;
    const uint64_t w = (t < nt_all && (g >> 3) < C::WORDS) ? s_slow[g >> 3] : 0ull;
    return __ballot(((w >> (8 * (g & 7))) & 0xffull) != 0ull);
  };
  uint64_t slow_bits = slow_window(0);
  const float c_log2 = p.scale * LOG2E;

  bf16x8 kf[8], vf[4][2], pf[2];
  auto qk_issue = [&](const unsigned koffs) {
    unsigned kb = kbase + koffs;
    int sw = ksw, hh = h;
    asm volatile("" : "+v"(kb), "+v"(sw), "+v"(hh));        // (derived here, per tile: not hoisted into 8 loop-carried registers)
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const unsigned a = kb + (((2 * st + hh) ^ sw) << 4);
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(kf[st]) : "v"(a));
    }
  };
  auto read_vt = [&](const unsigned soff) {
    unsigned vb = vbase + soff;
    int lc = vlc, s0 = vsw0, s1 = vsw1;
    asm volatile("" : "+v"(vb), "+v"(lc), "+v"(s0), "+v"(s1));
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      const unsigned a0 = vb + (((4 * dd + lc) ^ s0) << 4), a1 = vb + (((4 * dd + lc) ^ s1) << 4);
      vf[dd][0] = join8(asm_tr4<0>(a0), asm_tr4<8 * DK * 2>(a1));
      vf[dd][1] = join8(asm_tr4<16 * DK * 2>(a0), asm_tr4<24 * DK * 2>(a1));
    }
  };
  for (int t = 0; t < nt; ++t) {
    if ((t & 63) == 0 && t > 0) slow_bits = slow_window(t);
    const unsigned soff = (unsigned)(t % NS) * (VST * 2);
    qk_issue(soff);                             // X(t) row fragments: their LDS latency passes under the load issue below
    const bool more = t + NS - 1 < nt;
    if (more) issue_x(t + NS - 1);              // into the stage of X(t-1): its last reads were in iteration t-1
    f32x16 s_acc, dp_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
    BMHRL_SB();
    static_for<0, 8>([&](auto st_) {
      constexpr int ST = decltype(st_)::value;
      s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ST], qf[ST], s_acc, 0, 0, 0);
      dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ST], dof[ST], dp_acc, 0, 0, 0);
    });
    BMHRL_SB();
    read_vt(soff);                              // X^T(t) (into the registers of the spent row fragments): its LDS latency
                                                // passes under the softmax arithmetic below
    // dS^T of this tile, in registers
    const bool slow = (slow_bits >> (t & 63)) & 1;
    f32x16& ds = s_acc;                         // in place
    if (!slow) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[r], c_log2, -m2));
        ds[r] = e * fmaf(dp_acc[r], invl_s, -dl_s);
      }
    } else {
      const int key0 = t * BN + 32 * ki + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t mk = 0x01010101u;
        const int kk = key0 + 8 * g;
        if (mrow_b != nullptr && kk < p.Sk) {
          if (mask_al4 && kk + 4 <= p.Sk) {
            mk = *reinterpret_cast<const uint32_t*>(mrow_b + kk);
          } else {
            mk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kk + e < p.Sk) mk |= (uint32_t)mrow_b[kk + e] << (8 * e);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool keep = kk + j < p.Sk && ((mk >> (8 * j)) & 0xffu) != 0;
          const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[4 * g + j], c_log2, -m2));
          ds[4 * g + j] = keep ? e * fmaf(dp_acc[4 * g + j], invl_s, -dl_s) : 0.f;   // no gradient through a masked_fill
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf[0][j] = (bf16_t)ds[j];
      pf[1][j] = (bf16_t)ds[8 + j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),
                   "+v"(vf[3][0]), "+v"(vf[3][1]));
    BMHRL_SB();
    static_for<0, 8>([&](auto i_) {
      constexpr int I = decltype(i_)::value;
      o[I / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[I / 2][I % 2], pf[I % 2], o[I / 2], 0, 0, 0);
    });
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();             // every wave is done with the stages: the merge below reuses them
  asm volatile("" ::: "memory");

  // ---- sum over the key splits + output (the forward kernel's exchange / image, without the softmax weights)
  constexpr int XSLOTS = C::XSLOTS, ROWB = C::ROWB;
  f32x4* xch = reinterpret_cast<f32x4*>(smem_raw);
  char* img = smem_raw + C::XCH_BYTES + C::ML_BYTES + wave_s * 32 * ROWB;
  auto slot = [&](int ks, int ko) { return ((qi * KW + ko) * (KW - 1) + (ks < ko ? ks : ks - 1)) * XSLOTS * 64 + lane; };
  if constexpr (KW > 1) {
    static_for<0, KW>([&](auto ko_) {
      constexpr int KO = decltype(ko_)::value;
      if (ki != KO) {
        f32x4* dst = xch + slot(ki, KO);
#pragma unroll
        for (int dd = 0; dd < NH; ++dd)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = o[KO * NH + dd][4 * g + j];
            dst[(dd * 4 + g) * 64] = v;
          }
      }
    });
    __syncthreads();
  }
  static_for<0, KW>([&](auto k_) {
    constexpr int KEEP = decltype(k_)::value;
    if (ki == KEEP) {
      constexpr int MY = KEEP * NH;
#pragma unroll
      for (int dd = 0; dd < NH; ++dd)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 acc;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = o[MY + dd][4 * g + j];
          if constexpr (KW > 1) {
#pragma unroll
            for (int k = 0; k < KW; ++k) {
              if (k == KEEP) continue;
              acc += (xch + slot(k, KEEP))[(dd * 4 + g) * 64];
            }
          }
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (bf16_t)acc[j];
          *reinterpret_cast<bf16x4*>(img + r32 * ROWB + (dd * 32 + 8 * g + 4 * h) * 2) = w;
        }
      constexpr int CPRO = NH * 4, RPS = 64 / CPRO, NST = 32 / RPS;
      const int srow = lane / CPRO, sch = lane % CPRO;
      const int q0 = qt * (32 * QW) + qi * 32;
      bf16x8 wout[NST];
#pragma unroll
      for (int i = 0; i < NST; ++i) wout[i] = *reinterpret_cast<const bf16x8*>(img + (RPS * i + srow) * ROWB + sch * 16);
      bf16_t* op = p.dQ + ((long)b * p.Sq + q0 + srow) * p.lddq + hd * DK + MY * 32 + sch * 8;
      const long ostep = (long)RPS * p.lddq;
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        if (q0 + RPS * i + srow < p.Sq) *reinterpret_cast<bf16x8*>(op) = wout[i];
        op += ostep;
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------------------------ dX
// Key-centric: workgroup = 4 waves = 128 keys of one (batch row, head); wave w keeps the X rows of its 32 keys as the
// (stationary) B operand of S = Qp X^T and dP = dCx X^T -- key on the lane -- and accumulates dX^T (128 x 32 keys, BOTH
// products of the head: dCx^T P + Qp^T dS) in 64 registers while the workgroup sweeps the query rows of the head, 32 at a
// time: the Qp and dCx tiles (32 x 128 each, one LDS image each for row and transposed reads) and the 32 rows' softmax
// statistics arrive by direct-to-LDS loads, NS stages.  Output: this head's partial (B, H, Sk, 128), summed over the heads by
// attn_bwd_dx_reduce_kernel (deterministic; 4 x 6.5 MB of partials at the reference shape).
constexpr int DX_WAVES = 4;
template <int NS>
struct DxCfg {
  static constexpr int TILE = 32 * 128 * 2;                        // bytes of one 32-row tile image
  static constexpr int STAT = 3 * 64 * 4;                          // row max, row sum, delta of 64 rows (32 used)
  static constexpr int STAGE = 2 * TILE + STAT;
  static constexpr int OUT = DX_WAVES * 32 * (512 + 16);           // fp32 output image (padded rows)
  static constexpr int LDS = cmax(NS * STAGE, OUT);
};

template <int NS>
__global__ __launch_bounds__(64 * DX_WAVES, 2) void attn_bwd_dx128_kernel(const AttnBwdArgs p) {
  constexpr int DK = 128;
  using C = DxCfg<NS>;
  __shared__ __attribute__((aligned(16))) char smem_raw[C::LDS];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int kblocks = (p.Sk + 32 * DX_WAVES - 1) / (32 * DX_WAVES);
  const int bh = blockIdx.x / kblocks, kb = blockIdx.x - bh * kblocks;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int key = kb * (32 * DX_WAVES) + wave_s * 32 + r32;          // this lane's key
  const bool key_in = key < p.Sk;
  const bool keep = key_in && (p.mask == nullptr || p.mask[(long)b * p.mask_sb + key] != 0);
  const int nqt = (p.Sq + 31) / 32;
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };

  // ---- staging: tile t = query rows [32 t, 32 t + 32) of head hd: Qp rows, dCx rows (8 pieces of 1 KiB each, 4 rows per
  // piece: wave w stages pieces 2 w, 2 w + 1 of both), and 64 floats of each statistic (wave 0..2, 4-byte direct loads)
  const int hi = lane >> 4, pch = lane & 15;
  const char* __restrict__ Qgb = reinterpret_cast<const char*>(p.Q + (long)b * p.Sq * p.ldq + hd * DK);
  const char* __restrict__ Dgb = reinterpret_cast<const char*>(p.dO + (long)b * p.Sq * p.lddo + hd * DK);
  const long sbase = ((long)b * p.H + hd) * p.Sq;
  auto issue = [&](const int t) {
    char* st = smem_raw + (t % NS) * C::STAGE;
    const int q0 = t * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * (2 * wave_s + i) + hi;                       // row inside the tile
      const int gr = min(q0 + r, p.Sq - 1);                          // (rows past the end: clamped, their P / dS are zeroed)
      const int sc = (pch ^ swz(r)) << 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Qgb + (long)gr * p.ldq * 2 + sc),
                                       (__attribute__((address_space(3))) void*)(st + (2 * wave_s + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Dgb + (long)gr * p.lddo * 2 + sc),
                                       (__attribute__((address_space(3))) void*)(st + C::TILE + (2 * wave_s + i) * 1024), 16, 0, 0);
    }
    if (wave_s < 3) {
      const float* src = wave_s == 0 ? p.row_max : (wave_s == 1 ? p.row_sum : p.delta);
      const int gq = min(q0 + lane, p.Sq - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + sbase + gq),
                                       (__attribute__((address_space(3))) void*)(st + 2 * C::TILE + wave_s * 256), 4, 0, 0);
    }
  };
  constexpr int PT = 5;        // direct-to-LDS instructions per wave and tile (waves 0..2; wave 3 issues 4: counted separately)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nqt) issue(s);

  // X rows of this lane's key: B operand of S and dP (lane (key = r32, h): X[key][16 step + 8 h, +8))
  bf16x8 xf[8];
  {
    const bf16_t* xp = p.X + ((long)b * p.Sk + min(key, p.Sk - 1)) * p.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xp + 16 * s);
  }
  f32x16 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  // LDS addresses (stage 0, Qp tile; the dCx tile is TILE bytes further)
  //   row fragments (A of S / dP): row r32 (query), chunk (2 st + h) ^ swz(r32)
  //   transposed fragments (A of the dX^T products): query rows 4 h + q4 (+16 ks, +8), chunk (4 dd + 2 g1 + (p4 >> 1)) ^ swz
  unsigned r_addr[8], t_addr[4][2];
#pragma unroll
  for (int st = 0; st < 8; ++st) r_addr[st] = lds0 + r32 * 256 + (((2 * st + h) ^ swz(r32)) << 4);
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
      const int lc = 4 * dd + 2 * g1 + (p4 >> 1);
      const int pc = lc ^ ((q4 << 2) | ((h + 2 * sec) & 3));
      t_addr[dd][sec] = lds0 + (4 * h + q4) * 256 + (pc << 4) + ((p4 & 1) << 3);
    }
  const unsigned s_addr = lds0 + 2 * C::TILE + (4 * h) * 4;          // statistics of rows 4 h + 8 g + {0..3}: + 32 g bytes

  const float c_log2 = p.scale * LOG2E;
  const bool wave3 = wave_s == 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int t = 0; t < nqt; ++t) {
    const bool more = t + NS - 1 < nqt;
    if (more) issue(t + NS - 1);
    const unsigned soff = (unsigned)(t % NS) * C::STAGE;
    bf16x8 af[8];
    f32x16 s_acc, dp_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; dp_acc[r] = 0.f; }
    // S = Qp X^T
#pragma unroll
    for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(af[st]) : "v"(r_addr[st] + soff));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
    BMHRL_SB();
    static_for<0, 8>([&](auto st_) { s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[decltype(st_)::value], xf[decltype(st_)::value], s_acc, 0, 0, 0); });
    // dP = dCx X^T (the row fragments reuse the registers)
    BMHRL_SB();
#pragma unroll
    for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[st]) : "v"(r_addr[st] + soff), "n"(C::TILE));
    // statistics of the 16 rows this lane's accumulators hold
    f32x4 smax[4], ssum[4], sdel[4];
    smax[0] = asm_ldsf4<0>(s_addr + soff);   smax[1] = asm_ldsf4<32>(s_addr + soff);
    smax[2] = asm_ldsf4<64>(s_addr + soff);  smax[3] = asm_ldsf4<96>(s_addr + soff);
    ssum[0] = asm_ldsf4<256>(s_addr + soff); ssum[1] = asm_ldsf4<288>(s_addr + soff);
    ssum[2] = asm_ldsf4<320>(s_addr + soff); ssum[3] = asm_ldsf4<352>(s_addr + soff);
    sdel[0] = asm_ldsf4<512>(s_addr + soff); sdel[1] = asm_ldsf4<544>(s_addr + soff);
    sdel[2] = asm_ldsf4<576>(s_addr + soff); sdel[3] = asm_ldsf4<608>(s_addr + soff);
    asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
    BMHRL_SB();
    static_for<0, 8>([&](auto st_) { dp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[decltype(st_)::value], xf[decltype(st_)::value], dp_acc, 0, 0, 0); });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(smax[0]), "+v"(smax[1]), "+v"(smax[2]), "+v"(smax[3]), "+v"(ssum[0]), "+v"(ssum[1]), "+v"(ssum[2]), "+v"(ssum[3]),
                   "+v"(sdel[0]), "+v"(sdel[1]), "+v"(sdel[2]), "+v"(sdel[3]));
    BMHRL_SB();
    // transposed fragments of dCx (for dCx^T P), issued before the softmax arithmetic
    bf16x8 tf[4][2];
    auto read_t = [&](const unsigned base) {
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const unsigned a0 = t_addr[dd][0] + base, a1 = t_addr[dd][1] + base;
        tf[dd][0] = join8(asm_tr4<0>(a0), asm_tr4<8 * 256>(a1));
        tf[dd][1] = join8(asm_tr4<16 * 256>(a0), asm_tr4<24 * 256>(a1));
      }
    };
    read_t(soff + C::TILE);
    // P and dS of the tile (rows = queries (registers), column = this lane's key)
    bf16x8 pb[2], dsb[2];
    const int q0 = t * 32 + 4 * h;
    const bool ragged = t * 32 + 32 > p.Sq;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const float rmax = smax[g][j];
        const float m2 = rmax * LOG2E;
        float invl = __builtin_amdgcn_rcpf(ssum[g][j]);
        if (ragged && q0 + 8 * g + j >= p.Sq) invl = 0.f;               // query rows past the end (clamped loads)
        const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[r], c_log2, -m2));
        // a masked key: exp(-1e9 - row max) -- 0, or 1 for a fully masked row (row max == -1e9: uniform over every key)
        const float pr = keep ? e * invl : (rmax == NEG_MASK ? invl : 0.f);
        const float dsv = keep ? pr * (dp_acc[r] - sdel[g][j]) * p.scale : 0.f;
        pb[r >> 3][r & 7] = (bf16_t)pr;
        dsb[r >> 3][r & 7] = (bf16_t)dsv;
      }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(tf[0][0]), "+v"(tf[0][1]), "+v"(tf[1][0]), "+v"(tf[1][1]), "+v"(tf[2][0]), "+v"(tf[2][1]), "+v"(tf[3][0]), "+v"(tf[3][1]));
    BMHRL_SB();
    static_for<0, 8>([&](auto i_) {
      constexpr int I = decltype(i_)::value;
      o[I / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[I / 2][I % 2], pb[I % 2], o[I / 2], 0, 0, 0);
    });
    BMHRL_SB();
    read_t(soff);                                  // Qp^T (for Qp^T dS)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(tf[0][0]), "+v"(tf[0][1]), "+v"(tf[1][0]), "+v"(tf[1][1]), "+v"(tf[2][0]), "+v"(tf[2][1]), "+v"(tf[3][0]), "+v"(tf[3][1]));
    BMHRL_SB();
    static_for<0, 8>([&](auto i_) {
      constexpr int I = decltype(i_)::value;
      o[I / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[I / 2][I % 2], dsb[I % 2], o[I / 2], 0, 0, 0);
    });
    // publish tile t+1 (the loads of tiles t+2 .. stay in flight), retire stage t
    if (more && NS > 2) {
      if (wave3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (PT - 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PT) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- output: dX^T accumulators (lane = key, registers = columns) through a padded fp32 image -> whole 512-byte rows
  constexpr int ROWB = 512 + 16;
  char* img = smem_raw + wave_s * 32 * ROWB;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = o[dd][4 * g + j];
      *reinterpret_cast<f32x4*>(img + r32 * ROWB + (dd * 32 + 8 * g + 4 * h) * 4) = v;
    }
  // a row = 32 pieces of 16 bytes: two rows per instruction, 16 instructions (a wave reads back what it wrote itself)
  const int srow = lane >> 5, sch = lane & 31;
  const int k0 = kb * (32 * DX_WAVES) + wave_s * 32;
  float* outp = p.dXp + (((long)b * p.H + hd) * p.Sk + k0 + srow) * DK + sch * 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (2 * i + srow) * ROWB + sch * 16);
    if (k0 + 2 * i + srow < p.Sk) *reinterpret_cast<f32x4*>(outp) = v;
    outp += 2 * DK;
  }
}

// out[b][k][:] (+)= sum_h part[b][h][k][:]  (fp32, 128 columns)
__global__ void attn_bwd_dx_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long ldo, int B, int H, int Sk,
                                          int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one f32x4 per thread
  const long total = (long)B * Sk * 32;
  if (i >= total) return;
  const int c4 = (int)(i % 32);
  const long bk = i / 32;
  const long bb = bk / Sk, k = bk % Sk;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int hh = 0; hh < H; ++hh) acc += *reinterpret_cast<const f32x4*>(part + ((bb * H + hh) * Sk + k) * 128 + c4 * 4);
  float* dst = out + (bb * Sk + k) * ldo + c4 * 4;
  if (accumulate) acc += *reinterpret_cast<const f32x4*>(dst);
  *reinterpret_cast<f32x4*>(dst) = acc;
}

}  // namespace
