// Fused scaled-dot-product attention forward for gfx950 -- the kernel template shared by attention.hip (head dimension
// 256) and attention128.hip (head dimension 128, absorbed-projection form; built with the VGPR form of the MFMAs).
//
// Fused scaled-dot-product attention forward for gfx950 (flash style, bf16 MFMA, fp32 softmax).
//
// Replaces attention() + head split/merge of the reference (model/multihead_attention.py:7-31,75-86) for the
// self- and cross-modal attentions of BMEncoderLayer and the caption->memory attentions of BMFusionLayer.
//
// Formulation (everything transposed so that a query row lives on ONE lane):
//   S^T (keys x q)  = K_tile . Q^T          A = K rows, B = Q^T fragments held in registers
//   softmax over keys = over the 16 accumulator registers of a lane + one lane^32 exchange (no LDS, no permute)
//   O^T (d x q)    += V^T . P^T             A = V^T read from the row-major V tile by ds_read_b64_tr_b16,
//                                           B = the S^T accumulator converted to bf16 in place (k order of the
//                                           accumulator: key = 16s + 8(j>>2) + 4h + (j&3))
// so the running max / sum / rescale factors are per-lane scalars and O^T rescaling needs no cross-lane traffic.
//
// Work split (template parameters, chosen on the host per shape): a workgroup is QW x KW waves; wave (qi, ki) owns 32
// query rows and the keys [32 ki, 32 ki + 32) of every (32 KW)-key tile.  KW = 1: a wave sees every key, no merge;
// KW > 1: the key splits keep private online-softmax states that are merged once at the end through LDS (every wave
// ends up with 1 / KW of the head's columns, so the combine, the normalisation and the stores are shared).  The shapes of
// the reference are small for a 256-CU chip (config 2: 16 384 .. 51 200 (row, head) pairs): the split is what fills it
// -- head dimension 256 (512 registers, one wave per SIMD): 4 x 1 when that gives >= ~200 workgroups, else 2 x 2;
// head dimension 128 (256 registers, two waves per SIMD): 4 x 1 with two workgroups per CU, 2 x 4 (eight waves) when
// there are few query rows.
//
// Head dimension DK: 256 = d_model / H of the reference; 128 = the absorbed-projection form of the attentions whose
// keys / values are the 128-wide audio stream (scores_h = (Q_h Wk_h) A^T, context_h = P_h A): ONE key/value tile shared
// by all heads AND by the two products -- a single LDS image serves the row reads of S^T and the transposed reads of
// O^T (chunk ^= ((row & 3) << 2) | ((row >> 2) & 3): both kinds of read are bank-conflict free).
//
// Data movement and schedule:
//   * K and V tiles go global -> LDS with direct-to-LDS loads (no staging registers, no ds_write), NS stages each,
//     issued at the top of an iteration; rows are XOR-swizzled (on the source address) so that the ds_read_b128 of K and
//     the transposed reads of V are bank-conflict free without padding; with NS >= 3 (NS >= 4 for the shared image) the
//     loads of an iteration stay in flight across its barrier (counted vmcnt);
//   * the loop is software pipelined inside a wave: the S^T MFMA chain of tile t+1 carries the exponentials of tile t,
//     the O^T MFMAs of tile t carry the row sum of tile t and the row max / exponential arguments of tile t+1; the
//     interleave is pinned in the source (sched_barrier between the slices), the compiler otherwise clusters the MFMAs
//     and runs the vector work after them;
//   * tiles whose keys are all valid and unmasked (nearly all) scale scores by one constant; masked / padding keys are a
//     rare wave-uniform path that reads the mask bytes of the lane's 16 keys straight from global memory (no per-key
//     coefficient arrays in LDS: no limit on Sk); fully masked tiles behind the last valid key of a batch row are not
//     visited at all (their probabilities are exactly 0 in fp32 as soon as one key of the row is valid);
//   * LDS reads inside the loop are inline asm with hand-counted lgkmcnt waits: the compiler orders every ds_read it
//     knows about behind ALL outstanding direct-to-LDS loads (it cannot tell the stages apart);
//   * Q fragments are pinned to the accumulator half of the register file, the O^T accumulators are handed to the rare
//     rescale as whole 16-register tuples, and the file is built with -amdgpu-codegenprepare-break-large-phis=false.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

struct AttnArgs {
  const bf16_t* Q; long ldq;
  const bf16_t* K; long ldk;
  const bf16_t* V; long ldv;
  bf16_t* O; long ldo;
  float* row_max; float* row_sum;
  const uint8_t* mask; long mask_sb, mask_sq;
  int B, H, Sq, Sk;
  long k_hs, v_hs;          // element offset of head h inside a K / V row: h * k_hs (0: one tile for all heads)
  float scale, dropout_p; uint64_t seed; const uint64_t* seed_dev;
  int q_tiles, dbg;
  // workgroup -> (batch row, head, q-tile) map, chosen on the host; divisions by multiply-high (exact: see div_magic)
  int map_mode, per_b;
  unsigned magic_perb, magic_qt, magic_h;
};

// floor(n / d) == umulhi(n, ceil(2^32 / d)) whenever n * d < 2^32 (d == 1: the magic does not fit, n itself)
inline unsigned div_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }
__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }
inline void set_block_map(AttnArgs& a) {
  a.per_b = a.H * a.q_tiles;
  a.map_mode = (a.k_hs == 0 && a.B % 8 == 0) ? 0 : ((a.B * a.H) % 8 == 0 ? 1 : 2);
  a.magic_perb = div_magic((unsigned)a.per_b);
  a.magic_qt = div_magic((unsigned)a.q_tiles);
  a.magic_h = div_magic((unsigned)a.H);
}

// LDS reads the compiler must not see (see the header); `addr` is a byte address in LDS.
template <int OFF>
__device__ __forceinline__ bf16x4 asm_tr4(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// 16 bytes per lane, global -> LDS; IMM is added to both the global and the LDS address
template <int IMM>
__device__ __forceinline__ void glds16(const char* src, bf16_t* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, IMM, 0);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ float pair_max(float v) {   // max over lanes l and l^32, on both
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#define BMHRL_SB() __builtin_amdgcn_sched_barrier(0)

// Tuning aid (-DBMHRL_ATTN_TRACE): cycle stamps of wave 0 of the first and the last workgroup at the phase boundaries.
#ifdef BMHRL_ATTN_TRACE
__device__ long long g_attn_trace[2][16];
#define BMHRL_STAMP(i)                                                                                   \
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                              \
    g_attn_trace[blockIdx.x != 0][i] = (long long)__builtin_readcyclecounter();
#else
#define BMHRL_STAMP(i)
#endif

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int DK, int QW, int KW, int NS, bool SHARED>
struct AttnCfg {
  static constexpr int NW = QW * KW, NT = 64 * NW, BN = 32 * KW;
  static constexpr int VST = BN * DK;                                 // elements of one stage of one operand
  static constexpr int NOPS = SHARED ? 1 : 2;
  static constexpr int STAGE_BYTES = NOPS * NS * VST * 2;
  static constexpr int ND = DK / 32, NH = ND / KW;                    // O^T d-tiles per wave / kept after the merge
  static constexpr int XSLOTS = NH * 4;                               // float4 slots per lane and (sender, owner) pair
  static constexpr int XCH_BYTES = KW > 1 ? NW * (KW - 1) * XSLOTS * 1024 : 0;
  static constexpr int ML_BYTES = KW > 1 ? NW * 64 * 8 : 0;           // (m, l) per lane of every wave
  static constexpr int ROWB = NH * 64 + 16;                           // bytes per row of the output image (padded)
  static constexpr int IMG_BYTES = NW * 32 * ROWB;
  static constexpr int MAIN_BYTES = cmax(STAGE_BYTES, XCH_BYTES + ML_BYTES + IMG_BYTES);
  static constexpr int WORDS = 40;                                    // ballot words of 256 keys: Sk <= 10 240 - BN
  static constexpr int LDS_BYTES = MAIN_BYTES + 2 * WORDS * 8;
  static constexpr int MAX_SK = WORDS * 256 - 128;
  static_assert(ND % KW == 0, "every key split keeps a whole number of d-tiles");
  static_assert(!SHARED || NS >= 3, "the shared image is read as K one tile ahead of its use as V");
};

template <int DK, int QW, int KW, int NS, bool SHARED, bool QMASK, bool PF>
__global__ __launch_bounds__(64 * QW * KW, DK == 128 ? 2 : 1) void attn_fwd_kernel(const AttnArgs p) {
  using C = AttnCfg<DK, QW, KW, NS, SHARED>;
  constexpr int NW = C::NW, NT = C::NT, BN = C::BN, VST = C::VST, ND = C::ND, NH = C::NH;
  constexpr int NQ = DK / 16;                  // MFMAs of one S^T chain (= of the O^T update of one tile)
  // LDS layout of the operand stages.  Default: [stage][BN rows].  The scheduled head-dimension-128 loop addresses every
  // stage through the 16-bit immediate of the DS instructions, so a wave's rows of all stages must lie within 64 KiB of its
  // base register: [key split][stage][32 rows] -- a split's NS stages are contiguous (NS * 8 KiB).
  constexpr bool SCHED128 = DK == 128 && PF && SHARED;
  constexpr int STG = SCHED128 ? 32 * DK : VST;          // elements from a stage to the next (as seen by one key split)
  constexpr int KSB = SCHED128 ? NS * 32 * DK : 32 * DK; // elements from a key split's rows to the next split's
  // head dimension 256 needs both halves of the 512-register file (one wave per SIMD): Q^T and O^T in the accumulator
  // half.  Head dimension 128 (256 registers, two waves per SIMD) is built with the VGPR form of the MFMAs and no
  // accumulator-register operand at all (attention128.hip).
  constexpr bool USE_AGPR = DK == 256;
  __shared__ __attribute__((aligned(16))) char smem_raw[C::LDS_BYTES];
  uint64_t* s_slow = reinterpret_cast<uint64_t*>(smem_raw + C::MAIN_BYTES);   // per 256 keys: 4-key groups with a masked / padding key
  uint64_t* s_valid = s_slow + C::WORDS;                                      //               4-key groups with a valid key
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  constexpr float RESCALE_THR = 8.f;   // lazy rescale: keep a stale running max while it lags by < 2^8

  BMHRL_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int qi = wave_s / KW, ki = wave_s % KW;
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;

  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so the q-tiles of one (b, head)
  // -- which stream the same K/V -- are given block ids that differ by multiples of 8 and thus share an L2.
  int bh, qt;
  if (p.map_mode == 0) {
    // one key / value tile for all heads: every (head, q-tile) of a batch row goes to the same XCD
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q1 = fast_div(idx, p.per_b, p.magic_perb), rem = idx - q1 * p.per_b;
    const int hq = fast_div(rem, p.q_tiles, p.magic_qt);
    bh = (xcd + 8 * q1) * p.H + hq;
    qt = rem - hq * p.q_tiles;
  } else if (p.map_mode == 1) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q1 = fast_div(idx, p.q_tiles, p.magic_qt);
    bh = xcd + 8 * q1;
    qt = idx - q1 * p.q_tiles;
  } else {
    bh = fast_div((int)blockIdx.x, p.q_tiles, p.magic_qt);
    qt = (int)blockIdx.x - bh * p.q_tiles;
  }
  const int b = fast_div(bh, p.H, p.magic_h), hd = bh - b * p.H;
  const int q_row = qt * (32 * QW) + qi * 32 + r32;       // this lane's query row
  const bool q_ok = q_row < p.Sq;

  const bf16_t* __restrict__ Kg = p.K + (long)b * p.Sk * p.ldk + hd * p.k_hs;
  const bf16_t* __restrict__ Vg = p.V + (long)b * p.Sk * p.ldv + hd * p.v_hs;

  constexpr bool key_mask = !QMASK;    // same mask for every query row (or none)
  const int nt_all = (p.Sk + BN - 1) / BN;
  const uint8_t* __restrict__ mrow = QMASK ? p.mask + (long)b * p.mask_sb + (long)(q_ok ? q_row : 0) * p.mask_sq : nullptr;
  const uint8_t* __restrict__ mrow_b = (key_mask && p.mask) ? p.mask + (long)b * p.mask_sb : nullptr;
  const bool mask_al4 = (reinterpret_cast<uintptr_t>(mrow_b) & 3) == 0;          // uniform

  // ---- K/V staging: direct-to-LDS loads.  Wave w fills tile rows [RW w, RW w + RW) of an operand, 1 KiB (RPI rows)
  // per instruction: lane l writes chunk (l % CPR) of row RPI i + l / CPR.  Rows are XOR-swizzled in LDS on the SOURCE
  // address -- K: 16-byte chunk ^= row & 15 (the 16 rows of a ds_read_b128 group hit 16 different slots), V: chunk ^=
  // (row & 3) << 2 (the 4 rows of a transposed-read block hit 4 different bank quarters), shared image: both at once,
  // chunk ^= ((row & 3) << 2) | ((row >> 2) & 3).
  constexpr int CPR = DK / 8;                          // 16-byte chunks per row (32 / 16)
  constexpr int RPI = 64 / CPR;                        // rows per instruction (2 / 4)
  constexpr int RW = BN / NW;                          // tile rows a wave stages
  constexpr int GL = RW / RPI;                         // instructions per wave per operand per tile
  static_assert(RW % RPI == 0 && GL >= 1, "a wave stages whole 1 KiB pieces");
  constexpr int GLT = (SHARED ? 1 : 2) * GL;           // pieces per wave and iteration
  const int hi = lane / CPR, pch = lane % CPR;
  const int wrow = wave_s * RW;
  static_assert(!SCHED128 || (32 % RW == 0), "a wave's staged rows stay inside one key split");
  const int wdst = SCHED128 ? (wrow / 32) * KSB + (wrow % 32) * DK : wrow * DK;      // element offset of the wave's first row in a stage
  auto swz_k = [](int row) { return SHARED ? (((row & 3) << 2) | ((row >> 2) & 3)) : (row & 15); };
  auto swz_v = [](int row) { return SHARED ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row & 3) << 2); };
  // Addressing: every load of an operand tile uses ONE uniform base (tile row k0 + RW w, in SGPRs) plus a per-lane
  // 32-bit byte offset that is fixed for the whole launch, and instruction i carries the immediate offset 1024*(i & 3),
  // which the hardware adds to BOTH the global and the LDS address (so it is subtracted from the lane offset here).
  unsigned koffb[GL], voffb[GL];
#pragma unroll
  for (int i = 0; i < GL; ++i) {
    const int r = RPI * i + hi;                        // row inside the wave's row group
    koffb[i] = (unsigned)(r * (int)p.ldk * 2 + ((pch ^ swz_k(wrow + r)) << 4) - 1024 * (i & 3));
    voffb[i] = (unsigned)(r * (int)p.ldv * 2 + ((pch ^ swz_v(wrow + r)) << 4) - 1024 * (i & 3));
  }
  const char* __restrict__ Kgb = reinterpret_cast<const char*>(Kg);
  const char* __restrict__ Vgb = reinterpret_cast<const char*>(Vg);
  // one 1 KiB piece (instruction I of the wave's GL) of a FULL tile whose first row (of this wave) is at `base` (uniform)
  auto issue_piece = [&](auto i_, const char* base, const unsigned (&offb)[GL], bf16_t* sdst) {
    constexpr int I = decltype(i_)::value;
    unsigned o = offb[I];
    asm volatile("" : "+v"(o));      // keep the 32-bit lane offset as it is: (SGPR base + VGPR offset) addressing
    glds16<1024 * (I & 3)>(base + o, sdst + (I / 4) * 4 * RPI * DK);
  };
  auto issue_tile = [&](const char* gb, const long ld, const unsigned (&offb)[GL], const bool is_k, const int t, bf16_t* sdst) {
    const int k0 = t * BN;
    if (k0 + BN <= p.Sk) {
      const char* base = gb + ((long)k0 + wrow) * ld * 2;            // uniform: row k0 + RW w of this batch row
      static_for<0, GL>([&](auto i) { issue_piece(i, base, offb, sdst); });
    } else {   // ragged last tile: clamp the key row (its score gets -inf, so P is exactly 0 there; V must be finite)
      static_for<0, GL>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const int r = RPI * I + hi;
        const int gr = min(k0 + wrow + r, p.Sk - 1);
        const int sw = is_k ? (pch ^ swz_k(wrow + r)) : (pch ^ swz_v(wrow + r));
        glds16<0>(gb + (unsigned)(gr * (int)ld * 2 + (sw << 4)), sdst + RPI * I * DK);
      });
    }
  };
  // stage s of K at smem + s * VST, of V at smem + (NS + s) * VST; the shared image has the K stages only
  auto issue_k = [&](int t) { issue_tile(Kgb, p.ldk, koffb, true, t, smem + (t % NS) * STG + wdst); };
  auto issue_v = [&](int t) {
    if constexpr (!SHARED) issue_tile(Vgb, p.ldv, voffb, false, t, smem + (NS + t % NS) * VST + wrow * DK);
  };
  // Everything the prologue needs is requested up front, in the order of its first use: stage 0 (with Q and the mask
  // bytes: what the first S^T chain needs), then the other prologue stages (the loop's iteration t issues K(t+NS) and
  // V(t+NS-1), or X(t+NS-1) of the shared image) -- ONE exposed memory latency.  (Tiles behind the last valid key are
  // requested too when they fall into the prologue: harmless, they are just never read.)
  // (the first mask word of the lane goes out FIRST: loads return in order, and the ballots below would otherwise sit
  // behind every operand stage of the prologue)
  uint32_t v_pre = 0x01010101u;
  bool pre_ok = false;
  if constexpr (key_mask) {
    if (mrow_b != nullptr && mask_al4 && 4 * tid + 4 <= p.Sk) {
      v_pre = *reinterpret_cast<const uint32_t*>(mrow_b + 4 * tid);
      pre_ok = true;
    }
  }
  issue_k(0);
  issue_v(0);
  BMHRL_STAMP(1)

  // Q^T fragments: lane (q = r32, h) holds Q[q][16*step + 8h .. +8)
  bf16x8 qf[NQ];
  {
    const bf16_t* qp = p.Q + ((long)b * p.Sq + (q_ok ? q_row : 0)) * p.ldq + hd * DK + 8 * h;
#pragma unroll
    for (int s = 0; s < NQ; ++s) qf[s] = q_ok ? *reinterpret_cast<const bf16x8*>(qp + 16 * s) : zero_bf16x8();
  }
#pragma unroll
  for (int s = 1; s < (SHARED ? NS - 1 : NS); ++s)
    if (s < nt_all) issue_k(s);
  if constexpr (!SHARED) {
#pragma unroll
    for (int s = 1; s < NS - 1; ++s)
      if (s < nt_all) issue_v(s);
  }

  // ---- key mask: which 4-key groups hold a masked / padding key ("slow": per-key path) and which a valid key
  // (a tile behind the last valid key of the batch row is never visited).  A thread owns four consecutive keys per
  // pass; a wave's ballot covers 256 keys.
  const int n_words = (nt_all * BN + 255) >> 8;
  if constexpr (key_mask) {
    for (int j = 0; j * NW < n_words; ++j) {
      const int i0 = 4 * (tid + NT * j);
      uint32_t v = 0x01010101u;                                               // no mask: "keep"
      if (j == 0 && pre_ok) {
        v = v_pre;
      } else if (mrow_b != nullptr && i0 < p.Sk) {
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
  }

  // Q^T is loop invariant and only ever an MFMA B operand: pin the registers to the accumulator half of the register file
  // (MFMA reads A/B from there directly), which leaves the arch VGPRs to the K / V^T fragments and the softmax.
  if constexpr (USE_AGPR) {
#pragma unroll
    for (int s = 0; s < NQ; ++s) asm volatile("" : "+a"(qf[s]));
  }
  BMHRL_STAMP(2)

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // log2-domain running max (possibly stale by < RESCALE_THR); l_run is this
                                          // lane's half of the row sum (the two 32-lane halves are added at the end)

  // per-lane LDS byte addresses (stage 0)
  //   V^T fragments: row 32 ki + 4 h + q4 (+16 ks, +8 for the second read of a step), logical chunk 4 d + 2 g1 + (p4 >> 1)
  //   -- separate V image: chunk ^ ((row & 3) << 2): one address per d-tile serves the four reads of a key tile;
  //   -- shared image: chunk ^ ((q4 << 2) | ((h + 2 second) & 3)): the rows 8 apart differ in bit 1 of the chunk
  constexpr unsigned VBASE = SHARED ? 0u : (unsigned)(NS * VST * 2);
  unsigned v_addr[4][SHARED ? 2 : 1];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
#pragma unroll
    for (int sec = 0; sec < (SHARED ? 2 : 1); ++sec) {
      const int lc = 4 * dd + 2 * g1 + (p4 >> 1);
      const int pc = SHARED ? (lc ^ ((q4 << 2) | ((h + 2 * sec) & 3))) : (lc ^ (q4 << 2));
      v_addr[dd][sec] = lds0 + VBASE + 2 * (ki * KSB + (4 * h + q4) * DK) + (pc << 4) + ((p4 & 1) << 3);
    }
  //   K fragments: row 32 ki + r32, logical chunk 2 st + h; steps st and st + 8 are 256 bytes apart (DK = 256), so 8
  //   addresses + an immediate cover the 16 steps
  unsigned k_addr[8];
#pragma unroll
  for (int st = 0; st < 8; ++st)
    k_addr[st] = lds0 + 2 * (ki * KSB + r32 * DK) + (((2 * st + h) ^ swz_k(r32)) << 4);

  // scale + mask the raw scores of one tile through per-key coefficients (masked / padding keys): the mask bytes of the
  // lane's 16 keys come straight from global memory (rare path); returns the tile maximum over the lane pair
  //   valid key  : coef = scale*log2(e), pen = 0        masked key : coef = 0, pen = -1e9*log2(e)
  //   key >= Sk  : coef = 0, pen = -inf  (tile padding)
  const float c_log2 = p.scale * LOG2E;
  auto scale_scores = [&](const f32x16& raw, float (&sc)[16], const int k0) {
    float m_tile = -INFINITY;
    const int key0 = k0 + 32 * ki + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t mk = 0x01010101u;
      if constexpr (key_mask) {
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
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = key0 + 8 * g + j;
        const bool in = key < p.Sk;
        bool keep = in && ((mk >> (8 * j)) & 0xffu) != 0;
        if constexpr (QMASK) keep = in && mrow[in ? key : 0] != 0;   // per-query mask: exact masked_fill semantics
        float v = keep ? raw[4 * g + j] * c_log2 : (in ? NEG_MASK * LOG2E : -INFINITY);
        sc[4 * g + j] = v;
        m_tile = fmaxf(m_tile, v);
      }
    }
    return pair_max(m_tile);
  };
  // lazy rescale (only when some row's max grew by more than RESCALE_THR): everything accumulated so far is at the old
  // max and P of the new tile has not been exponentiated yet, so O and l are scaled exactly once
  // `fix_args`: the exponential arguments in `args` were already formed with the old max; shift them to the new one.
  auto maybe_rescale = [&](const float m_tile, const bool have_o, const bool fix_args, float (&args)[16]) {
    if (__any(m_tile > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      if (fix_args) {
        const float shift = ((m_run == -INFINITY) ? 0.f : m_run) - ((m_new == -INFINITY) ? 0.f : m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) args[r] += shift;
      }
      if (have_o && !USE_AGPR) {
        // one register file (VGPR form of the MFMAs): plain multiplies
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
      if constexpr (USE_AGPR) if (have_o) {
        // O^T lives in the accumulator file (MFMA C/D).  The rescale is rare; each d-tile is handed to the asm as ONE
        // 16-register operand bound to a fixed accumulator range, so the compiler neither splits the tuples nor copies
        // the accumulators to VGPRs around the loop.
#define BMHRL_RESCALE_TILE(D, A0, A1, A2, A3, A4, A5, A6, A7, A8, A9, A10, A11, A12, A13, A14, A15, RANGE)                  \
        {                                                                                                                   \
          float tmp;                                                                                                        \
          asm volatile(BMHRL_RS1(A0) BMHRL_RS1(A1) BMHRL_RS1(A2) BMHRL_RS1(A3) BMHRL_RS1(A4) BMHRL_RS1(A5) BMHRL_RS1(A6)      \
                       BMHRL_RS1(A7) BMHRL_RS1(A8) BMHRL_RS1(A9) BMHRL_RS1(A10) BMHRL_RS1(A11) BMHRL_RS1(A12) BMHRL_RS1(A13) \
                       BMHRL_RS1(A14) BMHRL_RS1(A15)                                                                        \
                       : "+{" RANGE "}"(o[D]), "=&v"(tmp) : "v"(alpha));                                                    \
        }
#define BMHRL_RS1(A) "v_accvgpr_read_b32 %1, " #A "\n\ts_nop 1\n\tv_mul_f32 %1, %2, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 " #A ", %1\n\t"
        BMHRL_RESCALE_TILE(0, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, "a[0:15]")
        BMHRL_RESCALE_TILE(1, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27, a28, a29, a30, a31, "a[16:31]")
        BMHRL_RESCALE_TILE(2, a32, a33, a34, a35, a36, a37, a38, a39, a40, a41, a42, a43, a44, a45, a46, a47, "a[32:47]")
        BMHRL_RESCALE_TILE(3, a48, a49, a50, a51, a52, a53, a54, a55, a56, a57, a58, a59, a60, a61, a62, a63, "a[48:63]")
        {
        BMHRL_RESCALE_TILE(4, a64, a65, a66, a67, a68, a69, a70, a71, a72, a73, a74, a75, a76, a77, a78, a79, "a[64:79]")
        BMHRL_RESCALE_TILE(5, a80, a81, a82, a83, a84, a85, a86, a87, a88, a89, a90, a91, a92, a93, a94, a95, "a[80:95]")
        BMHRL_RESCALE_TILE(6, a96, a97, a98, a99, a100, a101, a102, a103, a104, a105, a106, a107, a108, a109, a110, a111, "a[96:111]")
        BMHRL_RESCALE_TILE(7, a112, a113, a114, a115, a116, a117, a118, a119, a120, a121, a122, a123, a124, a125, a126, a127, "a[112:127]")
        }
#undef BMHRL_RS1
#undef BMHRL_RESCALE_TILE
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
      }
      m_run = m_new;
    }
  };
  auto max_for_exp = [&]() { return (m_run == -INFINITY) ? 0.f : m_run; };
  auto args_slow = [&](float (&sc)[16], const float m_use) {      // sc holds scaled + masked scores
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] -= m_use;
  };

  // ---- S^T chain of one tile: K fragments by ds_read_b128 (asm), first half of the chain starts as soon as the first
  // fragments are there; `mid` runs between the two halves (it issues the V^T reads of the tile in flight), `step(i)`
  // after MFMA i (the exponentials of the previous tile hide under the chain)
  f32x16 s_acc;
  float sc[16];           // scalars, not a 16-register tuple: a slice overwrites single elements in place
  bf16x8 kf[NQ];
  auto qk_issue = [&](const unsigned koffs) {
#pragma unroll
    for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(kf[st]) : "v"(k_addr[st] + koffs));
    if constexpr (DK == 256) {
#pragma unroll
      for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(kf[8 + st]) : "v"(k_addr[st] + koffs));
    }
  };
  // mode 0: first tile (`mid` issues no reads: the second half waits for everything); 1: in the loop, fragments
  // requested at its top; 2: in the loop, fragments requested an iteration ago (PF: one wait for all of them)
  auto qk_chain = [&](auto mode_, auto&& mid, auto&& step) {      // step(integral_constant i) after MFMA i
    constexpr int MODE = decltype(mode_)::value;
    constexpr bool PRE = MODE == 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
    if constexpr (DK == 256) {
      if constexpr (PRE) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      static_for<0, 8>([&](auto st) {
        s_acc = BMHRL_MFMA16(kf[decltype(st)::value], qf[decltype(st)::value], s_acc, 0, 0, 0);
        step(st);
        BMHRL_SB();
      });
      mid();
      // in-order returns: at most 15 younger reads outstanding means the 16 K fragments are all there
      if constexpr (PRE) asm volatile("" : "+v"(kf[8]), "+v"(kf[9]), "+v"(kf[10]), "+v"(kf[11]), "+v"(kf[12]), "+v"(kf[13]), "+v"(kf[14]), "+v"(kf[15]));
      else if constexpr (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[8]), "+v"(kf[9]), "+v"(kf[10]), "+v"(kf[11]), "+v"(kf[12]), "+v"(kf[13]), "+v"(kf[14]), "+v"(kf[15]));
      else asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(kf[8]), "+v"(kf[9]), "+v"(kf[10]), "+v"(kf[11]), "+v"(kf[12]), "+v"(kf[13]), "+v"(kf[14]), "+v"(kf[15]));
      static_for<8, 16>([&](auto st) {
        s_acc = BMHRL_MFMA16(kf[decltype(st)::value], qf[decltype(st)::value], s_acc, 0, 0, 0);
        step(st);
        BMHRL_SB();
      });
    } else {
      if constexpr (PRE) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
      static_for<0, 4>([&](auto st) {
        s_acc = BMHRL_MFMA16(kf[decltype(st)::value], qf[decltype(st)::value], s_acc, 0, 0, 0);
        step(st);
        BMHRL_SB();
      });
      mid();                                   // 16 V^T reads: younger than every K fragment
      if constexpr (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      else if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      static_for<4, 8>([&](auto st) {
        s_acc = BMHRL_MFMA16(kf[decltype(st)::value], qf[decltype(st)::value], s_acc, 0, 0, 0);
        step(st);
        BMHRL_SB();
      });
    }
  };

  // ---- the first tile's operands: every wave waits for its own pieces (and Q / the mask bytes), the barrier publishes
  // them and the ballot words.  Shared image: the later prologue stages (issued after stage 0 and Q) stay in flight.
  constexpr int PRO_LATER = SHARED ? (NS - 2) * GL : 0;       // pieces of stages 1 .. NS-2, when the row has that many tiles
  if (SHARED && nt_all >= NS - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PRO_LATER) : "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_STAMP(3)
  // tiles to visit: up to the last one that holds a valid key (all of them when the batch row has no valid key at all:
  // a fully masked row is uniform over EVERY key, model/multihead_attention.py:22)
  int nt = nt_all;
  if constexpr (key_mask) {
    if (mrow_b != nullptr) {
      int last_group = -1;                                  // last 4-key group with a valid key
      for (int w = 0; w < n_words; ++w) {
        const uint64_t bv = s_valid[w];
        if (bv != 0ull) last_group = 64 * w + 63 - __builtin_clzll(bv);
      }
      if (last_group >= 0) nt = (4 * last_group) / BN + 1;
    }
  }
  nt = __builtin_amdgcn_readfirstlane(nt);
  // bit i of the window: tile (base + i) has a masked or padding key among this wave's 32 (wave-uniform)
  auto slow_window = [&](const int base) -> uint64_t {
    if constexpr (QMASK) return ~0ull;
    const int t = base + lane;
    const int g = t * KW + ki;                               // 32-key group; 8 of them per ballot word
    const uint64_t w = (t < nt_all && (g >> 3) < C::WORDS) ? s_slow[g >> 3] : 0ull;
    return __ballot(((w >> (8 * (g & 7))) & 0xffull) != 0ull);
  };
  uint64_t slow_bits = slow_window(0);
  qk_issue(0u);
  qk_chain(std::integral_constant<int, 0>{}, [] {}, [](auto) {});
  BMHRL_STAMP(4)
  if (slow_bits & 1) {     // masked / padding keys in tile 0: per-key path (mask bytes from global memory)
    const float m_tile = scale_scores(s_acc, sc, 0);
    maybe_rescale(m_tile, false, false, sc);
    args_slow(sc, max_for_exp());
  } else {
    float rmx = s_acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) rmx = fmaxf(rmx, s_acc[r]);
    m_run = pair_max(rmx) * c_log2;              // first tile: the running max is this tile's (finite: every key is valid)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = fmaf(s_acc[r], c_log2, -m_run);
  }
  // what the first iteration reads -- K(1), V(0), and K(2) in its O^T phase -- has landed: every wave waits for its own pieces
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_STAMP(5)

  // ---- main loop.  Iteration t:  phase 1  P(t) = exp2(scores(t) - max)  ||  S^T(t+1) = K(t+1) . Q^T;
  // phase 2  O^T += V^T(t) . P^T(t)  ||  row sum of P(t), scores(t+1), row max, rescale.
  bf16x8 vf[4][2], pf[2];
  auto read_vt = [&](const unsigned soff, auto half) {      // V^T fragments of d-tiles 4*half .. 4*half+3
    constexpr int HOFF = decltype(half)::value * 256;
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      const unsigned a0 = v_addr[dd][0] + soff;
      const unsigned a1 = v_addr[dd][SHARED ? 1 : 0] + soff;
      vf[dd][0] = join8(asm_tr4<HOFF>(a0), asm_tr4<HOFF + 8 * DK * 2>(a1));
      vf[dd][1] = join8(asm_tr4<HOFF + 16 * DK * 2>(a0), asm_tr4<HOFF + 24 * DK * 2>(a1));
    }
  };
  auto wait_vt = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),
                   "+v"(vf[3][0]), "+v"(vf[3][1]));
  };
  // O^T update of d-tiles d0 .. d0+3: 8 MFMAs, `step(i)` after MFMA i
  auto pv = [&](auto d0_, auto&& step) {      // step(integral_constant i) after MFMA i, i = 0 .. 7
    constexpr int D0 = decltype(d0_)::value;
    static_for<0, 8>([&](auto i_) {
      constexpr int I = decltype(i_)::value;
      o[D0 + I / 2] = BMHRL_MFMA16(vf[I / 2][I % 2], pf[I % 2], o[D0 + I / 2], 0, 0, 0);
      step(i_);
      BMHRL_SB();
    });
  };
  auto pack_p = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf[0][j] = (bf16_t)sc[j];
      pf[1][j] = (bf16_t)sc[8 + j];
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  constexpr int EPS = 16 / NQ;            // exponentials per MFMA of the S^T chain (2 / 1)
  constexpr int NPV = NQ;                 // MFMAs of the O^T update
  constexpr int APS = 16 / NPV;           // scores per MFMA of the O^T update (2 / 1)

  if constexpr (PF) qk_issue((unsigned)(1 % NS) * (STG * 2));      // K(1) fragments (a stale stage when nt == 1: never used)
  BMHRL_STAMP(6)
#ifndef BMHRL_ABL
#define BMHRL_ABL 0      // timing ablations of the scheduled loop (tests/kbench/build_abl.sh only; any bit makes the results wrong):
#endif                   // 1 no barrier/vmcnt wait, 2 no global->LDS loads, 4 no exp, 8 no K reads, 16 no V^T reads, 32 no O^T MFMAs,
                         // 64 no S^T MFMAs, 128 no O^T-gap VALU
  if constexpr (SCHED128) {
    // ---- head dimension 128: the iteration as an explicit list of fillers per MFMA gap.  One wave per SIMD issues in
    // order, so the 16 MFMAs of a tile (512 cycles) only run back to back when (a) nothing sits between two of them that
    // waits (the r02 loop read the fresh S^T accumulator right behind its chain: MFMA result latency + 8 converts exposed
    // before the first O^T MFMA), (b) every gap carries <= ~24 issue cycles, (c) LDS reads are spread over the gaps instead
    // of issued in bursts of 8 / 16, and (d) addresses cost no VALU: the stage of every LDS access is a compile-time
    // immediate (the loop is unrolled NS times).  Order of the LDS reads of an iteration (hand-counted lgkmcnt waits):
    //   S^T gap i (i = 0..7):  V^T(t) fragments TR(2i), TR(2i+1) -> vf[i % 4][i / 4]
    //   O^T gap j (j = 0..7):  K(t+2) fragment KR(j) -> kf[j]
    // so S^T MFMA i waits for KR(i) of the previous iteration with at most (7 - i) + 2i reads behind it, and O^T MFMA j for
    // its two TRs with at most (14 - 2j) + j behind them.
    static_assert(!SCHED128 || ((NS - 1) * STG * 2 + 24 * DK * 2 + 16 < 65536), "stage offsets must fit the 16-bit DS immediate");
    static_assert(!SCHED128 || NS >= 4, "K(t+2) is read while tile t+NS-1 is being loaded: they must be different stages");
    auto exp_inplace = [&](float& v) {
      float x = __builtin_amdgcn_exp2f(v);
      asm volatile("" : "+v"(x));
      v = x;
    };
    // what a tile needs of the running maximum, kept in registers between the (rare) rescales: -max for the exponential
    // arguments, and the raw-score threshold beyond which the lazy rescale must run -- so a tile's own bookkeeping is ONE
    // compare (per lane: some lane of a pair exceeding it is exactly the pair's condition) instead of a cross-lane maximum, a
    // multiply, an add, a compare and two selects
    float neg_m = -max_for_exp(), thr_raw = (m_run + RESCALE_THR) / c_log2;
    const int nt_plain = min(nt, p.Sk / BN);                       // tiles loaded as whole tiles (the last one may be ragged)
    auto iter = [&](auto s_, const int t) {
      constexpr int S = decltype(s_)::value;                       // t % NS
      constexpr int SOFF = S * STG * 2, KOFF2 = ((S + 2) % NS) * STG * 2;
      if constexpr (S == NS - 1 && 64 % NS == 0) {
        if ((t & 63) == 63) slow_bits = slow_window(t + 1);        // (rare: Sk > 64 tiles)
      } else if constexpr (64 % NS != 0) {
        if ((t & 63) == 63) slow_bits = slow_window(t + 1);
      }
      const int t_k = t + NS - 1;
      if (!(BMHRL_ABL & 2) && p.dbg != 1) {
        bf16_t* sdst = smem + ((S + NS - 1) % NS) * STG + wdst;
        if (t_k < nt_plain) {                                      // whole tile: one uniform base, immediate piece offsets
          const char* base = Kgb + ((long)t_k * BN + wrow) * p.ldk * 2;
          static_for<0, GL>([&](auto i) { issue_piece(i, base, koffb, sdst); });
        } else if (t_k < nt) {
          issue_tile(Kgb, p.ldk, koffb, true, t_k, sdst);
        }
      }
      float part = 0.f, rmx = -INFINITY;
      const bool slow = (slow_bits >> ((t + 1) & 63)) & 1;         // wave-uniform, rare: masked / padding keys
      // S^T(t+1) = K(t+1) . Q^T   ||  P(t) = exp2(args), V^T(t) fragment reads, first half of the bf16 P^T operand
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
      static_for<0, 8>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        // (one wait per two MFMAs: K fragments I and I + 1 have at most (6 - I) + 2 I younger reads behind them)
        if constexpr (BMHRL_ABL & (8 | 16)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[I]));
        else if constexpr (I % 2 == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(kf[I]), "+v"(kf[I + 1]) : "n"(6 + I));
        if constexpr (!(BMHRL_ABL & 64)) s_acc = BMHRL_MFMA16(kf[I], qf[I], s_acc, 0, 0, 0);
        if constexpr (!(BMHRL_ABL & 4)) {
          exp_inplace(sc[2 * I]);
          exp_inplace(sc[2 * I + 1]);
        }
        if constexpr (!(BMHRL_ABL & 16)) {
          constexpr int DD = I % 4, KS = I / 4;
          vf[DD][KS] = join8(asm_tr4<SOFF + KS * 16 * DK * 2>(v_addr[DD][0]), asm_tr4<SOFF + (KS * 16 + 8) * DK * 2>(v_addr[DD][1]));
        }
        if constexpr (I >= 4) {                                    // elements 0..7 are exponentiated by gap 3
          pf[0][2 * (I - 4)] = (bf16_t)sc[2 * (I - 4)];
          pf[0][2 * (I - 4) + 1] = (bf16_t)sc[2 * (I - 4) + 1];
        }
        BMHRL_SB();
      });
      // O^T += V^T(t) . P^T(t)   ||  second half of P^T, row sum of P(t), K(t+2) fragment reads, and -- from gap 1 on, when the
      // S^T chain has retired -- row max and exponential arguments of tile t+1 (in place: the slot's P was summed already)
      auto arg = [&](const int e) {
        float x = fmaf(s_acc[e], c_log2, neg_m);
        asm volatile("" : "+v"(x));
        sc[e] = x;
      };
      static_for<0, 8>([&](auto j_) {
        constexpr int J = decltype(j_)::value;
        // (V^T fragments of MFMAs J and J + 1: at most (12 - 2 J) + J younger reads behind them)
        if constexpr (BMHRL_ABL & (8 | 16)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[J % 4][J / 4]));
        else if constexpr (J % 2 == 0)
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vf[J % 4][J / 4]), "+v"(vf[(J + 1) % 4][(J + 1) / 4]) : "n"(12 - J));
        if constexpr (!(BMHRL_ABL & 32)) o[J % 4] = BMHRL_MFMA16(vf[J % 4][J / 4], pf[J / 4], o[J % 4], 0, 0, 0);
        if constexpr (!(BMHRL_ABL & 8)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[J]) : "v"(k_addr[J]), "n"(KOFF2));
        if constexpr (BMHRL_ABL & 128) { BMHRL_SB(); return; }
        if constexpr (J < 4) {
          pf[1][2 * J] = (bf16_t)sc[8 + 2 * J];
          pf[1][2 * J + 1] = (bf16_t)sc[9 + 2 * J];
        }
        part += sc[2 * J];
        part += sc[2 * J + 1];
        asm volatile("" : "+v"(part));
        // (v_max3_f32 written out: fmaxf() costs a canonicalising v_max x, x per operand on top)
        if constexpr (J >= 1 && J <= 5) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx) : "v"(s_acc[2 * J - 2]), "v"(s_acc[2 * J - 1]));
          arg(2 * J - 2);
          arg(2 * J - 1);
        } else if constexpr (J == 6) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx) : "v"(s_acc[10]), "v"(s_acc[11]));
          arg(10); arg(11); arg(12);
        } else if constexpr (J == 7) {
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx) : "v"(s_acc[12]), "v"(s_acc[13]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx) : "v"(s_acc[14]), "v"(s_acc[15]));
          arg(13); arg(14); arg(15);
        }
        BMHRL_SB();
      });
      l_run += part;
      if (slow || __any(rmx > thr_raw)) {                          // rare: masked / padding keys in the tile, or the maximum grew by > 2^8
        float m_tile = pair_max(rmx) * c_log2;
        if (slow) {
          m_tile = scale_scores(s_acc, sc, (t + 1) * BN);
          args_slow(sc, -neg_m);
        }
        maybe_rescale(m_tile, true, true, sc);
        neg_m = -max_for_exp();
        thr_raw = (m_run + RESCALE_THR) / c_log2;
      }
      // the loads issued at the top are read as K fragments in the NEXT iteration's O^T phase: everything must have landed
      // (they have had the whole iteration).  Deeper rings (NS = 5, 6: a split's stage is 8 KiB, up to 7 fit the DS immediate)
      // that keep this iteration's pieces in flight across the barrier were measured SLOWER: loop 16.8-17.0 k vs 15.9 k cycles.
      // So was spreading a wave's four pieces over the MFMA gaps (one per four gaps, the barrier moved to the middle of the
      // iteration so that late pieces have half an iteration more to land: 14.5 vs 13.7 us on the V<-A launch, 34.4 vs 33.4
      // at Sq = Sk = 800); a loop body per wave slot (different gaps for different waves) sends the register allocator into
      // scratch (+150 .. 520 bytes of private segment at 256 registers), which costs more per launch than anything it can win
      if constexpr (BMHRL_ABL & 1) return;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    int t = 0;
    while (true) {
      bool done = false;
      static_for<0, NS>([&](auto s_) {
        if (!done) {
          if (t + 1 < nt) { iter(s_, t); ++t; }
          else done = true;
        }
      });
      if (done) break;
    }
  } else
  for (int t = 0; t + 1 < nt; ++t) {
    if ((t & 63) == 63) slow_bits = slow_window(t + 1);      // (rare: Sk > 64 tiles)
    // K(t+1) fragments: their LDS latency passes under the load issue below (PF: requested during the previous
    // iteration's O^T update -- the tile was published a barrier earlier -- so the chain starts right behind the barrier)
    if constexpr (!PF) qk_issue((unsigned)((t + 1) % NS) * (VST * 2));
    // The direct-to-LDS loads of this iteration -- K(t+NS) into the stage of K(t) (read in iteration t-1), V(t+NS-1) into
    // the stage of V(t-1); X(t+NS-1) for the shared image -- go out here, behind the K fragment reads (whose LDS latency
    // passes under the issue).  (Measured and dropped: one piece per MFMA gap, a different gap for every wave -- 8-12 %
    // slower on every shape: the branch + scalar work of a gap costs more than the queueing at the top.)
    const int t_k = SHARED ? t + NS - 1 : t + NS, t_v = t + NS - 1;
    const bool do_k = p.dbg != 1 && t_k < nt;      // (tuning aid: dbg 1 times the loop without its loads)
    const bool do_v = !SHARED && p.dbg != 1 && t_v < nt;
    if (do_k) issue_tile(Kgb, p.ldk, koffb, true, t_k, smem + (t_k % NS) * VST + wrow * DK);
    if (do_v) issue_tile(Vgb, p.ldv, voffb, false, t_v, smem + (NS + t_v % NS) * VST + wrow * DK);
    const unsigned soff = (unsigned)(t % NS) * (VST * 2);
    float part = 0.f;
    qk_chain(std::integral_constant<int, PF ? 2 : 1>{}, [&] { read_vt(soff, H0{}); },       // V^T(t), d-tiles 0..3: wanted at the start of phase 2
             [&](auto i_) {                      // exponentials of tile t under the MFMAs of tile t+1
               constexpr int I = decltype(i_)::value;
#pragma unroll
               for (int e = 0; e < EPS; ++e) {
                 float x = __builtin_amdgcn_exp2f(sc[EPS * I + e]);
                 asm volatile("" : "+v"(x));          // computed HERE, under this MFMA (not sunk to the first use)
                 sc[EPS * I + e] = x;
               }
             });
    pack_p();
    wait_vt();
    // phase 2 VALU work (independent of the MFMAs around it): row sum of P(t) first, then row max of tile t+1 and its
    // exponential arguments with the max as it stands (nearly always the final one)
    const bool slow = (slow_bits >> ((t + 1) & 63)) & 1;    // wave-uniform, rare: masked / padding keys
    float rmx = -INFINITY;
    const float m_use = max_for_exp();
    // slice i of n, one per MFMA: row-sum terms of tile t, row-max terms and exponential arguments of tile t+1 (formed
    // with the max as it stands -- nearly always the final one; the rare rescale shifts them)
    auto p2 = [&](const int i, const int n) {
      const int per = 16 / n;
#pragma unroll
      for (int e = 0; e < per; ++e) {
        part += sc[per * i + e];                                            // P(t) is packed already: the slot is free
        rmx = fmaxf(rmx, s_acc[per * i + e]);
        float x = fmaf(s_acc[per * i + e], c_log2, -m_use);
        asm volatile("" : "+v"(x), "+v"(part));     // computed HERE, under this MFMA (not sunk into the fast branch below)
        sc[per * i + e] = x;
      }
    };
    using D0 = std::integral_constant<int, 0>;
    using D4 = std::integral_constant<int, 4>;
    const unsigned koff2 = (unsigned)((t + 2) % NS) * (VST * 2);
    if constexpr (DK == 256) {
      pv(D0{}, [&](auto i_) { p2(decltype(i_)::value, 16); });
      read_vt(soff, H1{});
      wait_vt();
      if constexpr (PF) qk_issue(koff2);       // K(t+2) fragments land under the MFMAs below
      pv(D4{}, [&](auto i_) { p2(8 + decltype(i_)::value, 16); });
    } else {
      if constexpr (PF) qk_issue(koff2);
      pv(D0{}, [&](auto i_) { p2(decltype(i_)::value, 8); });
    }
    l_run += part;
    float m_tile = pair_max(rmx) * c_log2;
    if (slow) {
      m_tile = scale_scores(s_acc, sc, (t + 1) * BN);
      args_slow(sc, m_use);
    }
    maybe_rescale(m_tile, true, true, sc);

    // ---- the barrier publishes the loads that the next iteration reads and retires the stages it refills; loads that
    // are only needed later stay in flight (NS > 2, or NS > 3 for the shared image)
    constexpr bool KEEP = PF ? (SHARED ? NS >= 5 : NS >= 4) : (SHARED ? NS >= 4 : NS >= 3);
    if (KEEP && do_k && (SHARED || do_v)) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GLT) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  BMHRL_STAMP(7)
  {   // last tile: exponentials and O^T only
    const unsigned soff = (unsigned)((nt - 1) % NS) * (STG * 2);
    read_vt(soff, H0{});
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += sc[r];
    l_run += part;
    pack_p();
    wait_vt();
    pv(std::integral_constant<int, 0>{}, [](auto) {});
    if constexpr (DK == 256) {
      read_vt(soff, H1{});
      wait_vt();
      pv(std::integral_constant<int, 4>{}, [](auto) {});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // every wave is done with the stages: the merge below reuses them
    asm volatile("" ::: "memory");
  }

  BMHRL_STAMP(8)
  // ---- merge + output.  The KW waves of a query block hold partial states (O^T over all DK columns, row max, row sum)
  // for disjoint keys.  Each keeps NH of the d-tiles (wave ki: [ki*NH, ki*NH+NH)) and sends the others to their owners
  // through LDS, so all waves share the combine, the normalisation and the stores; the finished bf16 rows go through a
  // padded LDS image so that every store instruction writes whole 16-byte pieces of contiguous rows (the O^T register
  // layout holds one query row per lane: stored directly, one instruction touches 64 cache lines).
  constexpr int XSLOTS = C::XSLOTS, ROWB = C::ROWB;
  l_run += __shfl_xor(l_run, 32, 64);   // the two 32-lane halves hold disjoint keys of the same query row
  f32x4* xch = reinterpret_cast<f32x4*>(smem_raw);
  float* ml = reinterpret_cast<float*>(smem_raw + C::XCH_BYTES);
  char* img = smem_raw + C::XCH_BYTES + C::ML_BYTES + wave_s * 32 * ROWB;
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  float m_all = m_run, l_all = l_run;
  // exchange slot of (sender ks -> owner ko) of query block qi: index among the KW-1 senders of an owner
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
    ml[(wave_s * 64 + lane) * 2] = m_run;
    ml[(wave_s * 64 + lane) * 2 + 1] = l_run;
    __syncthreads();
  }
  BMHRL_STAMP(9)
  auto finish = [&](auto keep, auto drop) {
    constexpr int MY = decltype(keep)::value * NH;
    constexpr bool DROP = decltype(drop)::value;
    float a_own = 1.f;
    float a_oth[KW > 1 ? KW - 1 : 1];
    if constexpr (KW > 1) {
      float ms[KW], ls[KW];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        ms[k] = ml[((qi * KW + k) * 64 + lane) * 2];
        ls[k] = ml[((qi * KW + k) * 64 + lane) * 2 + 1];
        m = fmaxf(m, ms[k]);
      }
      const float mz = (m == -INFINITY) ? 0.f : m;
      m_all = m;
      l_all = 0.f;
      float a[KW];
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        a[k] = __builtin_amdgcn_exp2f(ms[k] - mz);
        l_all += ls[k] * a[k];
      }
      const float inv = __builtin_amdgcn_rcpf(l_all);      // 1 ulp; the output is rounded to bf16
      int n = 0;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        if (k == decltype(keep)::value) a_own = a[k] * inv;
        else a_oth[n++] = a[k] * inv;
      }
    } else {
      a_own = __builtin_amdgcn_rcpf(l_all);
    }
    const uint64_t ebase = ((uint64_t)b * p.Sq + q_row) * (uint64_t)(p.H * DK) + hd * DK + 4 * h;
#pragma unroll
    for (int dd = 0; dd < NH; ++dd) {
      f32x4 got[KW > 1 ? (KW - 1) * 4 : 1];                 // all reads of a d-tile first: one LDS latency
      if constexpr (KW > 1) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
          if (k == decltype(keep)::value) continue;
          const f32x4* src = xch + slot(k, decltype(keep)::value);
#pragma unroll
          for (int g = 0; g < 4; ++g) got[4 * n + g] = src[(dd * 4 + g) * 64];
          ++n;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = o[MY + dd][4 * g + j] * a_own;
          if constexpr (KW > 1) {
#pragma unroll
            for (int n = 0; n < KW - 1; ++n) x += got[4 * n + g][j] * a_oth[n];
          }
          if constexpr (DROP) x *= dropout_scale(p.dropout_p, seed, ebase + 32 * (MY + dd) + 8 * g + j);
          w[j] = (bf16_t)x;
        }
        *reinterpret_cast<bf16x4*>(img + r32 * ROWB + (dd * 32 + 8 * g + 4 * h) * 2) = w;
      }
    }
    // rows of the image -> global: CPRO 16-byte pieces per row, 64 / CPRO rows per instruction (a wave reads back what
    // it wrote itself: no barrier, the compiler's own lgkmcnt wait orders the two)
    constexpr int CPRO = NH * 4, RPS = (64 / CPRO) < 32 ? (64 / CPRO) : 32, NST = 32 / RPS;
    const int srow = lane / CPRO, sch = lane % CPRO;
    const int q0 = qt * (32 * QW) + qi * 32;
    if (srow < RPS) {
      bf16x8 wout[NST];
#pragma unroll
      for (int i = 0; i < NST; ++i) wout[i] = *reinterpret_cast<const bf16x8*>(img + (RPS * i + srow) * ROWB + sch * 16);
      bf16_t* op = p.O + ((long)b * p.Sq + q0 + srow) * p.ldo + hd * DK + MY * 32 + sch * 8;
      const long ostep = (long)RPS * p.ldo;
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        if (q0 + RPS * i + srow < p.Sq) *reinterpret_cast<bf16x8*>(op) = wout[i];
        op += ostep;
      }
    }
  };
  using DropOff = std::integral_constant<bool, false>;
  using DropOn = std::integral_constant<bool, true>;
  static_for<0, KW>([&](auto k_) {
    if (ki == decltype(k_)::value) {
      if (p.dropout_p > 0.f) finish(k_, DropOn{});
      else finish(k_, DropOff{});
    }
  });
  if (ki == 0 && h == 0 && q_ok) {
    const long si = ((long)b * p.H + hd) * p.Sq + q_row;
    // statistics in natural-log units: P = exp(score - row_max) / row_sum.  A fully masked row keeps the exact
    // fill value so that the backward recomputation exp(-1e9 - row_max) is exp(0).
    p.row_max[si] = (m_all <= NEG_MASK * LOG2E) ? NEG_MASK : m_all * LN2;
    p.row_sum[si] = l_all;
  }
  BMHRL_STAMP(10)
}

const int g_attn_dbg = getenv("BMHRL_ATTN_DBG") ? atoi(getenv("BMHRL_ATTN_DBG")) : 0;   // read once, at load time
const bool g_attn_pair_on = getenv("BMHRL_ATTN_PAIR") && atoi(getenv("BMHRL_ATTN_PAIR")) != 0;    // the pair form in the automatic choice

inline void attn_trace_dump(const char* what, int Sq, int Sk, hipStream_t stream, int last = 10, int total_at = 10) {
#ifdef BMHRL_ATTN_TRACE
  if (getenv("BMHRL_ATTN_TRACE")) {
    long long hh[2][16];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_attn_trace), sizeof(hh));
    for (int w = 0; w < 2; ++w) {
      fprintf(stderr, "%s trace (%s block, Sq %d Sk %d):", what, w ? "last" : "first", Sq, Sk);
      for (int i = 1; i <= last; ++i) fprintf(stderr, " %lld", hh[w][i] - hh[w][i - 1]);
      fprintf(stderr, "  total %lld\n", hh[w][total_at] - hh[w][0]);
    }
  }
#else
  (void)what; (void)Sq; (void)Sk; (void)stream; (void)last; (void)total_at;
#endif
}

template <int DK, int QW, int KW, int NS, bool SHARED, bool PF = false>
hipError_t launch_attn(AttnArgs a, hipStream_t stream) {
  using C = AttnCfg<DK, QW, KW, NS, SHARED>;
  if (a.Sk > C::MAX_SK) return hipErrorInvalidValue;
  a.q_tiles = (a.Sq + 32 * QW - 1) / (32 * QW);
  if ((int64_t)a.B * a.H * a.q_tiles * a.H * a.q_tiles >= (1ll << 32)) return hipErrorInvalidValue;
  set_block_map(a);
  dim3 grid((unsigned)(a.B * a.H * a.q_tiles)), block(C::NT);
  if constexpr (!SHARED) {
    if (a.mask != nullptr && a.mask_sq != 0) {
      hipLaunchKernelGGL((attn_fwd_kernel<DK, QW, KW, NS, SHARED, true, PF>), grid, block, 0, stream, a);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((attn_fwd_kernel<DK, QW, KW, NS, SHARED, false, PF>), grid, block, 0, stream, a);
  return hipGetLastError();
}

// ---- entry bodies shared by the bf16 and the fp16 translation units (the extern "C" symbols differ, the kernels do not).
// Templates, so that a translation unit only instantiates the kernels of the entry it defines.
template <int UNUSED = 0>
int attention256_entry(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O,
                              int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                              int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk, int32_t dk, float scale,
                              float dropout_p, uint64_t seed, const uint64_t* seed_dev, int code, hipStream_t stream) {
  constexpr int DK = 256;
  BMHRL_CHECK_ARG(Q && K && V && O && row_max && row_sum);
  BMHRL_CHECK_ARG(dk == DK);  // d_model 1024 / H 4 of the reference; other head sizes use the materialised path
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * DK && ldk >= (int64_t)H * DK && ldv >= (int64_t)H * DK && ldo >= (int64_t)H * DK);
  BMHRL_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0);
  BMHRL_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  BMHRL_CHECK_ARG((int64_t)Sk * ldk * 2 < (1ll << 31) && (int64_t)Sk * ldv * 2 < (1ll << 31));   // 32-bit lane offsets
  AttnArgs a;
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.O = (bf16_t*)O; a.ldo = ldo; a.row_max = row_max; a.row_sum = row_sum;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev;
  a.k_hs = DK; a.v_hs = DK;
  a.dbg = g_attn_dbg;
  // 4 x 1 (128 query rows per workgroup, every wave sees every key: no merge, K fragments requested across the barrier)
  // as soon as that gives most CUs a workgroup; else 2 x 2 (64 rows, two key halves).  Measured on MI355X at B16 H4
  // (tests/kbench/attn_bench time): Sq800 Sk256 35.0 vs 47.5 us, Sq256 Sk256 17.4 vs 14.4 us, Sq256 Sk800 36 vs 28 us.
  if (code == 0) code = ((int64_t)B * H * ((Sq + 127) / 128) >= 200) ? 41 : 22;
  hipError_t e;
  if (code == 41) e = launch_attn<DK, 4, 1, 3, false, true>(a, stream);
  else if (code == 22) e = launch_attn<DK, 2, 2, 2, false, false>(a, stream);
  else return -22;
  attn_trace_dump("attn256", Sq, Sk, stream);
  return hip_status(e);
}

// attention128p.hip (the pair form, bf16 operands only): code 14; nullptr in translation units that do not link it
typedef int (*attn128_pair_fn)(const void*, int64_t, const void*, int64_t, void*, int64_t, float*, float*, const uint8_t*, int64_t,
                               int32_t, int32_t, int32_t, int32_t, float, hipStream_t);
typedef bool (*attn128_pair_ok_fn)(int, int, int, int);

template <int UNUSED = 0>
int attention128_entry(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo, float* row_max,
                              float* row_sum, const uint8_t* mask, int64_t mask_sb, int32_t B, int32_t H, int32_t Sq,
                              int32_t Sk, float scale, int code, hipStream_t stream, attn128_pair_fn pair_fn = nullptr,
                              attn128_pair_ok_fn pair_ok = nullptr) {
  constexpr int DK = 128;
  BMHRL_CHECK_ARG(Qp && X && ctx && row_max && row_sum);
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * DK && ldx >= DK && ldo >= (int64_t)H * DK);
  BMHRL_CHECK_ARG((((uintptr_t)Qp | (uintptr_t)X | (uintptr_t)ctx) & 15) == 0);
  BMHRL_CHECK_ARG((int64_t)Sk * ldx * 2 < (1ll << 31));
  AttnArgs a;
  a.Q = (const bf16_t*)Qp; a.ldq = ldq; a.K = (const bf16_t*)X; a.ldk = ldx; a.V = (const bf16_t*)X; a.ldv = ldx;
  a.O = (bf16_t*)ctx; a.ldo = ldo; a.row_max = row_max; a.row_sum = row_sum;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = 0;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = 0.f; a.seed = 0; a.seed_dev = nullptr;
  a.k_hs = 0; a.v_hs = 0;                       // one 128-wide key / value row for every head
  a.dbg = g_attn_dbg;
  // 4 x 1 (128 query rows per workgroup, two workgroups per CU) when there are enough 32-row query blocks to give every
  // SIMD two waves that way; else 2 x 2 (64 rows, two key halves).  Both request the K fragments of the next tile across
  // the barrier (four stages of the shared image).  Measured on MI355X at B16 H4 (tests/kbench/attn_bench time):
  // Sq800 Sk800 30.4 vs 34.9 us, Sq256 Sk800 20.2 vs 14.3 us; eight waves (2 x 4) lose to 2 x 2 on every shape tried.
  // The pair form (attention_pair.h: 64 slots x 4 key splits per workgroup, no barrier in the loop) when its workgroups fill
  // the chip in one round and a key split has at least two tiles -- the shape of BASELINE configs[1]'s V<-A attention.
  if (code == 0) {
    const int64_t rows32 = (int64_t)B * H * ((Sq + 31) / 32);      // 32-row query blocks
    const bool pair = pair_fn != nullptr && pair_ok(B, H, Sq, Sk) && rows32 <= 2 * 304 && Sk >= 256 && g_attn_pair_on;
    code = pair ? 14 : (rows32 >= 1536 ? 41 : 22);
  }
  if (code == 14) {
    if (pair_fn == nullptr || !pair_ok(B, H, Sq, Sk)) return -22;
    return pair_fn(Qp, ldq, X, ldx, ctx, ldo, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, stream);
  }
  hipError_t e;
  if (code == 41) e = launch_attn<DK, 4, 1, 4, true, true>(a, stream);
  else if (code == 22) e = launch_attn<DK, 2, 2, 4, true, true>(a, stream);
  else if (code == 24) e = launch_attn<DK, 2, 4, 4, true, true>(a, stream);      // eight waves: two per SIMD at 64 rows per CU
  else return -22;
  attn_trace_dump("attn128", Sq, Sk, stream);
  return hip_status(e);
}

}  // namespace

