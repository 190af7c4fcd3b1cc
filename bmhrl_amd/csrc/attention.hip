// Fused scaled-dot-product attention forward for gfx950 (flash style, d_k = 256, bf16 MFMA, fp32 softmax).
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
// Work split: block = 2 x 2 waves, 64 query rows (B*H*Sq/64 >= 256 blocks at the reference shapes, one wave per SIMD).
// Wave (qi, ki) owns 32 query rows and the keys [32 ki, 32 ki + 32) of every 64-key tile; the two key halves keep
// private online-softmax state and are merged once at the end through LDS.
//
// Data movement and schedule:
//   * K and V tiles go global -> LDS with direct-to-LDS loads (no staging registers, no ds_write), two stages each,
//     issued at the top of an iteration for the next one; rows are XOR-swizzled (on the source address) so that the
//     ds_read_b128 of K and the transposed reads of V are bank-conflict free without padding;
//   * the loop is software pipelined inside the single wave a SIMD holds: the S^T MFMA chain of tile t+1 carries one
//     exponential of tile t under every MFMA, the O^T MFMAs of tile t carry the score scaling / row max of tile t+1;
//   * tiles whose keys are all valid and unmasked (nearly all) scale scores by one constant; the per-key coefficient
//     path (mask, padding, per-query masks) is a rare wave-uniform fix-up;
//   * LDS reads inside the loop are inline asm with hand-counted lgkmcnt waits: the compiler orders every ds_read it
//     knows about behind ALL outstanding direct-to-LDS loads (it cannot tell the stages apart, so each became a
//     vmcnt(0) that serialised the prefetch);
//   * Q fragments are pinned to the accumulator half of the register file (MFMA reads B operands from there), the
//     O^T accumulators are handed to the rare rescale as whole 16-register tuples, and the file is built with
//     -amdgpu-codegenprepare-break-large-phis=false: otherwise the loop-carried accumulators are split into 128 scalar
//     VGPR PHIs, i.e. 256 accumulator<->VGPR copies per tile.
// Bounds at the reference shape (64 query rows per CU): per 64-key tile a CU needs 1024 MFMA cycles per SIMD, 128 KiB
// of LDS reads (1024 cycles at 128 B/clk) and 64 KiB through the vector memory path (1024 cycles at 64 B/clk) -- the
// three pipes are balanced, so the kernel is as much LDS/L1-bandwidth bound as MFMA bound.  (Loading K fragments
// straight from global memory was tried: 32 rows x 32 bytes per load instruction makes it TCP-line-rate bound, slower.)
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

// The head dimension DK is a template parameter: 256 = d_model / H of the reference; 128 = the absorbed-projection form
// of the attentions whose keys / values are the 128-wide audio stream (scores_h = (Q_h Wk_h) A^T, context_h = P_h A:
// one key/value tile shared by all heads, half the FLOPs, and half the registers -> two workgroups per CU).
constexpr int QW = 2, KW = 2, NT = 64 * QW * KW;
constexpr int BN = 32 * KW;                 // keys per tile
// per-key softmax coefficients of one batch row: score2 = fma(q.k, coef[key], pen[key]) in the log2 domain
//   valid key  : coef = scale*log2(e), pen = 0        masked key : coef = 0, pen = -1e9*log2(e)
//   key >= Sk  : coef = 0, pen = -inf  (tile padding)
template <int DK> constexpr int max_keys() { return DK == 128 ? 1024 + 64 : 2048 + 64; }   // (128: 2 workgroups per CU must fit)

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
template <int OFF>
__device__ __forceinline__ f32x4 asm_ldsf4(unsigned addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
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

// Tuning aid (-DBMHRL_ATTN_TRACE): cycle stamps of wave 0 of the first and the last workgroup at the phase boundaries.
#ifdef BMHRL_ATTN_TRACE
__device__ long long g_attn_trace[2][16];
#define BMHRL_STAMP(i)                                                                                   \
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                              \
    g_attn_trace[blockIdx.x != 0][i] = (long long)__builtin_readcyclecounter();
#else
#define BMHRL_STAMP(i)
#endif

template <int DK, bool QMASK>
__global__ __launch_bounds__(NT, DK == 128 ? 2 : 1) void attn_fwd_kernel(const AttnArgs p) {
  constexpr int VST = BN * DK;                // elements of one K or V stage (32 / 16 KiB)
  constexpr int LDS_KV = 4 * VST * 2;         // bytes: K stage 0, K stage 1, V stage 0, V stage 1
  constexpr int MAXK = max_keys<DK>();
  constexpr int OREGS = DK / 2;               // O^T accumulator registers per lane
  constexpr int MERGE_FLOATS = (KW - 1) * QW * (OREGS + 2) * 64;
  static_assert(MERGE_FLOATS * 4 <= LDS_KV, "the merge area reuses the K/V stages");
  constexpr int MAXT = MAXK / BN;              // key tiles
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_KV + 2 * MAXK * 4 + (MAXT / 4 + 4) * 8];
  float* s_coef = reinterpret_cast<float*>(smem_raw + LDS_KV);
  float* s_pen = s_coef + MAXK;
  uint64_t* s_slow = reinterpret_cast<uint64_t*>(s_pen + MAXK);   // per 4 tiles: a byte per (tile, key half) with pen != 0
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  constexpr float RESCALE_THR = 8.f;   // lazy rescale: keep a stale running max while it lags by < 2^8

  BMHRL_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = wave / KW, ki = wave % KW;
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

  constexpr bool key_mask = !QMASK;    // same mask for every query row (or none) -> LDS coefficients
  const int nt = (p.Sk + BN - 1) / BN;
  const uint8_t* __restrict__ mrow = QMASK ? p.mask + (long)b * p.mask_sb + (long)(q_ok ? q_row : 0) * p.mask_sq : nullptr;

  // ---- K/V staging: direct-to-LDS loads (no staging registers, no ds_write).  Wave w fills tile rows [16w, 16w+16) of
  // an operand, two rows (1 KiB) per instruction: lane l writes chunk (l & 31) of row 16w + 2i + (l >> 5).  Rows are
  // XOR-swizzled in LDS -- K: 16-byte chunk ^= row & 15 (the 16 rows of a ds_read_b128 group hit 16 different slots),
  // V: chunk ^= (row & 3) << 2 (the 4 rows of a transposed-read block hit 4 different bank quarters) -- and the
  // swizzle is applied on the SOURCE address.
  constexpr int CPR = DK / 8;                          // 16-byte chunks per row (32 / 16)
  constexpr int RPI = 64 / CPR;                        // rows per instruction (2 / 4): one wave instruction moves 1 KiB
  constexpr int GL = BN / (QW * KW) / RPI;             // instructions per wave per operand per tile (8 / 4)
  const int hi = lane / CPR, pch = lane % CPR;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wrow = wave_s * (BN / (QW * KW));
  // Addressing: every load of an operand tile uses ONE uniform base (tile row k0 + 16w, in SGPRs) plus a per-lane
  // 32-bit byte offset that is fixed for the whole launch, and instruction i carries the immediate offset 1024*(i & 3),
  // which the hardware adds to BOTH the global and the LDS address (so it is subtracted from the lane offset here):
  // no per-load address arithmetic is left in the loop.
  unsigned koffb[GL], voffb[GL];
#pragma unroll
  for (int i = 0; i < GL; ++i) {
    const int r = RPI * i + hi;                        // row inside the wave's 16-row group
    koffb[i] = (unsigned)(r * (int)p.ldk * 2 + ((pch ^ (r & 15)) << 4) - 1024 * (i & 3));
    voffb[i] = (unsigned)(r * (int)p.ldv * 2 + ((pch ^ ((r & 3) << 2)) << 4) - 1024 * (i & 3));
  }
  const char* __restrict__ Kgb = reinterpret_cast<const char*>(Kg);
  const char* __restrict__ Vgb = reinterpret_cast<const char*>(Vg);
  // tiles are requested in order (K: 0, 1, 2, ...; V likewise), so each operand keeps a running uniform row pointer
  const char* k_next = Kgb + (long)wrow * p.ldk * 2;
  const char* v_next = Vgb + (long)wrow * p.ldv * 2;
  auto issue_tile = [&](const char* gb, const char*& next, const long ld, const unsigned (&offb)[GL], const int swz_k,
                        const int t, bf16_t* sdst) {
    const int k0 = t * BN;
    const char* base = next;                                       // uniform: row k0 + 16 w of this batch row
    next += (long)BN * ld * 2;
    if (k0 + BN <= p.Sk) {
      static_for<0, GL>([&](auto i) {
        constexpr int I = decltype(i)::value;
        unsigned o = offb[I];
        asm volatile("" : "+v"(o));      // keep the 32-bit lane offset as it is: (SGPR base + VGPR offset) addressing
        glds16<1024 * (I & 3)>(base + o, sdst + (I / 4) * 4 * RPI * DK);
      });
    } else {   // ragged last tile: clamp the key row (its score gets pen = -inf, so P is exactly 0 there; V must be finite)
      static_for<0, GL>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const int r = RPI * I + hi;
        const int gr = min(k0 + wrow + r, p.Sk - 1);
        const int sw = swz_k ? (pch ^ (r & 15)) : (pch ^ ((r & 3) << 2));
        glds16<0>(gb + (unsigned)(gr * (int)ld * 2 + (sw << 4)), sdst + RPI * I * DK);
      });
    }
  };
  auto issue_k = [&](int t, int buf) { issue_tile(Kgb, k_next, p.ldk, koffb, 1, t, smem + buf * VST + wrow * DK); };
  auto issue_v = [&](int t, int buf) { issue_tile(Vgb, v_next, p.ldv, voffb, 0, t, smem + (2 + buf) * VST + wrow * DK); };
  issue_k(0, 0);          // with Q and the mask bytes: what the first S^T chain needs; V(0) and K(1) follow below
  BMHRL_STAMP(1)
  // key-mask bytes: a thread owns FOUR consecutive keys per pass (one 32-bit load when the row is 4-byte aligned), all
  // passes requested before the first use (one exposed latency, shared with Q and the first K/V stage)
  constexpr int NCO = (MAXK + 4 * NT - 1) / (4 * NT);
  uint32_t mk[NCO];
  if constexpr (key_mask) {
    const uint8_t* mrow_b = p.mask ? p.mask + (long)b * p.mask_sb : nullptr;
    const bool al4 = (reinterpret_cast<uintptr_t>(mrow_b) & 3) == 0;          // uniform
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
      const int i0 = 4 * (tid + NT * j);
      uint32_t v = 0x01010101u;                                               // no mask / keys past Sk: "keep" (pen handles Sk)
      if (mrow_b != nullptr && i0 < p.Sk) {
        if (al4 && i0 + 4 <= p.Sk) {
          v = *reinterpret_cast<const uint32_t*>(mrow_b + i0);
        } else {
          v = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i0 + e < p.Sk) v |= (uint32_t)mrow_b[i0 + e] << (8 * e);
        }
      }
      mk[j] = v;
    }
  }

  // Q^T fragments: lane (q = r32, h) holds Q[q][16*step + 8h .. +8)
  bf16x8 qf[DK / 16];
  {
    const bf16_t* qp = p.Q + ((long)b * p.Sq + (q_ok ? q_row : 0)) * p.ldq + hd * DK + 8 * h;
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) qf[s] = q_ok ? *reinterpret_cast<const bf16x8*>(qp + 16 * s) : zero_bf16x8();
    // Loop invariant and only ever an MFMA B operand: pin the 64 registers to the accumulator half of the register file
    // (MFMA reads A/B from there directly), which leaves the arch VGPRs to the K / V^T fragments and the softmax.
#pragma unroll
    for (int s = 0; s < DK / 16; ++s) asm volatile("" : "+a"(qf[s]));
  }

  BMHRL_STAMP(2)
  // per-key coefficients, four keys per lane and pass.  A wave covers 256 keys = 4 tiles per pass (16 lanes per tile, 8
  // per key half), so ONE ballot of "some key of mine needs the per-key path" holds a byte per (tile, key half).
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    const int i0 = 4 * (tid + NT * j);
    f32x4 cf, pn;
    bool any_slow = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = i0 + e < p.Sk;
      bool keep = in;
      if constexpr (key_mask) keep = in && ((mk[j] >> (8 * e)) & 0xffu) != 0;
      cf[e] = (in && (keep || !key_mask)) ? p.scale * LOG2E : 0.f;
      pn[e] = in ? ((keep || !key_mask) ? 0.f : NEG_MASK * LOG2E) : -INFINITY;
      any_slow |= pn[e] != 0.f;
    }
    if (i0 < nt * BN) {                                  // nt * BN is a multiple of 4
      *reinterpret_cast<f32x4*>(s_coef + i0) = cf;
      *reinterpret_cast<f32x4*>(s_pen + i0) = pn;
    }
    const uint64_t bal = __ballot(any_slow);
    if (lane == 0) s_slow[(NT / 64) * j + wave_s] = bal;  // word w: tiles 4w .. 4w+3
  }

  f32x16 o[DK / 32];
#pragma unroll
  for (int d = 0; d < DK / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // log2-domain running max (possibly stale by < RESCALE_THR); l_run is this
                                          // lane's half of the row sum (the two 32-lane halves are added at the end)

  // per-lane LDS byte addresses
  //   V^T fragments: row 32 ki + 4 h + q4 (+16 ks, +8 for the second half), chunk 4 (d ^ (row & 3)) + 2 g1 + (p4 >> 1)
  unsigned v_addr[4];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
    v_addr[dd] = lds0 + 2 * VST * 2 + 2 * ((32 * ki + 4 * h + q4) * DK + ((dd ^ q4) << 5) + ((2 * g1 + (p4 >> 1)) << 3) + ((p4 & 1) << 2));
  //   K fragments: row 32 ki + r32, logical chunk 2 st + h at physical chunk (2 st + h) ^ (row & 15); steps st and st + 8
  //   are 256 bytes apart, so 8 addresses + an immediate cover the 16 steps
  unsigned k_addr[8];
#pragma unroll
  for (int st = 0; st < 8; ++st)
    k_addr[st] = lds0 + 2 * ((32 * ki + r32) * DK + (((2 * st + h) ^ (r32 & 15)) << 3));
  unsigned c_addr = lds0 + LDS_KV + 4 * (32 * ki + 4 * h);      // coefficients of this lane's keys in tile 0

  // scale + mask the raw scores of one tile (accumulator -> log2-domain scores), and their maximum over the lane pair
  auto scale_scores = [&](const f32x16& raw, f32x16& sc, const int k0, const f32x4 (&cf)[4], const f32x4 (&pn)[4]) {
    float m_tile = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = fmaf(raw[4 * g + j], cf[g][j], pn[g][j]);
        if constexpr (QMASK) {   // per-query mask (not the encoder's case): exact masked_fill semantics
          const int key = k0 + 32 * ki + 4 * h + 8 * g + j;
          if (key < p.Sk && !mrow[key]) v = NEG_MASK * LOG2E;
        }
        sc[4 * g + j] = v;
        m_tile = fmaxf(m_tile, v);
      }
    return pair_max(m_tile);
  };
  // lazy rescale (only when some row's max grew by more than RESCALE_THR): everything accumulated so far is at the old
  // max and P of the new tile has not been exponentiated yet, so O and l are scaled exactly once
  // `fix_args`: the exponential arguments in `sc` were already formed with the old max (the common case needs no second
  // pass over the scores); shift them to the new one.
  auto maybe_rescale = [&](const float m_tile, const bool have_o, const bool fix_args, f32x16& args) {
    if (__any(m_tile > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, m_tile);
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      if (fix_args) {
        const float shift = ((m_run == -INFINITY) ? 0.f : m_run) - ((m_new == -INFINITY) ? 0.f : m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) args[r] += shift;
      }
      if (have_o) {
        // O^T lives in the accumulator file (MFMA C/D).  The rescale is rare; each d-tile is handed to the asm as ONE
        // 16-register operand bound to a fixed accumulator range, so the compiler neither splits the tuples nor copies
        // the 128 accumulators to VGPRs around the loop.
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
        if constexpr (DK == 256) {
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
  auto read_coef = [&](const unsigned a, f32x4 (&cf)[4], f32x4 (&pn)[4]) {
    cf[0] = asm_ldsf4<0>(a);  cf[1] = asm_ldsf4<32>(a);  cf[2] = asm_ldsf4<64>(a);  cf[3] = asm_ldsf4<96>(a);
    pn[0] = asm_ldsf4<MAXK * 4>(a);      pn[1] = asm_ldsf4<MAXK * 4 + 32>(a);
    pn[2] = asm_ldsf4<MAXK * 4 + 64>(a); pn[3] = asm_ldsf4<MAXK * 4 + 96>(a);
  };
  // tiles whose 32 keys (of this wave) are all valid and unmasked -- nearly all of them -- only need score * c
  const float c_log2 = p.scale * LOG2E;
  // Fast tiles: max over the raw scores (the scale is positive), then  x = raw * c - max  as ONE packed fma per two scores;
  // `sc` always holds the exponential ARGUMENTS of the tile in flight (log2 domain, max already subtracted).
  auto raw_max = [&](const f32x16& raw) {
    float m = raw[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, raw[r]);
    return pair_max(m) * c_log2;
  };
  auto args_fast = [&](const f32x16& raw, f32x16& sc, const float m_use) {
    const f32x2 c2 = {c_log2, c_log2}, nm2 = {-m_use, -m_use};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 v = f32x2{raw[r], raw[r + 1]} * c2 + nm2;
      sc[r] = v[0];
      sc[r + 1] = v[1];
    }
  };
  auto args_slow = [&](f32x16& sc, const float m_use) {      // sc holds scaled + masked scores
    const f32x2 nm2 = {-m_use, -m_use};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 v = f32x2{sc[r], sc[r + 1]} + nm2;
      sc[r] = v[0];
      sc[r + 1] = v[1];
    }
  };
  auto max_for_exp = [&]() { return (m_run == -INFINITY) ? 0.f : m_run; };

  // ---- S^T chain of one tile: K fragments by ds_read_b128 (asm), first half of the chain starts as soon as the first 8
  // fragments are there; `mid` runs between the two halves (it issues the V^T reads of the tile in flight), `step(st)`
  // after every MFMA (the exponentials of the previous tile hide under the chain)
  f32x16 s_acc, sc;
  bf16x8 kf[DK / 16];
  auto qk_issue = [&](const unsigned koffs) {
#pragma unroll
    for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(kf[st]) : "v"(k_addr[st] + koffs));
    if constexpr (DK == 256) {
#pragma unroll
      for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(kf[8 + st]) : "v"(k_addr[st] + koffs));
    }
  };
  // `step(i)`, i = 0..7, is called at even spacing over the chain (two exponentials of the previous tile per call)
  auto qk_chain = [&](auto&& mid, auto&& step) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
    if constexpr (DK == 256) {
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s_acc, 0, 0, 0);
        if (st & 1) step(st >> 1);
      }
      mid();
      // in-order returns: at most 15 younger reads outstanding means the 16 K fragments are all there
      asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(kf[8]), "+v"(kf[9]), "+v"(kf[10]), "+v"(kf[11]), "+v"(kf[12]), "+v"(kf[13]), "+v"(kf[14]), "+v"(kf[15]));
#pragma unroll
      for (int st = 8; st < 16; ++st) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s_acc, 0, 0, 0);
        if (st & 1) step(st >> 1);
      }
    } else {
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s_acc, 0, 0, 0);
        step(st);
      }
      mid();                                   // 16 V^T reads: younger than every K fragment
      asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
#pragma unroll
      for (int st = 4; st < 8; ++st) {
        s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s_acc, 0, 0, 0);
        step(st);
      }
    }
  };

  BMHRL_STAMP(3)
  // K(0) has landed (the wait in front of the first use of Q / the mask bytes drained every load); V(0) and K(1) now go
  // out and land under the first S^T chain, so the barrier here must not wait for them: LDS counter only
  issue_v(0, 0);
  if (nt > 1) issue_k(1, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // coefficient / ballot writes
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_STAMP(4)
  qk_issue(0u);
  qk_chain([] {}, [](int) {});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // V(0), K(1) are there, and every wave is done with K stage 0 (the loop refills it)
  asm volatile("" ::: "memory");
  BMHRL_STAMP(5)
  uint64_t slow_bits = 0;            // bit t: tile t has a masked or padding key among this wave's 32 (wave-uniform)
  if constexpr (QMASK) {
    slow_bits = ~0ull;
  } else {
    static_assert(BN == 64 && MAXT <= 64 && NT == 256, "a ballot word per 4 tiles (a byte per tile and key half), a lane per tile");
    const uint64_t w = lane < nt ? s_slow[lane >> 2] : 0ull;
    slow_bits = __ballot(((w >> (16 * (lane & 3) + 8 * ki)) & 0xffull) != 0ull);
  }
  {
    f32x4 cf[4], pn[4];
    read_coef(c_addr, cf, pn);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(pn[3]));
    const float m_tile = scale_scores(s_acc, sc, 0, cf, pn);
    maybe_rescale(m_tile, false, false, sc);
    args_slow(sc, max_for_exp());
  }

  // ---- main loop.  Iteration t:  phase 1  P(t) = exp2(scores(t) - max)  ||  S^T(t+1) = K(t+1) . Q^T (K fragments are
  // refilled with tile t+2 as they are consumed);  phase 2  O^T += V^T(t) . P^T(t)  ||  scores(t+1), row max, rescale.
  bf16x8 vf[4][2], pf[2];
  auto read_vt = [&](const unsigned soff, auto half) {      // V^T fragments of d-tiles 4*half .. 4*half+3
    constexpr int HOFF = decltype(half)::value * 256;
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      const unsigned a = v_addr[dd] + soff;
      vf[dd][0] = join8(asm_tr4<HOFF>(a), asm_tr4<HOFF + 8 * DK * 2>(a));
      vf[dd][1] = join8(asm_tr4<HOFF + 16 * DK * 2>(a), asm_tr4<HOFF + 24 * DK * 2>(a));
    }
  };
  auto wait_vt = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[2][0]), "+v"(vf[2][1]),
                   "+v"(vf[3][0]), "+v"(vf[3][1]));
  };
  auto pv = [&](const int d0) {
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        o[d0 + dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dd][ks], pf[ks], o[d0 + dd], 0, 0, 0);
  };
  auto tile_sum = [](const f32x16& e) {      // packed adds, as a tree (no dependent chain)
    const f32x2 a = f32x2{e[0], e[1]} + f32x2{e[2], e[3]}, b = f32x2{e[4], e[5]} + f32x2{e[6], e[7]};
    const f32x2 c = f32x2{e[8], e[9]} + f32x2{e[10], e[11]}, d = f32x2{e[12], e[13]} + f32x2{e[14], e[15]};
    const f32x2 r = (a + b) + (c + d);
    return r[0] + r[1];
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

  BMHRL_STAMP(6)
  for (int t = 0; t + 1 < nt; ++t) {
    const int cur = t & 1;                     // V stage of tile t == K stage of tile t+2; K(t+1), V(t+1) use stage cur^1
    qk_issue((unsigned)(cur ^ 1) * (VST * 2)); // K(t+1) fragments: their LDS latency passes under the load issue below
    if (p.dbg != 1) {                          // (tuning aid: dbg 1 times the loop without its loads)
      if (t + 2 < nt) issue_k(t + 2, cur);     // K(t) was read in the previous iteration, V(t-1) too
      issue_v(t + 1, cur ^ 1);
    }
    const unsigned soff = (unsigned)cur * (VST * 2);
    {
      qk_chain([&] { read_vt(soff, H0{}); },     // V^T(t), d-tiles 0..3: wanted at the start of phase 2
               [&](const int i) {                // two exponentials of tile t at a time under the MFMAs of tile t+1
                 sc[2 * i] = __builtin_amdgcn_exp2f(sc[2 * i]);
                 sc[2 * i + 1] = __builtin_amdgcn_exp2f(sc[2 * i + 1]);
               });
      l_run += tile_sum(sc);
      pack_p();
    }
    wait_vt();
    pv(0);
    if constexpr (DK == 256) read_vt(soff, H1{});
    c_addr += BN * 4;
    // phase 2 VALU work (independent of the MFMAs around it, hides under them): row max of tile t+1, the rare rescale,
    // and the exponential arguments of tile t+1
    const bool slow = (slow_bits >> (t + 1)) & 1;    // wave-uniform, rare: per-key coefficients (masked / padding keys)
    float m_tile;
    if (!slow) {
      m_tile = raw_max(s_acc);
      args_fast(s_acc, sc, max_for_exp());           // with the max as it stands: nearly always the final one
    }
    if constexpr (DK == 256) {
      wait_vt();
      pv(4);
    }
    if (slow) {
      f32x4 cf[4], pn[4];
      read_coef(c_addr, cf, pn);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(pn[0]), "+v"(pn[1]), "+v"(pn[2]), "+v"(pn[3]));
      m_tile = scale_scores(s_acc, sc, (t + 1) * BN, cf, pn);
      args_slow(sc, max_for_exp());
    }
    maybe_rescale(m_tile, true, true, sc);

    // ---- the loads issued at the top of this iteration (K tile t+2, V tile t+1) have had the whole iteration to land;
    // the barrier publishes them and retires K stage cur^1 / V stage cur for the next refill
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  BMHRL_STAMP(7)
  {   // last tile: exponentials and O^T only
    const unsigned soff = (unsigned)((nt - 1) & 1) * (VST * 2);
    read_vt(soff, H0{});
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
    l_run += tile_sum(sc);
    pack_p();
    wait_vt();
    pv(0);
    if constexpr (DK == 256) {
      read_vt(soff, H1{});
      wait_vt();
      pv(4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // every wave is done with the V stages: the merge below reuses them
    asm volatile("" ::: "memory");
  }

  BMHRL_STAMP(8)
  // ---- merge + output.  The two waves of a query block hold partial states (O^T over all DK columns, row max, row sum)
  // for disjoint keys.  Each keeps HALF of the columns (wave ki: d-tiles [ki*NH, ki*NH+NH)) and sends the other half to
  // its partner through LDS, so both waves share the combine, the normalisation and the stores; the finished bf16 rows
  // go through a padded LDS image so that every store instruction writes whole 16-byte pieces of contiguous rows
  // (the O^T register layout holds one query row per lane: stored directly, one instruction touches 64 cache lines).
  static_assert(KW == 2, "pairwise exchange");
  constexpr int ND = DK / 32, NH = ND / 2;
  constexpr int XSLOTS = NH * 4 + 1;                       // float4 slots per lane: NH*16 accumulators + (m, l)
  constexpr int ROWB = NH * 64 + 16;                       // bytes per row of the output image (padded)
  constexpr int XCH_BYTES = QW * KW * XSLOTS * 64 * 16;
  static_assert(XCH_BYTES + QW * KW * 32 * ROWB <= LDS_KV, "exchange + output images reuse the K/V stages");
  l_run += __shfl_xor(l_run, 32, 64);   // the two 32-lane halves hold disjoint keys of the same query row
  f32x4* xch = reinterpret_cast<f32x4*>(smem_raw);
  f32x4* mine = xch + (qi * KW + ki) * XSLOTS * 64 + lane;
  const f32x4* theirs = xch + (qi * KW + (ki ^ 1)) * XSLOTS * 64 + lane;
  char* img = smem_raw + XCH_BYTES + (qi * KW + ki) * 32 * ROWB;
  const uint64_t seed = p.seed + ((p.dropout_p > 0.f && p.seed_dev) ? p.seed_dev[0] : 0ull);
  float m_all = 0.f, l_all = 0.f;
  auto send = [&](auto keep) {
    constexpr int SEND = (decltype(keep)::value ^ 1) * NH;
#pragma unroll
    for (int dd = 0; dd < NH; ++dd)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[SEND + dd][4 * g + j];
        mine[(dd * 4 + g) * 64] = v;
      }
    {
      f32x4 v;
      v[0] = m_run; v[1] = l_run; v[2] = 0.f; v[3] = 0.f;
      mine[NH * 4 * 64] = v;
    }
  };
  auto finish = [&](auto keep, auto drop) {
    constexpr int MY = decltype(keep)::value * NH;
    constexpr bool DROP = decltype(drop)::value;
    const f32x4 st = theirs[NH * 4 * 64];
    const float m = fmaxf(m_run, st[0]);
    const float ms = (m == -INFINITY) ? 0.f : m;
    float a1 = __builtin_amdgcn_exp2f(m_run - ms), a2 = __builtin_amdgcn_exp2f(st[0] - ms);
    m_all = m;
    l_all = l_run * a1 + st[1] * a2;
    const float inv = __builtin_amdgcn_rcpf(l_all);      // 1 ulp; the output is rounded to bf16
    a1 *= inv;
    a2 *= inv;
    const uint64_t ebase = ((uint64_t)b * p.Sq + q_row) * (uint64_t)(p.H * DK) + hd * DK + 4 * h;
    f32x4 got[NH * 4];                                     // all reads first: one LDS latency, not one per slot
#pragma unroll
    for (int sl = 0; sl < NH * 4; ++sl) got[sl] = theirs[sl * 64];
#pragma unroll
    for (int dd = 0; dd < NH; ++dd)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = o[MY + dd][4 * g + j] * a1 + got[dd * 4 + g][j] * a2;
          if constexpr (DROP) x *= dropout_scale(p.dropout_p, seed, ebase + 32 * (MY + dd) + 8 * g + j);
          w[j] = (bf16_t)x;
        }
        *reinterpret_cast<bf16x4*>(img + r32 * ROWB + (dd * 32 + 8 * g + 4 * h) * 2) = w;
      }
    // rows of the image -> global: CPRO 16-byte pieces per row, 64 / CPRO rows per instruction
    constexpr int CPRO = NH * 4, RPS = 64 / CPRO, NST = 32 / RPS;
    const int srow = lane / CPRO, sch = lane % CPRO;
    const int q0 = qt * (32 * QW) + qi * 32;
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
  };
  BMHRL_STAMP(9)
  if (ki == 0) send(std::integral_constant<int, 0>{});
  else send(std::integral_constant<int, 1>{});
  __syncthreads();
  using DropOff = std::integral_constant<bool, false>;
  using DropOn = std::integral_constant<bool, true>;
  if (p.dropout_p > 0.f) {
    if (ki == 0) finish(std::integral_constant<int, 0>{}, DropOn{});
    else finish(std::integral_constant<int, 1>{}, DropOn{});
  } else {
    if (ki == 0) finish(std::integral_constant<int, 0>{}, DropOff{});
    else finish(std::integral_constant<int, 1>{}, DropOff{});
  }
  if (ki == 0 && h == 0 && q_ok) {
    const long si = ((long)b * p.H + hd) * p.Sq + q_row;
    // statistics in natural-log units: P = exp(score - row_max) / row_sum.  A fully masked row keeps the exact
    // fill value so that the backward recomputation exp(-1e9 - row_max) is exp(0).
    p.row_max[si] = (m_all <= NEG_MASK * LOG2E) ? NEG_MASK : m_all * LN2;
    p.row_sum[si] = l_all;
  }
  BMHRL_STAMP(10)
}

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
__global__ void softmax_rows_kernel(const float* __restrict__ S, long lds, bf16_t* __restrict__ P, long ldp, long rows,
                                    int cols) {
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
  bf16_t* pr = P + row * ldp;
  for (int c = lane; c < cols; c += 64) pr[c] = (bf16_t)(__expf(s[c] - m) * inv);
}

}  // namespace

extern "C" int bmhrl_attention_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                   void* O, int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask,
                                   int64_t mask_sb, int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                                   int32_t dk, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                   bmhrl_stream_t stream) {
  constexpr int DK = 256;
  BMHRL_CHECK_ARG(Q && K && V && O && row_max && row_sum);
  BMHRL_CHECK_ARG(dk == DK);  // d_model 1024 / H 4 of the reference; other head sizes use the materialised path
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0 && Sk <= 2048);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * DK && ldk >= (int64_t)H * DK && ldv >= (int64_t)H * DK && ldo >= (int64_t)H * DK);
  BMHRL_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0);
  BMHRL_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
  AttnArgs a;
  a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
  a.O = (bf16_t*)O; a.ldo = ldo; a.row_max = row_max; a.row_sum = row_sum;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = mask_sq;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = dropout_p; a.seed = seed; a.seed_dev = seed_dev;
  a.k_hs = DK; a.v_hs = DK;
  a.q_tiles = (Sq + 32 * QW - 1) / (32 * QW);
  a.dbg = getenv("BMHRL_ATTN_DBG") ? atoi(getenv("BMHRL_ATTN_DBG")) : 0;
  BMHRL_CHECK_ARG((int64_t)B * H * a.q_tiles * H * a.q_tiles < (1ll << 32));
  set_block_map(a);
  dim3 grid((unsigned)(B * H * a.q_tiles)), block(NT);
  if (mask != nullptr && mask_sq != 0) hipLaunchKernelGGL((attn_fwd_kernel<DK, true>), grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<DK, false>), grid, block, 0, (hipStream_t)stream, a);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_attention_shared128_fwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo,
                                             float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                                             int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale,
                                             bmhrl_stream_t stream) {
  constexpr int DK = 128;
  BMHRL_CHECK_ARG(Qp && X && ctx && row_max && row_sum);
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0 && Sk <= 1024);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * DK && ldx >= DK && ldo >= (int64_t)H * DK);
  BMHRL_CHECK_ARG((((uintptr_t)Qp | (uintptr_t)X | (uintptr_t)ctx) & 15) == 0);
  AttnArgs a;
  a.Q = (const bf16_t*)Qp; a.ldq = ldq; a.K = (const bf16_t*)X; a.ldk = ldx; a.V = (const bf16_t*)X; a.ldv = ldx;
  a.O = (bf16_t*)ctx; a.ldo = ldo; a.row_max = row_max; a.row_sum = row_sum;
  a.mask = mask; a.mask_sb = mask_sb; a.mask_sq = 0;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dropout_p = 0.f; a.seed = 0; a.seed_dev = nullptr;
  a.k_hs = 0; a.v_hs = 0;                       // one 128-wide key / value row for every head
  a.q_tiles = (Sq + 32 * QW - 1) / (32 * QW);
  a.dbg = 0;
  BMHRL_CHECK_ARG((int64_t)B * H * a.q_tiles * H * a.q_tiles < (1ll << 32));
  set_block_map(a);
  dim3 grid((unsigned)(B * H * a.q_tiles)), block(NT);
  hipLaunchKernelGGL((attn_fwd_kernel<DK, false>), grid, block, 0, (hipStream_t)stream, a);
#ifdef BMHRL_ATTN_TRACE
  if (getenv("BMHRL_ATTN_TRACE")) {
    long long h[2][16];
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_trace), sizeof(h));
    for (int w = 0; w < 2; ++w) {
      fprintf(stderr, "attn128 trace (%s block, Sq %d Sk %d):", w ? "last" : "first", Sq, Sk);
      for (int i = 1; i <= 10; ++i) fprintf(stderr, " %lld", h[w][i] - h[w][i - 1]);
      fprintf(stderr, "  total %lld\n", h[w][10] - h[w][0]);
    }
  }
#endif
  return hip_status(hipGetLastError());
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

extern "C" int bmhrl_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int64_t rows, int32_t cols,
                                  bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(S && P && rows > 0 && cols > 0);
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipLaunchKernelGGL(softmax_rows_kernel, grid, block, 0, (hipStream_t)stream, S, (long)lds, (bf16_t*)P, (long)ldp,
                     (long)rows, cols);
  return hip_status(hipGetLastError());
}
