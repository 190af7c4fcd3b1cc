// "Pair" form of the shared-key/value attention forward (head dimension 128, absorbed-projection form: the 128-wide audio rows
// are keys AND values of every head; model/multihead_attention.py:7-31 as restated in DESIGN.md section 9) for the shapes of
// BASELINE configs[1]: the cross-modal V<-A attention has 16 * 4 * 256 = 16 384 (query row, head) slots against 800 keys -- one
// wave per SIMD of the chip gets 32 slots x 400 keys or 64 slots x 200 keys, and nothing else to hide anything behind.
//
// What r03's ablation builds of the (2 query blocks x 2 key halves) kernel showed: one wave per SIMD issues
// in order, every instruction costs its issue slot, and most of a key tile's cost does not depend on how many MFMAs the tile
// feeds -- LDS fragment reads, the workgroup barrier + wait, loop control.  This kernel changes the split instead of the schedule:
//
//   * a wave owns TWO 32-slot query blocks (the same 32 query rows of two heads) and ONE key split (every fourth 32-key tile):
//     each K row fragment and each V^T fragment read from LDS feeds two MFMAs (S^T of both blocks / O^T of both blocks) --
//     24 LDS fragment reads per 32 MFMAs instead of per 16 --, and a wave runs 7 (6) iterations of 32 MFMAs instead of 13 of 16;
//   * the four waves of a workgroup are four KEY SPLITS of the same 64 slots, and a wave stages exactly the 32 key rows it reads
//     (8 direct-to-LDS pieces per tile into a private ring of 4 stages): no wave ever reads what another wave loaded, so the
//     main loop has NO workgroup barrier -- a wave only waits for its own pieces (counted vmcnt) -- and the waves drift apart
//     instead of queueing their loads on the CU's one vector-memory path at the same instant after every barrier;
//   * Q' goes through LDS once (16 KiB per workgroup, whole 256-byte rows by direct-to-LDS loads; the fragments of all four
//     waves are ds_read_b128 of that image) instead of 32-byte fragment-shaped global loads per lane and wave;
//   * S^T accumulators live in arch VGPRs (the softmax reads them in place), O^T and the Q'^T fragments in the accumulator half
//     of the register file: the MFMAs are written out with their register classes (inline asm), every hazard between them and
//     the code around them is covered by construction (see the notes at the asm helpers);
//   * the four key splits merge once at the end: every wave keeps 2 of the 8 (block, d-tile pair) pieces and sends the other 6
//     through its own ring (no barrier before the writes), one barrier, combine + normalise + 128-byte row stores.
//
// Formulation, LDS image, swizzles, mask handling (ballots of 4-key groups, per-key path only for tiles with a masked or
// padding key, tiles behind the last valid key skipped), lazy rescale: as attention_fwd.h, whose helpers this file uses.
#pragma once
#include "attention_fwd.h"

namespace {

struct PairArgs {
  const bf16_t* Q; long ldq;
  const bf16_t* X; long ldx;
  bf16_t* O; long ldo;
  float* row_max; float* row_sum;
  const uint8_t* mask; long mask_sb;
  int B, H, Sq, Sk;
  float scale;
  int q_tiles, per_b, map_mode;
  unsigned magic_perb, magic_qt;
  int dbg;
};

constexpr int P_NS = 4;                           // stages of a wave's private ring
constexpr int P_STAGE = 32 * 128 * 2;             // bytes of a stage: 32 key rows of 256 bytes
constexpr int P_RING = P_NS * P_STAGE;            // 32 KiB per wave
constexpr int P_QIMG = 4 * P_RING;                // Q' image: 64 slots x 256 bytes; later the output image (padded rows)
constexpr int P_IMG_ROWB = 128 + 16;              // bytes per row of a wave's output image (32 rows x 64 columns bf16, padded)
constexpr int P_IMG_BYTES = 4 * 32 * P_IMG_ROWB;  // 18 432 >= the 16 KiB of the Q' image
constexpr int P_WORDS = 40;                       // ballot words of 256 keys
constexpr int P_BAL = P_QIMG + P_IMG_BYTES;
constexpr int P_ML = P_BAL + 2 * P_WORDS * 8;     // (max, sum) of every (wave, block, lane)
constexpr int P_KEEP = P_ML + 4 * 2 * 64 * 8;     // one byte per 4-key group: bit j = key 4 g + j is valid and not masked
constexpr int P_MAX_SK = P_WORDS * 256 - 128;
constexpr int P_LDS = P_KEEP + P_WORDS * 64;      // (256 keys per ballot word = 64 groups)
static_assert(P_LDS <= 160 * 1024, "one workgroup per CU");

// ---- MFMAs with explicit register classes.  Hazards (the compiler pads nothing around an asm statement):
//  * MFMA result -> VALU / accvgpr_read of it: the loop reads S^T no earlier than eight MFMAs after its chain, O^T only behind
//    the final barrier; everywhere else (first tile, rare paths) the reader sits behind pair_settle() (two s_nop 15);
//  * MFMA result -> the next MFMA's C of the same registers (the accumulate chains): no wait states needed;
//  * VALU write of an A / B operand (the bf16 P^T packs) -> MFMA: ordinary in-order operand read, and the packs sit at least
//    one MFMA gap ahead of their consumer;
//  * ds_read results: hand-counted lgkmcnt waits that name the registers ("+v"), as in attention_fwd.h.
__device__ __forceinline__ void pair_mfma_o(f32x16& c, const bf16x8& a, const bf16x8& b) {      // c (accumulator file) += a . b
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pair_mfma_s0(f32x16& c, const bf16x8& a, const bf16x8& bq) {    // c (VGPRs) = a . bq, bq in the accumulator file
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(bq));
}
__device__ __forceinline__ void pair_mfma_s(f32x16& c, const bf16x8& a, const bf16x8& bq) {     // c (VGPRs) += a . bq
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(bq));
}
__device__ __forceinline__ void pair_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ bf16x8 asm_ldsb128(unsigned addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

#ifndef BMHRL_PABL
#define BMHRL_PABL 0     // timing ablations of the loop (tests/kbench/build_pabl.sh only; any bit makes the results wrong): 1 no pieces
#endif                   // in the loop, 2 no exp, 4 no V^T reads, 8 no K reads, 16 no phase-2 VALU, 32 no S^T MFMAs, 64 no O^T MFMAs,
                         // 128 no bf16 packs

__global__ __launch_bounds__(256, 1) void attn_pair128_kernel(const PairArgs p) {
  constexpr int DK = 128, NS = P_NS;
  __shared__ __attribute__((aligned(16))) char smem_raw[P_LDS];
  // P_BAL: per 256 keys, one 64-bit word of 4-key groups with a masked / padding key ("slow"), then one of groups with a valid key
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  constexpr float RESCALE_THR = 8.f;

  BMHRL_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int ki = __builtin_amdgcn_readfirstlane(tid >> 6);               // this wave's key split: tiles ki, ki + 4, ...
  const int r32 = lane & 31, h = lane >> 5;
  const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int hi = lane >> 4, pch = lane & 15;                             // staging: row inside a 4-row piece, 16-byte chunk

  // workgroup -> (batch row, head pair, 32-row query tile); every workgroup of a batch row on one XCD when B % 8 == 0
  int b, rem;
  if (p.map_mode == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q1 = fast_div(idx, p.per_b, p.magic_perb);
    b = xcd + 8 * q1;
    rem = idx - q1 * p.per_b;
  } else {
    b = fast_div((int)blockIdx.x, p.per_b, p.magic_perb);
    rem = (int)blockIdx.x - b * p.per_b;
  }
  const int hp = fast_div(rem, p.q_tiles, p.magic_qt), qt = rem - hp * p.q_tiles;
  const int q0 = qt * 32, q_row = q0 + r32;
  const bool q_ok = q_row < p.Sq;

  const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X + (long)b * p.Sk * p.ldx);
  const uint8_t* __restrict__ mrow_b = p.mask ? p.mask + (long)b * p.mask_sb : nullptr;
  const bool mask_al4 = (reinterpret_cast<uintptr_t>(mrow_b) & 3) == 0;
  const int Sk = p.Sk, dbg = p.dbg;
  const long ldx2 = p.ldx * 2;
  const int nt_all = (Sk + 31) >> 5;                                     // 32-key tiles of the batch row

  // ---- requests, in the order of their first use: the lane's first mask word, Q', then the wave's first tiles
  // (the mask word is loaded by hand: an ordinary load would make the compiler drain every direct-to-LDS piece requested
  // behind it at its first use -- it cannot tell that the word is the OLDEST request; the wait below names it)
  uint32_t v_pre = 0x01010101u;
  const bool pre_ok = mrow_b != nullptr && mask_al4 && 4 * tid + 4 <= p.Sk;
  if (pre_ok) asm volatile("global_load_dword %0, %1, off" : "=v"(v_pre) : "v"(mrow_b + 4 * tid) : "memory");
  {
    // Q' image: slot row R = 32 qb + r holds query row q0 + r of head 2 hp + qb; wave w stages rows 16 w .. 16 w + 15 (4 pieces
    // of 4 rows); logical chunk ^ swizzle(r) lands at physical chunk position pch (the K fragment reads' swizzle)
    const int qb_w = ki >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * (ki & 1) + 4 * i + hi;
      const int grow = min(q0 + r, p.Sq - 1);
      const int lc = pch ^ ((hi << 2) | (i & 3));
      const char* src = reinterpret_cast<const char*>(p.Q + ((long)b * p.Sq + grow) * p.ldq + (2 * hp + qb_w) * DK) + (lc << 4);
      glds16<0>(src, reinterpret_cast<bf16_t*>(smem_raw + P_QIMG + ki * 4096 + i * 1024));
    }
  }
  // tile staging: piece i = j2 + 4 u (j2 = i & 3) covers rows 4 i .. 4 i + 3 of the wave's 32; lane (hi, pch) loads logical chunk
  // pch ^ ((hi << 2) | j2) of row 4 i + hi.  One uniform base per half tile + a per-lane offset fixed for the launch + the
  // instruction's immediate (added to the global AND the LDS address, hence subtracted here).
  unsigned toff[4];
#pragma unroll
  for (int j2 = 0; j2 < 4; ++j2)
    toff[j2] = (unsigned)((4 * j2 + hi) * (int)p.ldx * 2 + ((pch ^ ((hi << 2) | j2)) << 4) - 1024 * j2);
  bf16_t* const ring = reinterpret_cast<bf16_t*>(smem_raw + ki * P_RING);
  // piece I of a WHOLE tile whose first row is at base0 (base1: its 17th row), stage at sdst0 -- one instruction + the M0 write
  auto whole_piece = [&](auto i_, const char* base0, const char* base1, bf16_t* sdst0) {
    constexpr int I = decltype(i_)::value;
    unsigned o = toff[I & 3];
    asm volatile("" : "+v"(o));      // keep the 32-bit lane offset as it is: (SGPR base + VGPR offset) addressing
    glds16<1024 * (I & 3)>(((I >> 2) ? base1 : base0) + o, sdst0 + (I >> 2) * 2048);
  };
  auto issue_tile = [&](const int T, const int stage) {                  // all 8 pieces of global tile T into `stage`
    const int k0 = T * 32;
    bf16_t* sdst0 = ring + stage * (P_STAGE / 2);
    if (k0 + 32 <= Sk) {
      const char* base0 = Xb + (long)k0 * ldx2;
      const char* base1 = base0 + 16 * ldx2;
      static_for<0, 8>([&](auto i) { whole_piece(i, base0, base1, sdst0); });
    } else {   // ragged last tile: clamp the key row (its score gets -inf, so P is exactly 0 there; the value must be finite)
      static_for<0, 8>([&](auto i) {
        constexpr int I = decltype(i)::value;
        const int gr = min(k0 + 4 * I + hi, Sk - 1);
        const int lc = pch ^ ((hi << 2) | (I & 3));
        glds16<0>(Xb + (unsigned)(gr * (int)ldx2 + (lc << 4)), sdst0 + I * 512);
      });
    }
  };
  // The number of tiles is not known before the ballots (they trim fully masked tails): request the first three of the
  // untrimmed row; a tile behind the last valid key is simply never read.
  // (two of them here, the third under the first S^T chain: every CU's workgroup is in its prologue at the same time, and what
  // is requested here goes out at the ~36 bytes / cycle a CU gets from L2 -- 3.1 k cycles for Q' + three tiles of four waves,
  // all of it in front of the first MFMA)
  const int n_all = nt_all > ki ? (nt_all - ki + 3) >> 2 : 0;             // this wave's tiles of the untrimmed row
  const int n_pro = min(n_all, NS - 1);
  constexpr int N_EARLY = NS - 2;
  for (int j = 0; j < min(n_pro, N_EARLY); ++j) issue_tile(ki + 4 * j, j);
  BMHRL_STAMP(1)

  // per-lane LDS byte addresses (stage 0 of this wave's ring)
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  unsigned k_addr[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) k_addr[st] = lds0 + ki * P_RING + r32 * 256 + (((2 * st + h) ^ swz(r32)) << 4);
  unsigned v_addr[4][2];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
      const int lc = 4 * dd + 2 * g1 + (p4 >> 1);
      const int pc = lc ^ ((q4 << 2) | ((h + 2 * sec) & 3));
      v_addr[dd][sec] = lds0 + ki * P_RING + (4 * h + q4) * 256 + (pc << 4) + ((p4 & 1) << 3);
    }

  // ---- the mask word and the Q' pieces are older than every tile piece of this wave
  if (n_pro == NS - 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v_pre) : "n"(8 * N_EARLY) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(v_pre) :: "memory");
  // key mask ballots (as attention_fwd.h): which 4-key groups hold a masked / padding key, which a valid key.  A thread owns four
  // consecutive keys per round, a wave's ballot covers 256 keys.  (LDS accesses by hand: the compiler orders every LDS access
  // it knows about behind ALL outstanding direct-to-LDS loads, which would drain the tiles in flight.)
  const int n_words = (nt_all * 32 + 255) >> 8;
  for (int j = 0; j * 4 < n_words; ++j) {
    const int i0 = 4 * (tid + 256 * j);
    uint32_t v = 0x01010101u;
    if (j == 0 && pre_ok) {
      v = v_pre;
    } else if (mrow_b != nullptr && i0 < p.Sk) {      // (Sk > 1024 or an unaligned mask: ordinary loads, the tiles in flight drain)
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
    unsigned nib = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool keep = i0 + e < p.Sk && ((v >> (8 * e)) & 0xffu) != 0;
      any_slow |= !keep;
      any_valid |= keep;
      nib |= (keep ? 1u : 0u) << e;
    }
    if (tid + 256 * j < P_WORDS * 64) {
      const unsigned ka = lds0 + P_KEEP + tid + 256 * j;
      asm volatile("ds_write_b8 %0, %1" ::"v"(ka), "v"(nib) : "memory");
    }
    const uint64_t bs = __ballot(any_slow), bv = __ballot(any_valid);
    if (lane == 0 && 4 * j + ki < P_WORDS) {
      const unsigned wa = lds0 + P_BAL + 8 * (4 * j + ki);
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%3" ::"v"(wa), "v"(bs), "v"(bv), "n"(8 * P_WORDS) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BMHRL_STAMP(2)

  // Q'^T fragments of both blocks: lane (q = r32, h) holds Q'[q][16 st + 8 h .. + 8) -- the K fragment read pattern on the Q' image
  bf16x8 qf[2][8];
  {
    const unsigned qdelta = (unsigned)(P_QIMG - ki * P_RING);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const unsigned a = k_addr[st] + qdelta;
      qf[0][st] = asm_ldsb128<0>(a);
      qf[1][st] = asm_ldsb128<8192>(a);
    }
  }
  // tiles to visit: up to the last one that holds a valid key (all of them when the batch row has no valid key at all)
  auto lds_u64 = [&](const unsigned addr) {
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };
  int nt = nt_all;
  if (mrow_b != nullptr) {
    int last_group = -1;
    for (int w = 0; w < n_words && w < P_WORDS; ++w) {
      const uint64_t bv = lds_u64(lds0 + P_BAL + 8 * P_WORDS + 8 * w);
      if (bv != 0ull) last_group = 64 * w + 63 - __builtin_clzll(bv);
    }
    if (last_group >= 0) nt = (4 * last_group) / 32 + 1;
  }
  nt = __builtin_amdgcn_readfirstlane(nt);
  const int n_loc = nt > ki ? (nt - ki + 3) >> 2 : 0;                    // this wave's tiles: local j <-> global tile ki + 4 j
  // bit i of the window: local tile (base + i) has a masked or padding key (wave-uniform)
  auto slow_window = [&](const int base) -> uint64_t {
    const int g = ki + 4 * (base + lane);                                // 32-key group; 8 of them per ballot word
    const bool in = g < nt_all && (g >> 3) < P_WORDS;
    const uint64_t w = lds_u64(lds0 + P_BAL + 8 * (in ? (g >> 3) : 0));
    return __ballot(in && ((w >> (8 * (g & 7))) & 0xffull) != 0ull);
  };
  uint64_t slow_bits = slow_window(0);
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[0][4]), "+v"(qf[0][5]), "+v"(qf[0][6]),
                 "+v"(qf[0][7]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]), "+v"(qf[1][4]), "+v"(qf[1][5]),
                 "+v"(qf[1][6]), "+v"(qf[1][7]));
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int st = 0; st < 8; ++st) asm volatile("" : "+a"(qf[qb][st]));   // loop invariant B operands: accumulator half of the file

  f32x16 o[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][d][r] = 0.f;
      asm volatile("" : "+a"(o[qb][d]));
    }
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float c_log2 = p.scale * LOG2E;

  // scale + mask the raw scores of one tile (rare path: the tile holds a masked or padding key) through the keep bits the
  // prologue left in LDS -- 8 bytes cover the tile's 32 keys, the lane's groups are bytes h + 2 g --; returns the tile maximum.
  // (No global load here: an ordinary load would make the compiler drain the tile pieces in flight.)
  auto scale_scores = [&](const f32x16& raw, float (&sc)[16], const int k0) {
    float m_tile = -INFINITY;
    const uint64_t m8 = lds_u64(lds0 + P_KEEP + (k0 >> 2));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned nib = (unsigned)(m8 >> (8 * (h + 2 * g))) & 0xfu;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = k0 + 4 * h + 8 * g + j;
        const bool keep = (nib >> j) & 1u;
        const float v = keep ? raw[4 * g + j] * c_log2 : (key < Sk ? NEG_MASK * LOG2E : -INFINITY);
        sc[4 * g + j] = v;
        m_tile = fmaxf(m_tile, v);
      }
    }
    return pair_max(m_tile);
  };
  // lazy rescale of one block (attention_fwd.h: everything accumulated so far is at the old max, P of the new tile has not been
  // exponentiated yet).  O^T lives in the accumulator file: the multiplies go through arch VGPRs (rare).
  auto maybe_rescale = [&](const int qb, const float m_tile, const bool have_o, const bool fix_args, float (&args)[16]) {
    if (__any(m_tile > m_run[qb] + RESCALE_THR)) {
      const float m_new = fmaxf(m_run[qb], m_tile);
      const float alpha = (m_run[qb] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      l_run[qb] *= alpha;
      if (fix_args) {
        const float shift = ((m_run[qb] == -INFINITY) ? 0.f : m_run[qb]) - ((m_new == -INFINITY) ? 0.f : m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) args[r] += shift;
      }
      if (have_o) {
        pair_settle();                                     // the O^T MFMAs issued last have retired
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          f32x16 t = o[qb][d];
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] *= alpha;
          o[qb][d] = t;
          asm volatile("" : "+a"(o[qb][d]));
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // accvgpr_write -> MFMA C
      }
      m_run[qb] = m_new;
    }
  };
  auto max_for_exp = [&](const int qb) { return (m_run[qb] == -INFINITY) ? 0.f : m_run[qb]; };
  auto exp_inplace = [&](float& v) {
    float x = __builtin_amdgcn_exp2f(v);
    asm volatile("" : "+v"(x));
    v = x;
  };

  f32x16 s_acc[2];
  float sc[2][16];
  bf16x8 kf[8], vf[4][2], pf[2][2];
  float neg_m[2] = {0.f, 0.f}, thr_raw[2] = {0.f, 0.f};

  if (n_loc > 0) {
    // ---- first tile: its pieces have landed when at most the later prologue tiles are outstanding
    if (n_pro == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (N_EARLY - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < 8; ++st) kf[st] = asm_ldsb128<0>(k_addr[st]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
    BMHRL_SB();
    {
      // the third prologue tile's pieces, one per K fragment of the first chain (whole tile; a ragged one -- a row of 65 .. 95
      // keys per split -- goes out in one go in front of the chain)
      const int T2 = ki + 4 * (NS - 2);
      bool late = n_pro == NS - 1;
      if (late && T2 * 32 + 32 > Sk) {
        issue_tile(T2, NS - 2);
        late = false;
      }
      const char* const base0 = Xb + (long)T2 * 32 * ldx2;
      const char* const base1 = base0 + 16 * ldx2;
      bf16_t* const sdst0 = ring + (NS - 2) * (P_STAGE / 2);
      static_for<0, 8>([&](auto st_) {
        constexpr int ST = decltype(st_)::value;
        if constexpr (ST == 0) {
          pair_mfma_s0(s_acc[0], kf[0], qf[0][0]);
          pair_mfma_s0(s_acc[1], kf[0], qf[1][0]);
        } else {
          pair_mfma_s(s_acc[0], kf[ST], qf[0][ST]);
          pair_mfma_s(s_acc[1], kf[ST], qf[1][ST]);
        }
        if (late) whole_piece(st_, base0, base1, sdst0);
        BMHRL_SB();
      });
    }
    pair_settle();
    BMHRL_SB();
    BMHRL_STAMP(3)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (slow_bits & 1) {     // masked / padding keys in the wave's first tile
        const float m_tile = scale_scores(s_acc[qb], sc[qb], 32 * ki);
        maybe_rescale(qb, m_tile, false, false, sc[qb]);
        const float m_use = max_for_exp(qb);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[qb][r] -= m_use;
      } else {
        float rmx = s_acc[qb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) rmx = fmaxf(rmx, s_acc[qb][r]);
        m_run[qb] = pair_max(rmx) * c_log2;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[qb][r] = fmaf(s_acc[qb][r], c_log2, -m_run[qb]);
      }
      neg_m[qb] = -max_for_exp(qb);
      thr_raw[qb] = (m_run[qb] + RESCALE_THR) / c_log2;
    }
    // the second tile's K fragments (a stale stage when the wave has one tile only: never used)
    if (n_pro == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 3)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < 8; ++st) kf[st] = asm_ldsb128<P_STAGE>(k_addr[st]);
    BMHRL_STAMP(4)

    // ---- main loop.  Iteration j (stage S = j % NS):  phase 1  S^T(j+1) of both blocks  ||  P(j) = exp2(args), V^T(j) fragment
    // reads, the pieces of tile j + 3, first half of the bf16 P^T operands;  phase 2  O^T += V^T(j) . P^T(j) of both blocks  ||
    // second half of P^T, row sums of P(j), K(j+2) fragment reads, row maxima and exponential arguments of tile j + 1.
    // LDS reads in program order (hand-counted lgkmcnt): phase 1 even gap I: the two transposed reads of vf[(I>>1)&3][I>>3];
    // phase 2 even gap J: K(j+2) fragment J / 2.  MFMA I of phase 1 (I even) needs kf[I / 2]: behind it 7 - I/2 K reads + I
    // transposed reads; MFMA J of phase 2 (J even) needs vf[(J>>1)&3][J>>3]: behind it 14 - J transposed reads + J/2 K reads.
    auto iter = [&](auto s_, const int j) {
      constexpr int S = decltype(s_)::value;
      constexpr int SOFF = S * P_STAGE, KOFF2 = ((S + 2) % NS) * P_STAGE, STG3 = (S + 3) % NS;
      if constexpr (S == NS - 1) {
        if ((j & 63) == 63) slow_bits = slow_window(j + 1);
      }
      // tile j + 3 -> stage (S + 3) % NS, one piece per odd MFMA gap of phase 1; a ragged tile (the last of a row whose length
      // is no multiple of 32) goes out here in one go through the clamping path, so that the spread pieces carry no branch on it
      const int T3 = ki + 4 * (j + 3);
      if (j == 1) { BMHRL_STAMP(9) }
      bool load = (j + 3 < n_loc) && dbg != 1 && !(BMHRL_PABL & 1);
      if (load && T3 * 32 + 32 > Sk) {
        issue_tile(T3, STG3);
        load = false;
      }
      const char* const base0 = Xb + (long)T3 * 32 * ldx2;               // uniform: rows of the tile / of its second half
      const char* const base1 = base0 + 16 * ldx2;
      bf16_t* const sdst0 = ring + STG3 * (P_STAGE / 2);
      const bool ragged_now = j + 3 < n_loc && !load;
      float part[2] = {0.f, 0.f}, rmx[2] = {-INFINITY, -INFINITY};
      const bool slow = (slow_bits >> ((j + 1) & 63)) & 1;
      static_for<0, 16>([&](auto i_) {
        constexpr int I = decltype(i_)::value;
        constexpr int ST = I >> 1, QB = I & 1;
        if constexpr (QB == 0) {
          if constexpr (BMHRL_PABL & (4 | 8)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[ST]));
          else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kf[ST]) : "n"(7 + ST));
          BMHRL_SB();
        }
        if constexpr (!(BMHRL_PABL & 32)) {
          if constexpr (ST == 0) pair_mfma_s0(s_acc[QB], kf[0], qf[QB][0]);
          else pair_mfma_s(s_acc[QB], kf[ST], qf[QB][ST]);
        }
        if constexpr (!(BMHRL_PABL & 2)) {
          exp_inplace(sc[0][I]);
          exp_inplace(sc[1][I]);
        }
        if constexpr (QB == 0) {
          constexpr int DD = (I >> 1) & 3, KS = I >> 3;
          if constexpr (!(BMHRL_PABL & 4))
            vf[DD][KS] = join8(asm_tr4<SOFF + KS * 16 * DK * 2>(v_addr[DD][0]), asm_tr4<SOFF + (KS * 16 + 8) * DK * 2>(v_addr[DD][1]));
        } else {
          if (load) whole_piece(std::integral_constant<int, (I >> 1)>{}, base0, base1, sdst0);
        }
        if constexpr (I >= 8 && !(BMHRL_PABL & 128)) {     // elements 0..7 of both blocks are exponentiated by gap 7
          constexpr int X = I - 8, PB = X & 1, PR = X >> 1;
          pf[PB][0][2 * PR] = (bf16_t)sc[PB][2 * PR];
          pf[PB][0][2 * PR + 1] = (bf16_t)sc[PB][2 * PR + 1];
        }
        BMHRL_SB();
      });
      // tile j + 2 has landed when at most the pieces of tile j + 3 are outstanding
      if (j == 1) { BMHRL_STAMP(10) }
      if (load || ragged_now) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (j == 1) { BMHRL_STAMP(11) }
      auto arg = [&](const int qb, const int e) {
        float x = fmaf(s_acc[qb][e], c_log2, neg_m[qb]);
        asm volatile("" : "+v"(x));
        sc[qb][e] = x;
      };
      static_for<0, 16>([&](auto j_) {
        constexpr int J = decltype(j_)::value;
        constexpr int KS = J >> 3, DD = (J >> 1) & 3, QB = J & 1;
        if constexpr (QB == 0) {
          if constexpr (BMHRL_PABL & (4 | 8)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[DD][KS]));
          else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(vf[DD][KS]) : "n"(14 - J / 2));
          BMHRL_SB();
        }
        if constexpr (!(BMHRL_PABL & 64)) pair_mfma_o(o[QB][DD], vf[DD][KS], pf[QB][KS]);
        if constexpr (QB == 0 && !(BMHRL_PABL & 8))
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[J >> 1]) : "v"(k_addr[J >> 1]), "n"(KOFF2));
        if constexpr (BMHRL_PABL & 16) {
          BMHRL_SB();
        } else if constexpr (J < 8) {
          constexpr int PB = J & 1, PR = J >> 1;           // second half of P^T: elements 8 + 2 PR, 9 + 2 PR of block PB
          if constexpr (!(BMHRL_PABL & 128)) {
            pf[PB][1][2 * PR] = (bf16_t)sc[PB][8 + 2 * PR];
            pf[PB][1][2 * PR + 1] = (bf16_t)sc[PB][9 + 2 * PR];
          }
          part[0] += sc[0][2 * J];
          part[0] += sc[0][2 * J + 1];
          part[1] += sc[1][2 * J];
          part[1] += sc[1][2 * J + 1];
          asm volatile("" : "+v"(part[0]), "+v"(part[1]));
        } else {
          constexpr int E = 2 * (J - 8);                   // (S^T retired eight MFMAs ago; the slots' P are summed and packed)
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx[0]) : "v"(s_acc[0][E]), "v"(s_acc[0][E + 1]));
          asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx[1]) : "v"(s_acc[1][E]), "v"(s_acc[1][E + 1]));
          arg(0, E); arg(0, E + 1); arg(1, E); arg(1, E + 1);
        }
        BMHRL_SB();
      });
      if (j == 1) { BMHRL_STAMP(12) }
      l_run[0] += part[0];
      l_run[1] += part[1];
      if (slow || __any(rmx[0] > thr_raw[0] || rmx[1] > thr_raw[1])) {   // rare: masked / padding keys, or a maximum grew by > 2^8
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          float m_tile = pair_max(rmx[qb]) * c_log2;
          if (slow) {
            m_tile = scale_scores(s_acc[qb], sc[qb], 32 * (ki + 4 * (j + 1)));
            const float m_use = -neg_m[qb];
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[qb][r] -= m_use;
          }
          maybe_rescale(qb, m_tile, true, true, sc[qb]);
          neg_m[qb] = -max_for_exp(qb);
          thr_raw[qb] = (m_run[qb] + RESCALE_THR) / c_log2;
        }
      }
    };
    {
      int j = 0;
      while (true) {
        bool done = false;
        static_for<0, NS>([&](auto s_) {
          if (!done) {
            if (j + 1 < n_loc) { iter(s_, j); ++j; }
            else done = true;
          }
        });
        if (done) break;
      }
    }
    BMHRL_STAMP(5)
    {   // last tile: exponentials and O^T only
      const unsigned soff = (unsigned)((n_loc - 1) % NS) * P_STAGE;
      bf16x4 t0[8], t1[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int dd = v & 3, ks = v >> 2;
        t0[v] = asm_tr4<0>(v_addr[dd][0] + soff + ks * 16 * DK * 2);
        t1[v] = asm_tr4<0>(v_addr[dd][1] + soff + (ks * 16 + 8) * DK * 2);
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[qb][r] = __builtin_amdgcn_exp2f(sc[qb][r]);
          part += sc[qb][r];
        }
        l_run[qb] += part;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pf[qb][0][e] = (bf16_t)sc[qb][e];
          pf[qb][1][e] = (bf16_t)sc[qb][8 + e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t0[4]), "+v"(t0[5]), "+v"(t0[6]), "+v"(t0[7]),
                     "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3]), "+v"(t1[4]), "+v"(t1[5]), "+v"(t1[6]), "+v"(t1[7]));
      BMHRL_SB();
#pragma unroll
      for (int v = 0; v < 8; ++v) vf[v & 3][v >> 2] = join8(t0[v], t1[v]);
      asm volatile("s_nop 1" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          pair_mfma_o(o[0][dd], vf[dd][ks], pf[0][ks]);
          pair_mfma_o(o[1][dd], vf[dd][ks], pf[1][ks]);
        }
      pair_settle();
    }
  }
  // every piece this wave requested has landed and every LDS read of its ring has returned: the ring is the wave's own
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  BMHRL_STAMP(6)

  // ---- merge of the four key splits + output.  Owner wave w keeps block w & 1, d-tiles 2 (w >> 1) and 2 (w >> 1) + 1; a wave
  // writes what the three other owners need into ITS OWN ring (nobody else touches it before the barrier), then one barrier.
  l_run[0] += __shfl_xor(l_run[0], 32, 64);     // the two 32-lane halves hold disjoint keys of the same query row
  l_run[1] += __shfl_xor(l_run[1], 32, 64);
  float* ml = reinterpret_cast<float*>(smem_raw + P_ML);
  {
    f32x4* mine = reinterpret_cast<f32x4*>(smem_raw + ki * P_RING);
    static_for<0, 4>([&](auto ko_) {
      constexpr int KO = decltype(ko_)::value;
      if (ki != KO) {
        const int n = KO < ki ? KO : KO - 1;
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
          const f32x16 t = o[KO & 1][2 * (KO >> 1) + d2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[4 * g + e];
            mine[((n * 2 + d2) * 4 + g) * 64 + lane] = v;
          }
        }
      }
    });
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      ml[((ki * 2 + qb) * 64 + lane) * 2] = m_run[qb];
      ml[((ki * 2 + qb) * 64 + lane) * 2 + 1] = l_run[qb];
    }
  }
  __syncthreads();
  BMHRL_STAMP(7)
  static_for<0, 4>([&](auto kw_) {
    constexpr int KW_ = decltype(kw_)::value;
    if (ki == KW_) {
      constexpr int QB = KW_ & 1, DH = KW_ >> 1;
      float ms[4], ls[4], m = -INFINITY;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ms[k] = ml[((k * 2 + QB) * 64 + lane) * 2];
        ls[k] = ml[((k * 2 + QB) * 64 + lane) * 2 + 1];
        m = fmaxf(m, ms[k]);
      }
      const float mz = (m == -INFINITY) ? 0.f : m;
      float a[4], l_all = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = __builtin_amdgcn_exp2f(ms[k] - mz);
        l_all += ls[k] * a[k];
      }
      const float inv = __builtin_amdgcn_rcpf(l_all);
      char* img = smem_raw + P_QIMG + KW_ * 32 * P_IMG_ROWB;
#pragma unroll
      for (int d2 = 0; d2 < 2; ++d2) {
        f32x4 got[3][4];
#pragma unroll
        for (int ks = 0, n = 0; ks < 4; ++ks) {
          if (ks == KW_) continue;
          const f32x4* src = reinterpret_cast<const f32x4*>(smem_raw + ks * P_RING);
          const int slot = KW_ < ks ? KW_ : KW_ - 1;       // this owner's index among sender ks's three destinations
#pragma unroll
          for (int g = 0; g < 4; ++g) got[n][g] = src[((slot * 2 + d2) * 4 + g) * 64 + lane];
          ++n;
        }
        const f32x16 own = o[QB][2 * DH + d2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = own[4 * g + e] * (a[KW_] * inv);
#pragma unroll
            for (int ks = 0, n = 0; ks < 4; ++ks) {
              if (ks == KW_) continue;
              x += got[n][g][e] * (a[ks] * inv);
              ++n;
            }
            w[e] = (bf16_t)x;
          }
          *reinterpret_cast<bf16x4*>(img + r32 * P_IMG_ROWB + (d2 * 32 + 8 * g + 4 * h) * 2) = w;
        }
      }
      // rows of the image -> global: 8 lanes x 16 bytes per row (the wave's 64 columns), 8 rows per instruction (a wave reads
      // back what it wrote itself: the compiler's own lgkmcnt wait orders the two)
      const int srow = lane >> 3, sch = lane & 7;
      bf16x8 wout[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wout[i] = *reinterpret_cast<const bf16x8*>(img + (8 * i + srow) * P_IMG_ROWB + sch * 16);
      bf16_t* op = p.O + ((long)b * p.Sq + q0 + srow) * p.ldo + (2 * hp + QB) * DK + DH * 64 + sch * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (q0 + 8 * i + srow < p.Sq) *reinterpret_cast<bf16x8*>(op) = wout[i];
        op += 8 * p.ldo;
      }
      if (DH == 0 && h == 0 && q_ok) {
        const long si = ((long)b * p.H + 2 * hp + QB) * p.Sq + q_row;
        // statistics in natural-log units; a fully masked row keeps the exact fill value (attention_fwd.h)
        p.row_max[si] = (m <= NEG_MASK * LOG2E) ? NEG_MASK : m * LN2;
        p.row_sum[si] = l_all;
      }
    }
  });
  BMHRL_STAMP(8)
}

inline bool pair128_ok(int B, int H, int Sq, int Sk) {
  return H % 2 == 0 && Sk <= P_MAX_SK && Sk >= 1 && (int64_t)B * (H / 2) * ((Sq + 31) / 32) < (1ll << 24);
}

inline hipError_t launch_pair128(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo, float* row_max,
                                 float* row_sum, const uint8_t* mask, int64_t mask_sb, int B, int H, int Sq, int Sk, float scale,
                                 int dbg, hipStream_t stream) {
  PairArgs a;
  a.Q = (const bf16_t*)Qp; a.ldq = ldq; a.X = (const bf16_t*)X; a.ldx = ldx; a.O = (bf16_t*)ctx; a.ldo = ldo;
  a.row_max = row_max; a.row_sum = row_sum; a.mask = mask; a.mask_sb = mask_sb;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale; a.dbg = dbg;
  a.q_tiles = (Sq + 31) / 32;
  a.per_b = (H / 2) * a.q_tiles;
  a.map_mode = (B % 8 == 0) ? 0 : 1;
  a.magic_perb = div_magic((unsigned)a.per_b);
  a.magic_qt = div_magic((unsigned)a.q_tiles);
  hipLaunchKernelGGL(attn_pair128_kernel, dim3((unsigned)(B * a.per_b)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace
