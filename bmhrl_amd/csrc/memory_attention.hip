// Few-query attention over a long memory, core of the fusion layers' caption -> audio / video attentions in the absorbed
// form (model/bm_hrl_agent.py:87-101 through functional.PairMemAttnFn: Q' = Q Wk per head, keys = values = the memory rows):
// L <= 32 caption positions of a (sample, head) against Sk <= 896 memory rows of width dm (128 audio / 1024 video).
//
// On the batched-GEMM path the core is score GEMM + row softmax + context GEMM forward and dP GEMM + row kernel + dQ' GEMM
// backward, six launches per block whose 30-row outputs waste most of every tile.  Here a workgroup owns one (sample, head)
// and runs ONE template in two modes:
//
//   pass 1   T^T[key][q] = mem[key] . X[q]      X = Q' (forward) or dCx (backward), staged in LDS; the wave's key tiles
//            (tile t to wave t mod 8) stay in registers, the key rows fed in the order that makes the accumulator layout the
//            B-operand layout of pass 2 (as in small_attention.hip)
//   middle   forward : scale, -1e9 at masked keys, softmax over ALL keys (register partials -> one LDS exchange between the
//                      eight waves), P -> bf16 -> HBM (the backward and d(mem) need it) and an LDS image
//            backward: dS = scale P (dP - sum_k P dP) with P read back in the same layout, 0 at masked keys, -> HBM and LDS
//   pass 2   Y^T[d][q] = memT[d] . E[q]         E = P / dS rows from the LDS image, memT = the memory transposed once per step
//            (functional.StepScratch.memo_mem) so that the reduction over keys is contiguous for the MFMA A operand; the d
//            tiles are dealt to the waves, the rows leave through an LDS image as whole 16-byte pieces
//
// The memory is read straight from L2 into the A operand (each (sample, head) streams it once per pass: 2 x 512 KB for the video
// side).  MEASURED alone (tests/bench_memattn.py): r03 video side 44 us forward / 46 us backward against 35 us for the three
// launches it replaces, audio side 28 / 31 against 33; r04 with the XCD-aware block map below (all heads and both stacks of a
// sample on one XCD: the memory leaves the Infinity Cache once instead of four times) 38.8 / 44.2 us and 23.0 / 24.6 us, the
// GEMM path 29.9 / 33.2 us.  Deeper unrolling of the load streams (32 loads in flight per wave) changed nothing.  In the captured
// step the two paths are indistinguishable (4.907 vs 4.906 - 4.918 ms) and this one is 16 launches shorter, so it is the
// default since r04 (functional.FUSED_MEMATTN; BMHRL_FUSED_MEMATTN=0 selects the GEMM path; parity:
// tests/test_memory_attention_gpu.py, tests/test_blocks_gpu.py).
#include "../../include/bmhrl_hip.h"
#include "common.h"

namespace {

// eight waves: the memory comes straight from L2 into the A operand, so the bytes in flight per CU are what the waves hold in
// registers -- four waves with eight 16-byte loads each in flight streamed 16 GB/s per CU (61 us for the video side)
constexpr int MA_WAVES = 8, MA_THREADS = 64 * MA_WAVES;
constexpr int MA_MAXD = 1024, MA_MAXK = 896, MA_PAD = 8, MA_NT = (MA_MAXK / 32 + MA_WAVES - 1) / MA_WAVES;   // <= 4 key tiles per wave

struct MemAttnArgs {
  const bf16_t* X; long ldx;            // row (b2 * L + q) * ldx + h * dm
  const bf16_t* mem; long mem_sb;       // + (b2 % nb) * mem_sb + key * dm
  const bf16_t* memT; long memT_sb; int ldt;   // + (b2 % nb) * memT_sb + d * ldt + key   (columns >= Sk zero up to a multiple of 16)
  bf16_t* PD; long pd_row; long pd_slot;       // + (b2 * L + q) * pd_row + slot * pd_slot + h * Skp + key
  bf16_t* Y; long ldy;                  // row (b2 * L + q) * ldy + h * dm
  const uint8_t* mask; long mask_sb;    // + b2 * mask_sb + key
  int nb, H, L, Sk, Skp, dm;
  float scale;
  int xcd_map;                          // 1: block -> (sample, head) through the XCD-aware map (B2 a multiple of 8)
};

__device__ __forceinline__ int ma_key_of_row(int rho) {
  return 8 * ((rho >> 2) & 1) + (rho & 3) + 4 * ((rho >> 3) & 1) + 16 * ((rho >> 4) & 1);
}

template <bool BWD>
__global__ __launch_bounds__(MA_THREADS) void mem_attn_kernel(const MemAttnArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Xs[32 * (MA_MAXD + MA_PAD)];   // X rows; later the output image
  __shared__ __attribute__((aligned(16))) bf16_t Es[32 * (MA_MAXK + MA_PAD)];   // P / dS rows [q][key]
  __shared__ float red[MA_WAVES][32][2];
  // workgroups are dealt round-robin over the 8 XCDs (private L2s).  All heads of a sample -- and the same sample of the other
  // fusion stack (b2 % nb) -- stream the SAME memory: give them block ids that differ by multiples of 8 so that one L2 fetches
  // it once (with the plain order the four heads of a sample sit on four XCDs: the memory came out of the Infinity Cache 4 x)
  int b2, hd;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    hd = idx % p.H;
    b2 = xcd + 8 * (idx / p.H);
  } else {
    b2 = blockIdx.x / p.H;
    hd = blockIdx.x % p.H;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int dm = p.dm, ldxs = dm + MA_PAD, ldes = p.Skp + MA_PAD;
  const int nt = (p.Sk + 31) / 32;
  const bf16_t* __restrict__ memb = p.mem + (long)(b2 % p.nb) * p.mem_sb;
  const bf16_t* __restrict__ memt = p.memT + (long)(b2 % p.nb) * p.memT_sb;

  // ---- X rows -> LDS (rows >= L zero), E image cleared (its padding keys must read as zero in pass 2)
  {
    const int cpr = dm / 8;
    for (int c = threadIdx.x; c < 32 * cpr; c += MA_THREADS) {
      const int row = c / cpr, ch = c % cpr;
      bf16x8 v = zero_bf16x8();
      if (row < p.L) v = *reinterpret_cast<const bf16x8*>(p.X + ((long)b2 * p.L + row) * p.ldx + hd * dm + ch * 8);
      *reinterpret_cast<bf16x8*>(Xs + row * ldxs + ch * 8) = v;
    }
  }
  __syncthreads();

  // ---- pass 1: this wave's key tiles
  f32x16 acc[MA_NT];
  const bf16_t* xb = Xs + r32 * ldxs + 8 * hh;
#pragma unroll
  for (int i = 0; i < MA_NT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int t = wave + MA_WAVES * i;
    if (t < nt) {
      const int key = min(t * 32 + ma_key_of_row(r32), p.Sk - 1);        // (rows past the memory: clamped, masked below)
      const bf16_t* ka = memb + (long)key * dm + 8 * hh;
#pragma unroll 16
      for (int ks = 0; ks < dm / 16; ++ks)
        acc[i] = BMHRL_MFMA16(*reinterpret_cast<const bf16x8*>(ka + 16 * ks), *reinterpret_cast<const bf16x8*>(xb + 16 * ks), acc[i], 0, 0, 0);
    }
  }
  // this lane: query r32; register r of tile i: key 32 (wave + MA_WAVES i) + 8 hh + (r & 7) + 16 (r >> 3)
  const int q = r32;
  const uint8_t* mrow = p.mask ? p.mask + (long)b2 * p.mask_sb : nullptr;
  bf16_t* pd = p.PD + ((long)b2 * p.L + (q < p.L ? q : 0)) * p.pd_row + hd * p.Skp;
  if constexpr (!BWD) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MA_NT; ++i) {
      const int k0 = 32 * (wave + MA_WAVES * i) + 8 * hh;
      if (wave + MA_WAVES * i < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + (r & 7) + 16 * (r >> 3);
          float s = acc[i][r] * p.scale;
          if (key >= p.Sk) s = -INFINITY;
          else if (mrow && !mrow[key]) s = NEG_MASK;
          acc[i][r] = s;
          m = fmaxf(m, s);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (hh == 0) red[wave][q][0] = m;
    __syncthreads();
    m = red[0][q][0];
#pragma unroll
    for (int w = 1; w < MA_WAVES; ++w) m = fmaxf(m, red[w][q][0]);
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < MA_NT; ++i)
      if (wave + MA_WAVES * i < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][r] = __expf(acc[i][r] - m);
          l += acc[i][r];
        }
      }
    l += __shfl_xor(l, 32, 64);
    if (hh == 0) red[wave][q][1] = l;
    __syncthreads();
    float lsum = 0.f;
#pragma unroll
    for (int w = 0; w < MA_WAVES; ++w) lsum += red[w][q][1];
    const float inv = 1.f / lsum;
#pragma unroll
    for (int i = 0; i < MA_NT; ++i)
      if (wave + MA_WAVES * i < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] *= inv;
      }
  } else {
    // P in the layout of the accumulators, delta = sum_k P dP over all keys, dS
    float delta = 0.f;
    bf16x8 pf[MA_NT][2];
#pragma unroll
    for (int i = 0; i < MA_NT; ++i) {
      pf[i][0] = pf[i][1] = zero_bf16x8();
      const int k0 = 32 * (wave + MA_WAVES * i) + 8 * hh;
      if (wave + MA_WAVES * i < nt && q < p.L) {
        if (k0 < p.Skp) pf[i][0] = *reinterpret_cast<const bf16x8*>(pd + k0);
        if (k0 + 16 < p.Skp) pf[i][1] = *reinterpret_cast<const bf16x8*>(pd + k0 + 16);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) delta = fmaf((float)pf[i][r >> 3][r & 7], acc[i][r], delta);
    }
    delta += __shfl_xor(delta, 32, 64);
    if (hh == 0) red[wave][q][0] = delta;
    __syncthreads();
    delta = 0.f;
#pragma unroll
    for (int w = 0; w < MA_WAVES; ++w) delta += red[w][q][0];
#pragma unroll
    for (int i = 0; i < MA_NT; ++i) {
      const int k0 = 32 * (wave + MA_WAVES * i) + 8 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + (r & 7) + 16 * (r >> 3);
        float v = p.scale * (float)pf[i][r >> 3][r & 7] * (acc[i][r] - delta);
        if (key >= p.Sk || (mrow && !mrow[min(key, p.Sk - 1)])) v = 0.f;      // masked_fill: no gradient reaches a masked score
        acc[i][r] = v;
      }
    }
  }
  // E (P or dS) -> bf16: HBM (slot of the interleaved buffer) and the LDS image of pass 2
  bf16_t* eg = pd + (BWD ? p.pd_slot : 0);
#pragma unroll
  for (int i = 0; i < MA_NT; ++i) {
    const int k0 = 32 * (wave + MA_WAVES * i) + 8 * hh;
    if (wave + MA_WAVES * i < nt) {
      bf16x8 e0, e1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e0[j] = (bf16_t)acc[i][j];
        e1[j] = (bf16_t)acc[i][8 + j];
      }
      if (k0 < p.Skp) *reinterpret_cast<bf16x8*>(Es + q * ldes + k0) = e0;
      if (k0 + 16 < p.Skp) *reinterpret_cast<bf16x8*>(Es + q * ldes + k0 + 16) = e1;
      if (q < p.L) {
        if (k0 < p.Skp) *reinterpret_cast<bf16x8*>(eg + k0) = e0;
        if (k0 + 16 < p.Skp) *reinterpret_cast<bf16x8*>(eg + k0 + 16) = e1;
      }
    }
  }
  // keys between Skp and the next multiple of 16 (pass 2 walks whole 16-key steps): zero in the image
  {
    const int kend = (p.Sk + 15) & ~15;
    for (int c = threadIdx.x; c < 32 * (kend - p.Skp); c += MA_THREADS) Es[(c / (kend - p.Skp)) * ldes + p.Skp + c % (kend - p.Skp)] = (bf16_t)0.f;
  }
  __syncthreads();                                  // E image complete; every wave is done with Xs

  // ---- pass 2: Y^T[d][q] = sum_key memT[d][key] E[q][key], the d-tiles dealt to the waves
  const int nks = (p.Sk + 15) / 16;
  const bf16_t* eb = Es + r32 * ldes + 8 * hh;
  for (int dt = wave; dt < dm / 32; dt += MA_WAVES) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const bf16_t* ta = memt + (long)(dt * 32 + r32) * p.ldt + 8 * hh;
#pragma unroll 16
    for (int ks = 0; ks < nks; ++ks)
      o = BMHRL_MFMA16(*reinterpret_cast<const bf16x8*>(ta + 16 * ks), *reinterpret_cast<const bf16x8*>(eb + 16 * ks), o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Xs[q * ldxs + dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh] = (bf16_t)o[r];
  }
  __syncthreads();
  {
    const int cpr = dm / 8;
    for (int c = threadIdx.x; c < 32 * cpr; c += MA_THREADS) {
      const int row = c / cpr, ch = c % cpr;
      if (row < p.L)
        *reinterpret_cast<bf16x8*>(p.Y + ((long)b2 * p.L + row) * p.ldy + hd * dm + ch * 8) = *reinterpret_cast<const bf16x8*>(Xs + row * ldxs + ch * 8);
    }
  }
}

// fp32 (B * Sk, dm) -> bf16 row-major copy (B * Sk, dm) AND the per-sample transposed copy (B, dm, ldt), one pass
__global__ void cast_memory_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ yt, int Sk, int dm, int ldt) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, k0 = blockIdx.y * 32, d0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int key = k0 + r;
    float v = 0.f;
    if (key < Sk) {
      v = x[((long)b * Sk + key) * dm + d0 + tx];
      y[((long)b * Sk + key) * dm + d0 + tx] = (bf16_t)v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int key = k0 + tx;
    if (key < ldt) yt[((long)b * dm + d0 + r) * ldt + key] = (bf16_t)tile[tx][r];      // (keys in [Sk, ldt): zero)
  }
}

bool ma_aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int bmhrl_memory_attention_ok(int32_t L, int32_t Sk, int32_t dm) {
  return L >= 1 && L <= 32 && Sk >= 1 && Sk <= MA_MAXK && dm >= 32 && dm <= MA_MAXD && dm % 32 == 0;
}

extern "C" int bmhrl_cast_memory(const float* x, void* y, void* y_t, int32_t B, int32_t Sk, int32_t dm, int32_t ldt,
                                 bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && y && y_t && B > 0 && Sk > 0 && dm > 0 && dm % 32 == 0 && ldt >= Sk && ldt % 8 == 0 && B <= 65535);
  dim3 grid((unsigned)(dm / 32), (unsigned)((ldt + 31) / 32), (unsigned)B), block(256);
  hipLaunchKernelGGL(cast_memory_kernel, grid, block, 0, (hipStream_t)stream, x, (bf16_t*)y, (bf16_t*)y_t, Sk, dm, ldt);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_memory_attention(int32_t backward, const void* X, int64_t ldx, const void* mem, int64_t mem_sb,
                                      const void* mem_t, int64_t mem_t_sb, int32_t ldt, void* PD, int64_t pd_row,
                                      int64_t pd_slot, void* Y, int64_t ldy, const uint8_t* mask, int64_t mask_sb, int32_t n_mem,
                                      int32_t B2, int32_t H, int32_t L, int32_t Sk, int32_t dm, float scale,
                                      bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(X && mem && mem_t && PD && Y && B2 > 0 && H > 0 && n_mem > 0 && bmhrl_memory_attention_ok(L, Sk, dm));
  const int Skp = (Sk + 7) & ~7;
  BMHRL_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && pd_row % 8 == 0 && pd_slot % 8 == 0 && ldt % 8 == 0 && ldt >= ((Sk + 15) & ~15));
  BMHRL_CHECK_ARG(mem_sb % 8 == 0 && mem_t_sb % 8 == 0 && pd_row >= (int64_t)H * Skp && (long)B2 * H < (1l << 31));
  BMHRL_CHECK_ARG(ma_aligned16(X) && ma_aligned16(mem) && ma_aligned16(mem_t) && ma_aligned16(PD) && ma_aligned16(Y));
  MemAttnArgs a;
  a.X = (const bf16_t*)X; a.ldx = ldx; a.mem = (const bf16_t*)mem; a.mem_sb = mem_sb;
  a.memT = (const bf16_t*)mem_t; a.memT_sb = mem_t_sb; a.ldt = ldt;
  a.PD = (bf16_t*)PD; a.pd_row = pd_row; a.pd_slot = pd_slot; a.Y = (bf16_t*)Y; a.ldy = ldy;
  a.mask = mask; a.mask_sb = mask_sb; a.nb = n_mem; a.H = H; a.L = L; a.Sk = Sk; a.Skp = Skp; a.dm = dm; a.scale = scale;
  static const bool plain = getenv("BMHRL_MEMATTN_PLAINMAP") != nullptr;      // (A/B switch)
  a.xcd_map = (B2 % 8 == 0 && !plain) ? 1 : 0;
  dim3 grid((unsigned)(B2 * H)), block(MA_THREADS);
  if (backward) hipLaunchKernelGGL(mem_attn_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mem_attn_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
  return hip_status(hipGetLastError());
}
