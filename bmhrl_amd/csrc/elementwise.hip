// HBM-bound kernels of the hot path for gfx950: LayerNorm fwd/bwd, feature add + positional encoding, embedding,
// casts with dropout, bias-gradient column sums, fusion gate, expand_goals, fused Adam.
// One wave (64 lanes) owns one row wherever a row reduction is needed; reductions are wavefront shuffles.
#include <cstdlib>

#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 256 threads = 4 waves = 4 rows in flight per block

// ------------------------------------------------------------------------------------------------ LayerNorm
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                              bf16_t* __restrict__ yb, long ldy, float* __restrict__ yf, float* __restrict__ mean,
                              float* __restrict__ rstd, long rows, int D) {
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c];
  const float mu = wave_sum(s) / D;
  float v = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] - mu;
    v += d * d;
  }
  const float rs = rsqrtf(wave_sum(v) / D + 1e-5f);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  for (int c = lane; c < D; c += 64) {
    const float y = (xr[c] - mu) * rs * gamma[c] + beta[c];
    if (yb) yb[row * ldy + c] = (bf16_t)y;
    if (yf) yf[row * D + c] = y;
  }
}

// Same, D % 4 == 0 and aligned: the row is read once with 16-byte loads and kept in registers (NV groups of 4 columns per
// lane), mean / variance come from the registers, the bf16 output leaves as 8-byte stores.
// RPW rows per wave (64 / RPW lanes each): rows of <= 128 columns (the audio stream) would otherwise leave half of every
// wave idle and need twice the waves.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {   // sum over aligned groups of LPR lanes, on every lane
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <int NV, int RPW>
__global__ void ln_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  bf16_t* __restrict__ yb, long ldy, float* __restrict__ yf, float* __restrict__ mean,
                                  float* __restrict__ rstd, long rows, int D) {
  static_assert(RPW == 1 || NV == 1, "several rows per wave only for rows that fit one 4-column group per lane");
  if (blockIdx.y) {                     // group blockIdx.y of a grouped launch: `rows` rows of its own, its own gamma / beta
    const long go = (long)blockIdx.y * rows;
    x += go * D; gamma += blockIdx.y * D; beta += blockIdx.y * D;
    if (yb) yb += go * ldy;
    if (yf) yf += go * D;
    if (mean) mean += go;
    if (rstd) rstd += go;
  }
  constexpr int LPR = 64 / RPW;
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
  const long row = ((long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW + sub;
  const bool valid = row < rows;                       // (lanes of a missing row still take part in the shuffles)
  const float* xr = x + (valid ? row : 0) * D;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * l + 4 * LPR * i;
    v[i] = (valid && c < D) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mu = group_sum<LPR>(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * l + 4 * LPR * i;
    if (c < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mu;
        q += d * d;
      }
    }
  }
  const float rs = rsqrtf(group_sum<LPR>(q) / D + 1e-5f);
  if (!valid) return;
  if (l == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * l + 4 * LPR * i;
    if (c < D) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 y;
      bf16x4 yo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (v[i][j] - mu) * rs * g[j] + b[j];
        yo[j] = (bf16_t)y[j];
      }
      if (yb) *reinterpret_cast<bf16x4*>(yb + row * ldy + c) = yo;
      if (yf) *reinterpret_cast<f32x4*>(yf + row * D + c) = y;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma.  Each wave walks `rows_per_wave` rows and keeps
// its lanes' dgamma/dbeta partial sums in registers (<= 16 columns per lane, D <= 1024), then one atomic per column.
constexpr int LN_MAXC = 16;
template <int NC>
__global__ void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                              const float* __restrict__ dx_add, float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int D,
                              int rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const long r0 = wave_id * rows_per_wave;
  float dg[NC], db[NC], xv[NC], gv[NC];   // NC = ceil(D / 64): columns per lane, row cached in registers
#pragma unroll
  for (int i = 0; i < NC; ++i) dg[i] = db[i] = 0.f;
  for (long row = r0; row < r0 + rows_per_wave && row < rows; ++row) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + row * D;
    const float* dyr = dy + row * D;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      xv[i] = gv[i] = 0.f;
      if (c < D) {
        const float d = dyr[c];
        const float xh = (xr[c] - mu) * rs, g = d * gamma[c];
        xv[i] = xh;
        gv[i] = g;
        s1 += g;
        s2 += g * xh;
        dg[i] += d * xh;
        db[i] += d;
      }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        const float v = rs * (gv[i] - s1 - xv[i] * s2);
        dx[row * D + c] = dx_add ? dx_add[row * D + c] + v : v;
      }
    }
  }
  // block-level reduction of the 4 waves' partial sums, then ONE atomic per column per block (per-row or per-wave
  // atomics onto the same D addresses serialise: 14x slower, MI355X_MICROARCH 'Global float atomics')
  __shared__ float red[ROWS_PER_BLOCK][64 * NC];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* out = pass == 0 ? dgamma : dbeta;
    if (!out) continue;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) red[w][c] = pass == 0 ? dg[i] : db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x)
      atomicAdd(out + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// Same, D % 4 == 0: every lane owns NV groups of 4 consecutive columns (16-byte loads / stores, gamma hoisted out of
// the row loop) -- 4x fewer memory instructions per row than the column-strided kernel above.
// PART: instead of the atomics, every block stores its column sums to part[block][0..D) (dgamma) and [D..2D) (dbeta);
// ln_bwd_reduce_kernel adds them up.  All blocks adding into the same 2*D addresses is the slow case of the float
// atomics (serialised at the memory side: 256 blocks x 2048 columns cost ~23 us), and it is what kept the block count
// (hence the rows in flight) low.
template <int NV, bool PART>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                  const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                  const float* __restrict__ dx_add, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  long rows, int D, int rows_per_wave, float* __restrict__ part) {
  if (blockIdx.y) {                     // grouped launch (atomic form only): see ln_fwd_vec_kernel
    const long go = (long)blockIdx.y * rows;
    dy += go * D; x += go * D; dx += go * D; gamma += blockIdx.y * D; mean += go; rstd += go;
    if (dx_add) dx_add += go * D;
    if (dgamma) dgamma += blockIdx.y * D;
    if (dbeta) dbeta += blockIdx.y * D;
  }
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const long r0 = wave_id * rows_per_wave;
  f32x4 dg[NV], db[NV], gam[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    dg[i] = db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    gam[i] = c < D ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // rows go two at a time: the loads of both are requested before either is reduced, so a wave keeps twice the bytes
  // in flight (the kernel runs at a few waves per SIMD and every row is a dependent load -> reduce -> store chain)
  struct Row { f32x4 d[NV], x[NV], a[NV]; float mu, rs; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_row = [&](const long row, Row& R) {
    R.mu = mean[row];
    R.rs = rstd[row];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * lane + 256 * i;
      const bool in = c < D;
      R.d[i] = in ? *reinterpret_cast<const f32x4*>(dy + row * D + c) : zero4;
      R.x[i] = in ? *reinterpret_cast<const f32x4*>(x + row * D + c) : zero4;
      R.a[i] = (in && dx_add) ? *reinterpret_cast<const f32x4*>(dx_add + row * D + c) : zero4;
    }
  };
  auto finish_row = [&](const long row, Row& R) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (R.x[i][j] - R.mu) * R.rs, g = R.d[i][j] * gam[i][j];   // columns >= D: d = 0, gamma = 0
        s1 += g;
        s2 += g * xh;
        dg[i][j] += R.d[i][j] * xh;
        db[i][j] += R.d[i][j];
        R.x[i][j] = xh;
        R.d[i][j] = g;
      }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * lane + 256 * i;
      if (c < D) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = R.rs * (R.d[i][j] - s1 - R.x[i][j] * s2) + R.a[i][j];
        *reinterpret_cast<f32x4*>(dx + row * D + c) = v;
      }
    }
  };
  const long rend = (r0 + rows_per_wave < rows) ? r0 + rows_per_wave : rows;
  long row = r0;
  for (; row + 1 < rend; row += 2) {
    Row A, B;
    load_row(row, A);
    load_row(row + 1, B);
    finish_row(row, A);
    finish_row(row + 1, B);
  }
  if (row < rend) {
    Row A;
    load_row(row, A);
    finish_row(row, A);
  }
  __shared__ float red[ROWS_PER_BLOCK][256 * NV];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* out = pass == 0 ? dgamma : dbeta;
    if (!out) continue;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * lane + 256 * i;
      if (c < D) *reinterpret_cast<f32x4*>(&red[w][c]) = pass == 0 ? dg[i] : db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
      const float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      if constexpr (PART) part[((long)blockIdx.x * 2 + pass) * D + c] = v;
      else atomicAdd(out + c, v);
    }
  }
}

// part (nblk, 2*D) -> dgamma / dbeta += column sums.  Block (cg, rg): 64 columns x every RG-th... the partial rows
// rg, rg + RG, ...; wave w of the block takes every 4th of those.  One atomic per column per block (RG per column).
__global__ void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int D, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;              // column of the (2*D)-wide partial rows
  float acc = 0.f;
  if (c < 2 * D)
    for (int r = blockIdx.y * 4 + w; r < nblk; r += gridDim.y * 4) acc += part[(long)r * 2 * D + c];
  __shared__ float red[4][64];
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < 2 * D) {
    const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* out = c < D ? dgamma : dbeta;
    if (out) atomicAdd(out + (c < D ? c : c - D), v);
  }
}

// ------------------------------------------------------------------------------------------------ K1: add + posenc
__global__ void add_posenc_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ pe,
                                  float* __restrict__ out, bf16_t* __restrict__ ob, long ldob, int S, int D, long total,
                                  float p, uint64_t seed0, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = seed0 + ((p > 0.f && seed_dev) ? seed_dev[0] : 0ull);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % D;
    const long row = i / D;
    const int s = row % S;
    float v = a[i] + (b ? b[i] : 0.f) + pe[(long)s * D + c];
    if (p > 0.f) v *= dropout_scale(p, seed, i);
    out[i] = v;
    if (ob) ob[row * ldob + c] = (bf16_t)v;
  }
}

// the same, four columns per thread (D % 4 == 0, 16-byte aligned operands): the video stream's 4 M elements are the first launch
// of the encoder chain -- 23.7 us one element at a time with three 64-bit divisions each.  The dropout mask is the per-element
// hash of the scalar form (same bits for the same (seed, index)).
__global__ void add_posenc_vec4_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ pe,
                                       float* __restrict__ out, bf16_t* __restrict__ ob, long ldob, int S, int D4, long total4,
                                       float p, uint64_t seed0, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = seed0 + ((p > 0.f && seed_dev) ? seed_dev[0] : 0ull);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long row = i / D4;
    const int c4 = (int)(i - row * D4), s = (int)(row % S);
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    if (b) v += reinterpret_cast<const f32x4*>(b)[i];
    v += reinterpret_cast<const f32x4*>(pe)[(long)s * D4 + c4];
    if (p > 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= dropout_scale(p, seed, (uint64_t)(4 * i + j));
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
    if (ob) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x4*>(ob + row * ldob + 4 * c4) = o;
    }
  }
}

__global__ void embed_posenc_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ tok2, float mix,
                                    const float* __restrict__ table, const float* __restrict__ pe, float* __restrict__ emb,
                                    float* __restrict__ out, int L, int D, long total, float scale, float p, uint64_t seed0,
                                    const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = seed0 + ((p > 0.f && seed_dev) ? seed_dev[0] : 0ull);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % D;
    const long row = i / D;
    const int l = row % L;
    float e = table[tok[row] * D + c] * scale;
    if (tok2) e = e * (1.f - mix) + table[tok2[row] * D + c] * scale * mix;
    if (emb) emb[i] = e;
    float v = e + pe[(long)l * D + c];
    if (p > 0.f) v *= dropout_scale(p, seed, i);
    out[i] = v;
  }
}

__global__ void embed_bwd_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ tok2, float mix,
                                 const float* __restrict__ dC, float* __restrict__ dtable, int D, long total, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % D;
    const long row = i / D;
    const float g = dC[i] * scale;
    if (tok2) {
      atomicAdd(dtable + tok[row] * D + c, g * (1.f - mix));
      atomicAdd(dtable + tok2[row] * D + c, g * mix);
    } else {
      atomicAdd(dtable + tok[row] * D + c, g);
    }
  }
}

// BMHRL_DETERMINISTIC: a thread owns a column and walks the rows in order (rows that repeat a token add in row order)
__global__ void embed_bwd_ordered_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ tok2, float mix,
                                         const float* __restrict__ dC, float* __restrict__ dtable, int D, long rows, float scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  for (long row = 0; row < rows; ++row) {
    const float g = dC[row * D + c] * scale;
    if (tok2) {
      dtable[tok[row] * D + c] += g * (1.f - mix);
      dtable[tok2[row] * D + c] += g * mix;
    } else {
      dtable[tok[row] * D + c] += g;
    }
  }
}
__global__ void scatter_add_rows_ordered_kernel(const float* __restrict__ dout, const int32_t* __restrict__ src, float* __restrict__ dx,
                                                int D, long rows) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  for (long r = 0; r < rows; ++r) {
    const int s = src[r];
    if (s >= 0) dx[(long)s * D + c] += dout[r * D + c];
  }
}

// ------------------------------------------------------------------------------------------------ casts / sums
__global__ void cast_bf16_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long rows,
                                 int cols, float scale, float p, uint64_t seed0, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = seed0 + ((p > 0.f && seed_dev) ? seed_dev[0] : 0ull);
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % cols;
    const long r = i / cols;
    float v = x[r * ldx + c] * scale;
    if (p > 0.f) v *= dropout_scale(p, seed, i);
    y[r * ldy + c] = (bf16_t)v;
  }
}

// the same cast written `copies` times, copy c at y + c * copy_stride (the paired fusion stacks address the encoder memory as 2 B
// samples: functional.StepScratch.memo_bf16)
__global__ void cast_bf16_copies_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long rows, int cols,
                                        int copies, long copy_stride) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % cols;
    const long r = i / cols;
    const bf16_t v = (bf16_t)x[r * ldx + c];
    for (int k = 0; k < copies; ++k) y[k * copy_stride + r * ldy + c] = v;
  }
}

// Split operand: x = hi + lo with hi = bf16(x), lo = bf16(x - hi).  Three column blocks of width `part` at y: block 0 = hi,
// block lo_slot (1 or 2) = lo, the remaining block = hi again.  An activation laid out [hi | hi | lo] against a weight laid out
// [hi | lo | hi] gives x_hi W_hi + x_hi W_lo + x_lo W_hi in ONE GEMM with K = 3 part: the product to ~16 mantissa bits (the
// lo x lo term, 2^-18 relative, is dropped).  Used for the vocabulary projection, whose logits carry north_star's 1e-3 bound.
__global__ void cast_split3_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long part, int lo_slot,
                                   long rows, int cols, const float* __restrict__ x2, long ldx2, int cols2) {
  // (x2: a second source whose cols2 columns follow x's -- the operand cat[x, goal completion] of the vocabulary head)
  const int ct = cols + cols2;
  const long total = rows * ct;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % ct;
    const long r = i / ct;
    const float v = c < cols ? x[r * ldx + c] : x2[r * ldx2 + c - cols];
    const bf16_t hi = (bf16_t)v;
    bf16_t* o = y + r * ldy + c;
    o[0] = hi;
    o[lo_slot * part] = (bf16_t)(v - (float)hi);
    o[(3 - lo_slot) * part] = hi;
  }
}

// Many (rows, cols) fp32 matrices -> bf16 (padded leading dimension) or fp32 copies, ONE launch: the per-step refresh
// of all bf16 weight shadows and concatenated biases.  seg = 6 int64 per segment: source address, destination address,
// rows, cols, destination leading dimension (elements; <= 0: the destination is fp32, tightly packed; bits 32.. = `part` > 0: the
// destination is a split shadow, bf16(x) at column c and c + 2 part, bf16(x - bf16(x)) at c + part), first block.
// A block owns 4096 consecutive elements of one segment and finds it by bisection over the first-block column.
constexpr int SEG_WORDS = 6, SEG_ELEMS_PER_BLOCK = 4096;
__global__ void cast_segments_kernel(const int64_t* __restrict__ seg, int n_seg) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid * SEG_WORDS + 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* e = seg + lo * SEG_WORDS;
  const float* __restrict__ src = reinterpret_cast<const float*>(e[0]);
  const long rows = e[2], cols = e[3];
  const long ldd = e[4] <= 0 ? e[4] : (e[4] & 0xffffffffL), part = e[4] <= 0 ? 0 : (e[4] >> 32);
  const long total = rows * cols;
  const long base = ((long)blockIdx.x - e[5]) * SEG_ELEMS_PER_BLOCK;
  if (ldd <= 0) {
    float* __restrict__ dst = reinterpret_cast<float*>(e[1]);
    for (long i = base + threadIdx.x; i < base + SEG_ELEMS_PER_BLOCK && i < total; i += blockDim.x) dst[i] = src[i];
    return;
  }
  bf16_t* __restrict__ dst = reinterpret_cast<bf16_t*>(e[1]);
  if (part) {                                          // split shadow [hi | lo | hi] (see bmhrl_cast_split3_bf16)
    for (long i = base + threadIdx.x; i < base + SEG_ELEMS_PER_BLOCK && i < total; i += blockDim.x) {
      const long r = i / cols, c = i - r * cols;
      const float v = src[i];
      const bf16_t hi = (bf16_t)v;
      dst[r * ldd + c] = hi;
      dst[r * ldd + part + c] = (bf16_t)(v - (float)hi);
      dst[r * ldd + 2 * part + c] = hi;
    }
    return;
  }
  if ((cols & 3) == 0 && (ldd & 3) == 0 && ((e[0] | e[1]) & 15) == 0) {      // 16-byte loads, 8-byte stores
    for (long i = base + 4 * threadIdx.x; i < base + SEG_ELEMS_PER_BLOCK && i < total; i += 4 * blockDim.x) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
      const long r = i / cols, c = i - r * cols;
      bf16x4 o;
      o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
      *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = o;
    }
  } else {
    for (long i = base + threadIdx.x; i < base + SEG_ELEMS_PER_BLOCK && i < total; i += blockDim.x) {
      const long r = i / cols, c = i - r * cols;
      dst[r * ldd + c] = (bf16_t)src[i];
    }
  }
}

// y = bf16(x * scale * dropout) AND colsum[n] += sum_m y[m][n] in one pass (dY cast + bias gradient of the same layer).
// Block = up to 256 columns x `rows_per_block` rows; thread = (4-column group, row lane).  `cgs` (power of two <= 64) column
// groups cover the block's columns and the other 256 / cgs thread rows take rows in parallel, so a narrow matrix (the
// 128-wide audio stream: cgs = 32, 8 row lanes) keeps all threads busy; one atomic per column per block.
__global__ void cast_colsum_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long rows, int cols,
                                   float scale, float p, uint64_t seed0, const uint64_t* __restrict__ seed_dev,
                                   float* __restrict__ colsum, int rows_per_block, int cgs, long group_rows, int blocks_per_group,
                                   long cs_stride) {
  // rows come in groups of `group_rows` with their own column-sum vector (colsum + group * cs_stride): two layers' dY in one
  // buffer; a block never straddles two groups.  One group = the plain form.
  __shared__ float red[1024];                       // [row lane][4 * cgs]
  const uint64_t seed = seed0 + ((p > 0.f && seed_dev) ? seed_dev[0] : 0ull);
  const int cg = threadIdx.x % cgs, ry = threadIdx.x / cgs, rl = 256 / cgs;
  const int col = blockIdx.x * 256 + cg * 4;
  const int grp = (int)blockIdx.y / blocks_per_group;
  const long r0 = grp * group_rows + (long)((int)blockIdx.y - grp * blocks_per_group) * rows_per_block;
  rows = (grp + 1) * group_rows < rows ? (grp + 1) * group_rows : rows;
  colsum += grp * cs_stride;
  const bool vec = (ldx & 3) == 0 && (ldy & 3) == 0 && col + 4 <= cols && (((uintptr_t)x & 15) | ((uintptr_t)y & 7)) == 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  long r_first = r0 + ry;
  const long r_end = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  if (col < cols && vec) {
    // four rows per pass, every load of a pass requested before its first store: a loop that loads, converts and stores row by
    // row waits for the store of the row before in front of every load (vmcnt counts stores too) -- two round trips per row
    for (; r_first + 3 * rl < r_end; r_first += 4 * rl) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(x + (r_first + u * rl) * ldx + col);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = r_first + u * rl;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float w = t[u][j] * scale;
          if (p > 0.f) w *= dropout_scale(p, seed, (uint64_t)r * cols + col + j);
          o[j] = (bf16_t)w;
          acc[j] += (float)o[j];
        }
        *reinterpret_cast<bf16x4*>(y + r * ldy + col) = o;
      }
    }
  }
  if (col < cols) {
    for (long r = r_first; r < r_end; r += rl) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * ldx + col);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else {
        for (int j = 0; j < 4 && col + j < cols; ++j) v[j] = x[r * ldx + col + j];
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float w = v[j] * scale;
        if (p > 0.f && col + j < cols) w *= dropout_scale(p, seed, (uint64_t)r * cols + col + j);
        o[j] = (bf16_t)w;
        acc[j] += (float)o[j];
      }
      if (vec) *reinterpret_cast<bf16x4*>(y + r * ldy + col) = o;
      else for (int j = 0; j < 4 && col + j < cols; ++j) y[r * ldy + col + j] = o[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[ry * (4 * cgs) + cg * 4 + j] = acc[j];
  __syncthreads();
  if ((int)threadIdx.x < 4 * cgs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < cols) {
      float t = 0.f;
      for (int k = 0; k < rl; ++k) t += red[k * (4 * cgs) + threadIdx.x];
      atomicAdd(colsum + c, t);
    }
  }
}

// db[n] (+)= sum_m dY[m][n]: a block covers 512 columns x `rows_per_block` rows; thread = (column group of 8 bf16 =
// one 16-byte load, row lane 0..3); grid.y splits the rows; one atomic per column per block.
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ dY, long ld, float* __restrict__ db, long rows, int cols,
                                   int rows_per_block, long db_stride) {
  __shared__ float red[4][512];
  dY += (long)blockIdx.z * rows * ld;            // group blockIdx.z: `rows` rows of its own, its own sums
  db += (long)blockIdx.z * db_stride;
  const int cg = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + cg * 8;
  const long r0 = (long)blockIdx.y * rows_per_block;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (col < cols) {   // ld is a multiple of 8, so the whole 8-wide group is inside the (padded) row
    for (long r = r0 + ry; r < r0 + rows_per_block && r < rows; r += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dY + r * ld + col);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ry][cg * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int gc = blockIdx.x * 512 + c;
    if (gc < cols) atomicAdd(db + gc, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// ------------------------------------------------------------------------------------------------ fusion gate
__device__ __forceinline__ float gate_of(const float* a_v) {
  const float a = fminf(fmaxf(a_v[0], -2.f), 2.f);
  return 1.f / (1.f + __expf(-a));
}
__global__ void gate_fwd_kernel(const float* __restrict__ cv, const float* __restrict__ ca, const float* __restrict__ a_v,
                                float* __restrict__ out, bf16_t* __restrict__ ob, long ldob, int D, long total) {
  const float g = gate_of(a_v);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float v = g * cv[i] + (1.f - g) * ca[i];
    out[i] = v;
    if (ob) ob[(i / D) * ldob + (i % D)] = (bf16_t)v;
  }
}
__global__ void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ cv, const float* __restrict__ ca,
                                const float* __restrict__ a_v, float* __restrict__ dcv, float* __restrict__ dca,
                                float* __restrict__ da_v, long total) {
  __shared__ float red[16];
  const float a = a_v[0];
  const float g = gate_of(a_v);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float d = dout[i];
    dcv[i] = g * d;
    dca[i] = (1.f - g) * d;
    acc += d * (cv[i] - ca[i]);
  }
  acc = block_sum(acc, red);
  // d sigmoid(clamp(a)) / da = g(1-g) inside [-2, 2], 0 outside (torch.clamp passes gradient on the closed interval)
  if (threadIdx.x == 0 && da_v) atomicAdd(da_v, (a >= -2.f && a <= 2.f) ? acc * g * (1.f - g) : 0.f);
}

// ------------------------------------------------------------------------------------------------ expand_goals
// One thread per batch row reproduces the reference's sequential visit order (model/bm_hrl_agent.py:415-429):
//  rows with labels: each segment [prev_end+1 .. end] reads the goal at `end`; the tail after the last label is
//  zeroed iff a LATER row has a label; rows without labels are untouched, except row 0, which is zeroed when any
//  later row has a label (old_b starts at 0).
__global__ void expand_goals_index_kernel(const int32_t* __restrict__ seg, int32_t* __restrict__ src, int B, int L) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  bool later = false;
  for (int bb = b + 1; bb < B && !later; ++bb)
    for (int l = 0; l < L; ++l)
      if (seg[bb * L + l] != 0) { later = true; break; }
  int prev = 0;
  bool any = false;
  for (int l = 0; l < L; ++l) {
    if (seg[b * L + l] != 0) {
      for (int j = prev; j <= l; ++j) src[b * L + j] = b * L + l;
      prev = l + 1;
      any = true;
    }
  }
  const bool zero_tail = later && (any || b == 0);
  for (int j = prev; j < L; ++j) src[b * L + j] = zero_tail ? -1 : b * L + j;
}
// expand_goals_index + gather_rows as one launch (r03): block b rebuilds the row map of sample b from the labels (the
// "a later row has a label" test is a parallel scan, the map itself one backward pass over the L positions in LDS) and
// gathers its L rows; `src` is still written (the backward scatters along it).
constexpr int EG_MAXL = 1024;
constexpr int EG_MAXD = 1024;
// EXPLORE (Manager.forward with exploration on, model/bm_hrl_agent.py:444-452): before the segment copy the reference adds ONE
// (D,) Gaussian vector to every token's goal: noise = normal(mean' , std') - 0.5 mean' with mean' = nanmean(x) / mean_factor,
// std' = sqrt(nanmean(|x - nanmean(x)|^2)) / std_factor over ALL of x (detached), i.e. noise[c] = z_c std' + 0.5 mean'.  Every
// block recomputes the two statistics over the whole (B L D) tensor itself -- 120 KB at the reference's sizes, served by L2 --
// in the same order, so all blocks add the same vector without a second launch or a grid-wide hand-off; z_c comes from the
// counter RNG of the dropout sites (seed + the device word a captured step advances), Box-Muller on two uniforms per column.
__device__ __forceinline__ float goal_noise_z(uint64_t seed, int c) {
  const float u1 = ((hash_u32(seed, 2ull * (uint64_t)c) >> 8) + 1u) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (hash_u32(seed, 2ull * (uint64_t)c + 1ull) >> 8) * (1.0f / 16777216.0f);       // [0, 1)
  return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}
template <bool EXPLORE>
__global__ __launch_bounds__(256) void expand_goals_kernel(const int32_t* __restrict__ seg, const float* __restrict__ x,
                                                          int32_t* __restrict__ src, float* __restrict__ out,
                                                          bf16_t* __restrict__ ob, long ldob, int B, int L, int D,
                                                          float mean_factor, float std_factor, uint64_t seed,
                                                          const uint64_t* __restrict__ seed_dev, float* __restrict__ noise_out) {
  __shared__ int s_src[EG_MAXL];
  __shared__ float s_noise[EXPLORE ? EG_MAXD : 1];
  __shared__ float red[16];
  const int b = blockIdx.x;
  if constexpr (EXPLORE) {
    // (16-byte loads, eight of them in flight per thread: the first form -- one dependent 4-byte load per element and pass --
    // took 50 us on the caption-side chain of the captured step for the 120 KB of the reference's goals)
    const long total = (long)B * L * D;
    const bool v4 = (total & 3) == 0 && ((uintptr_t)x & 15) == 0;
    const long n4 = v4 ? total >> 2 : 0;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
    float s = 0.f, cnt = 0.f;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 8 * 256) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long i = i0 + u * 256;
        t[u] = i < n4 ? x4[i] : f32x4{NAN, NAN, NAN, NAN};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = t[u][e];
          if (v == v) { s += v; cnt += 1.f; }
        }
    }
    for (long i = 4 * n4 + threadIdx.x; i < total; i += 256) {
      const float v = x[i];
      if (v == v) { s += v; cnt += 1.f; }
    }
    s = block_sum(s, red);
    cnt = block_sum(cnt, red);                 // (exact below 2^24 elements per thread stride; the goals are B L 64)
    const float mean = s / cnt;                // all-NaN input: NaN, as torch.nanmean
    float sq = 0.f;
    for (long i0 = threadIdx.x; i0 < n4; i0 += 8 * 256) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long i = i0 + u * 256;
        t[u] = i < n4 ? x4[i] : f32x4{NAN, NAN, NAN, NAN};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = t[u][e];
          if (v == v) sq += (v - mean) * (v - mean);
        }
    }
    for (long i = 4 * n4 + threadIdx.x; i < total; i += 256) {
      const float v = x[i];
      if (v == v) sq += (v - mean) * (v - mean);
    }
    sq = block_sum(sq, red);
    const float stdv = sqrtf(sq / cnt) / std_factor, m = mean / mean_factor;
    const uint64_t sd = seed + (seed_dev ? seed_dev[0] : 0ull);
    for (int c = threadIdx.x; c < D; c += 256) {
      const float nz = goal_noise_z(sd, c) * stdv + 0.5f * m;
      s_noise[c] = nz;
      if (noise_out && b == 0) noise_out[c] = nz;
    }
  }
  int mine = 0, after = 0;
  for (int i = threadIdx.x; i < L; i += 256) {
    s_src[i] = seg[b * L + i];
    mine |= s_src[i] != 0;
  }
  for (long i = (long)(b + 1) * L + threadIdx.x; i < (long)B * L; i += 256) after |= seg[i] != 0;
  const int any = __syncthreads_or(mine), later = __syncthreads_or(after);
  if (threadIdx.x == 0) {
    const bool zero_tail = later && (any || b == 0);
    int next = -1;
    for (int l = L - 1; l >= 0; --l) {
      if (s_src[l] != 0) next = l;
      s_src[l] = next >= 0 ? b * L + next : (zero_tail ? -1 : b * L + l);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += 256) src[b * L + i] = s_src[i];
  for (int i = threadIdx.x; i < L * D; i += 256) {
    const int r = i / D, c = i - r * D;
    const int s0 = s_src[r];
    float v = s0 >= 0 ? x[(long)s0 * D + c] : 0.f;
    if constexpr (EXPLORE) { if (s0 >= 0) v += s_noise[c]; }      // (the noise is added BEFORE the copy: zeroed tails stay zero)
    out[((long)b * L + r) * D + c] = v;
    if (ob) ob[((long)b * L + r) * ldob + c] = (bf16_t)v;
  }
}
__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ src, float* __restrict__ out,
                                   bf16_t* __restrict__ ob, long ldob, int D, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = i % D;
    const int s = src[r];
    const float v = s >= 0 ? x[(long)s * D + c] : 0.f;
    out[i] = v;
    if (ob) ob[r * ldob + c] = (bf16_t)v;
  }
}
// Backward of the row gather: dx[s] = sum of dout[r] over the rows r with src[r] == s.  The row map of expand_goals sends a
// run of CONSECUTIVE rows (a segment) to its last row and every other row to itself (or nowhere), so the owner of row s walks
// down from s while the map still points at s and adds in that fixed order -- an activation gradient (the manager's goals) that
// does not depend on the arrival order of atomics.  A general map (any row may point anywhere) would need the atomics back.
__global__ void scatter_add_rows_kernel(const float* __restrict__ dout, const int32_t* __restrict__ src,
                                        float* __restrict__ dx, int D, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long s = i / D;
    const int c = (int)(i - s * D);
    float a = 0.f;
    for (long r = s; r >= 0 && src[r] == (int32_t)s; --r) a += dout[r * D + c];
    dx[i] = a;
  }
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                            float gscale, const int32_t* __restrict__ step_dev) {
  if (step_dev) {  // step count lives on the device so a captured graph advances it between replays
    const float st = (float)step_dev[0];
    bc1 = 1.f - powf(b1, st);
    bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    // torch.optim.Adam: denom = sqrt(v)/sqrt(bias_correction2) + eps ; p -= lr/bias_correction1 * m/denom
    p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// ---- tail of a BMFusionLayer in one launch (model/bm_hrl_agent.py:107-114): out = g * LN_CV(cv) + (1 - g) * LN_CA(ca),
// g = sigmoid(clamp(a_v, -2, 2)), for up to two parameter groups (the worker and the manager stack advance together as one
// (2, B, L, D) activation: group = row / rows_per_group).  One wave per row, the row in registers (NC columns per lane).
struct TailTable {
  bmhrl_fusion_tail_params g[2];
};
template <int NC>
__global__ void fusion_tail_fwd_kernel(const float* __restrict__ ca, const float* __restrict__ cv, const TailTable T, long rows_per_group,
                                       int n_groups, int D, float* __restrict__ out, float* __restrict__ stats,
                                       bf16_t* __restrict__ out_bf16, long ldob) {
  const long rows = rows_per_group * n_groups;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bmhrl_fusion_tail_params& P = T.g[row / rows_per_group];
  const float gate = gate_of(P.a_v);
  float xa[NC], xv[NC];
  float sa = 0.f, sv = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    xa[i] = c < D ? ca[row * D + c] : 0.f;
    xv[i] = c < D ? cv[row * D + c] : 0.f;
    sa += xa[i];
    sv += xv[i];
  }
  const float ma = wave_sum(sa) / D, mv = wave_sum(sv) / D;
  float qa = 0.f, qv = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      qa += (xa[i] - ma) * (xa[i] - ma);
      qv += (xv[i] - mv) * (xv[i] - mv);
    }
  }
  const float ra = rsqrtf(wave_sum(qa) / D + 1e-5f), rv = rsqrtf(wave_sum(qv) / D + 1e-5f);
  if (lane == 0) {
    stats[row] = ma; stats[rows + row] = ra; stats[2 * rows + row] = mv; stats[3 * rows + row] = rv;
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      const float ya = (xa[i] - ma) * ra * P.gamma_ca[c] + P.beta_ca[c];
      const float yv = (xv[i] - mv) * rv * P.gamma_cv[c] + P.beta_cv[c];
      const float o = gate * yv + (1.f - gate) * ya;
      out[row * D + c] = o;
      if (out_bf16) out_bf16[row * ldob + c] = (bf16_t)o;       // (the next blocks' GEMM operand: no cast launch)
    }
  }
}

// backward of the same: dca / dcv, and (atomic +=, zeroed by the caller) d gamma / d beta of both norms and d a_v of the row's
// group.  A block walks RPB rows of ONE group (blockIdx.y); its four waves keep per-column partial sums in registers and add
// them up through LDS: one atomic per column and block.
constexpr int TAIL_RPB = 8, TAIL_MAXD = 512;    // (2 rows per wave, both in flight: 120 blocks per group at 480 rows; 4 rows one after
                                                //  the other ran 22-27 us inside the step, 64 rows per block 15 us on 16 CUs)
template <int NC>
__global__ __launch_bounds__(256) void fusion_tail_bwd_kernel(const float* __restrict__ dout0, const float* __restrict__ dout1,
                                                              long ldd0, long ldd1, const float* __restrict__ ca,
                                                              const float* __restrict__ cv, const float* __restrict__ stats,
                                                              const TailTable T, long rows_per_group, int n_groups, int D,
                                                              float* __restrict__ dca, float* __restrict__ dcv) {
  __shared__ float red[4][4][64 * NC];
  __shared__ float red_a[4];
  const long rows = rows_per_group * n_groups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = blockIdx.y;
  const bmhrl_fusion_tail_params& P = T.g[grp];
  // the incoming gradient of a group: its own base and row stride (the two stacks' outputs are consumed by different heads)
  const float* __restrict__ dout = grp ? dout1 : dout0;
  const long ldd = grp ? ldd1 : ldd0;
  const float a = P.a_v[0], gate = gate_of(P.a_v);
  float dga[NC], dba[NC], dgv[NC], dbv[NC], ga[NC], gv[NC], ba[NC], bv[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    dga[i] = dba[i] = dgv[i] = dbv[i] = 0.f;
    ga[i] = c < D ? P.gamma_ca[c] : 0.f;
    gv[i] = c < D ? P.gamma_cv[c] : 0.f;
    ba[i] = c < D ? P.beta_ca[c] : 0.f;
    bv[i] = c < D ? P.beta_cv[c] : 0.f;
  }
  float da = 0.f;
  const long r_end = min((long)(blockIdx.x + 1) * TAIL_RPB, rows_per_group);
  static_assert(TAIL_RPB == 8, "a wave owns rows w and w + 4 of the block: both rows' loads are requested before either is reduced");
  struct RowIn { float ma, ra, mv, rv, xa[NC], xv[NC], d[NC]; };
  auto load_row = [&](const long r, RowIn& R) {
    const long rg = r < r_end ? r : r_end - 1, row = grp * rows_per_group + rg;
    R.ma = stats[row]; R.ra = stats[rows + row]; R.mv = stats[2 * rows + row]; R.rv = stats[3 * rows + row];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      const bool in = c < D;
      R.xa[i] = in ? ca[row * D + c] : 0.f;
      R.xv[i] = in ? cv[row * D + c] : 0.f;
      R.d[i] = in ? dout[rg * ldd + c] : 0.f;
    }
  };
  RowIn R2[2];
  const long rw = (long)blockIdx.x * TAIL_RPB + wave;
  load_row(rw, R2[0]);
  load_row(rw + 4, R2[1]);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long r = rw + 4 * k;
    if (r >= r_end) break;
    const RowIn& R = R2[k];
    const long row = grp * rows_per_group + r;
    const float ma = R.ma, ra = R.ra, mv = R.mv, rv = R.rv;
    float ha[NC], hv[NC], d[NC];
    float s1a = 0.f, s2a = 0.f, s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      const bool in = c < D;
      ha[i] = in ? (R.xa[i] - ma) * ra : 0.f;
      hv[i] = in ? (R.xv[i] - mv) * rv : 0.f;
      d[i] = R.d[i];
      const float dya = (1.f - gate) * d[i], dyv = gate * d[i];
      da += d[i] * ((hv[i] * gv[i] + bv[i]) - (ha[i] * ga[i] + ba[i]));
      dga[i] += dya * ha[i]; dba[i] += dya;
      dgv[i] += dyv * hv[i]; dbv[i] += dyv;
      s1a += dya * ga[i]; s2a += dya * ga[i] * ha[i];
      s1v += dyv * gv[i]; s2v += dyv * gv[i] * hv[i];
    }
    s1a = wave_sum(s1a) / D; s2a = wave_sum(s2a) / D; s1v = wave_sum(s1v) / D; s2v = wave_sum(s2v) / D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        dca[row * D + c] = ra * ((1.f - gate) * d[i] * ga[i] - s1a - ha[i] * s2a);
        dcv[row * D + c] = rv * (gate * d[i] * gv[i] - s1v - hv[i] * s2v);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    red[wave][0][lane + 64 * i] = dga[i]; red[wave][1][lane + 64 * i] = dba[i];
    red[wave][2][lane + 64 * i] = dgv[i]; red[wave][3][lane + 64 * i] = dbv[i];
  }
  da = wave_sum(da);
  if (lane == 0) red_a[wave] = da;
  __syncthreads();
  for (int idx = threadIdx.x; idx < 4 * D; idx += 256) {
    const int k = idx / D, c = idx % D;
    const float v = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    float* dst = k == 0 ? P.dgamma_ca : k == 1 ? P.dbeta_ca : k == 2 ? P.dgamma_cv : P.dbeta_cv;
    if (dst) atomicAdd(dst + c, v);
  }
  // d sigmoid(clamp(a)) / da = g (1 - g) on the closed interval [-2, 2], 0 outside (as bmhrl_gate_bwd)
  if (threadIdx.x == 0 && P.da_v)
    atomicAdd(P.da_v, (a >= -2.f && a <= 2.f) ? (red_a[0] + red_a[1] + red_a[2] + red_a[3]) * gate * (1.f - gate) : 0.f);
}

// Adam over the flat bucket, parameter by parameter, writing each updated weight's bf16 shadow (or fp32 copy: concatenated
// biases) in the same pass: the per-step shadow refresh (cast_segments over every weight: 221 MB read again) disappears.
// seg = 7 int64 per parameter: offset in the flat bucket (elements), shadow address (0: none), rows, cols, shadow leading
// dimension (elements; <= 0: the shadow is fp32, tightly packed), first block, gradient address (0: the flat gradient bucket
// at the parameter's offset; else the parameter's fp32 gradient where autograd left it -- no gather copy).
// Block = 4096 consecutive elements.
constexpr int ADAM_WORDS = 7;
__global__ void adam_segments_kernel(const int64_t* __restrict__ seg, int n_seg, float* __restrict__ p, const float* __restrict__ g,
                                     float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2, float eps,
                                     float wd, float bc1, float bc2_sqrt, float gscale, const int32_t* __restrict__ step_dev) {
  if (step_dev) {
    const float st = (float)step_dev[0];
    bc1 = 1.f - powf(b1, st);
    bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid * ADAM_WORDS + 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* e = seg + lo * ADAM_WORDS;
  const long off = e[0], rows = e[2], cols = e[3];
  const long ldd = e[4] <= 0 ? e[4] : (e[4] & 0xffffffffL), part = e[4] <= 0 ? 0 : (e[4] >> 32);   // part: split shadow, as cast_segments
  g = e[6] ? reinterpret_cast<const float*>(e[6]) - off : g;        // (indexed with off + i below)
  const long total = rows * cols;
  const long base = ((long)blockIdx.x - e[5]) * SEG_ELEMS_PER_BLOCK;
  const long end = base + SEG_ELEMS_PER_BLOCK < total ? base + SEG_ELEMS_PER_BLOCK : total;
  const float step_size = lr / bc1;
  auto upd = [&](const long i) -> float {
    float gi = g[off + i] * gscale;
    const float pi = p[off + i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[off + i] + (1.f - b1) * gi;
    const float vi = b2 * v[off + i] + (1.f - b2) * gi * gi;
    m[off + i] = mi;
    v[off + i] = vi;
    const float pn = pi - step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[off + i] = pn;
    return pn;
  };
  if (e[1] == 0) {
    for (long i = base + threadIdx.x; i < end; i += blockDim.x) upd(i);
    return;
  }
  if (ldd <= 0) {
    float* __restrict__ dst = reinterpret_cast<float*>(e[1]);
    for (long i = base + threadIdx.x; i < end; i += blockDim.x) dst[i] = upd(i);
    return;
  }
  bf16_t* __restrict__ dst = reinterpret_cast<bf16_t*>(e[1]);
  if (part) {
    for (long i = base + threadIdx.x; i < end; i += blockDim.x) {
      const long r = i / cols, c = i - r * cols;
      const float pn = upd(i);
      const bf16_t hi = (bf16_t)pn;
      dst[r * ldd + c] = hi;
      dst[r * ldd + part + c] = (bf16_t)(pn - (float)hi);
      dst[r * ldd + 2 * part + c] = hi;
    }
    return;
  }
  if ((cols & 3) == 0 && (ldd & 3) == 0 && (off & 3) == 0 && (e[1] & 7) == 0 && (e[6] & 15) == 0) {     // 16-byte accesses of the four arrays
    if (end - base == SEG_ELEMS_PER_BLOCK && blockDim.x * 16 == SEG_ELEMS_PER_BLOCK) {
      // a whole block (the usual case): all sixteen 16-byte loads of a thread are requested before the first result is used --
      // the pass is a pure stream, what bounds it is the number of bytes in flight
      f32x4 gv[4], pv[4], mv[4], vv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long i = base + 4 * threadIdx.x + it * (SEG_ELEMS_PER_BLOCK / 4);
        gv[it] = *reinterpret_cast<const f32x4*>(g + off + i);
        pv[it] = *reinterpret_cast<const f32x4*>(p + off + i);
        mv[it] = *reinterpret_cast<const f32x4*>(m + off + i);
        vv[it] = *reinterpret_cast<const f32x4*>(v + off + i);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long i = base + 4 * threadIdx.x + it * (SEG_ELEMS_PER_BLOCK / 4);
        f32x4 pn;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gi = gv[it][j] * gscale;
          if (wd != 0.f) gi += wd * pv[it][j];
          mv[it][j] = b1 * mv[it][j] + (1.f - b1) * gi;
          vv[it][j] = b2 * vv[it][j] + (1.f - b2) * gi * gi;
          pn[j] = pv[it][j] - step_size * mv[it][j] / (sqrtf(vv[it][j]) / bc2_sqrt + eps);
          o[j] = (bf16_t)pn[j];
        }
        *reinterpret_cast<f32x4*>(m + off + i) = mv[it];
        *reinterpret_cast<f32x4*>(v + off + i) = vv[it];
        *reinterpret_cast<f32x4*>(p + off + i) = pn;
        const long r = i / cols, c = i - r * cols;
        *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = o;
      }
      return;
    }
    for (long i = base + 4 * threadIdx.x; i < end; i += 4 * blockDim.x) {
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + off + i);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(p + off + i);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + off + i), vv = *reinterpret_cast<const f32x4*>(v + off + i), pn;
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gi = gv[j] * gscale;
        if (wd != 0.f) gi += wd * pv[j];
        mv[j] = b1 * mv[j] + (1.f - b1) * gi;
        vv[j] = b2 * vv[j] + (1.f - b2) * gi * gi;
        pn[j] = pv[j] - step_size * mv[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
        o[j] = (bf16_t)pn[j];
      }
      *reinterpret_cast<f32x4*>(m + off + i) = mv;
      *reinterpret_cast<f32x4*>(v + off + i) = vv;
      *reinterpret_cast<f32x4*>(p + off + i) = pn;
      const long r = i / cols, c = i - r * cols;
      *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = o;
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += blockDim.x) {
      const long r = i / cols, c = i - r * cols;
      dst[r * ldd + c] = (bf16_t)upd(i);
    }
  }
}

// All masks of a step in one launch (model/masking.py:18-55 for the bimodal case): V_mask[b][t] = rgb[b][t][0] != 0,
// A_mask[b][t] = audio[b][t][0] != 0, C_mask[b][i][j] = (trg[b][j] != pad) && j <= i; each written `copies` times back to
// back (the paired fusion stacks address them as 2 B samples).
__global__ void make_masks_kernel(const float* __restrict__ rgb, long ld_rgb, const float* __restrict__ audio, long ld_aud,
                                  const int64_t* __restrict__ trg, int B, int Tv, int Ta, int L, int64_t pad, int copies,
                                  uint8_t* __restrict__ vm, uint8_t* __restrict__ am, uint8_t* __restrict__ cm) {
  const long nv = (long)B * Tv, na = (long)B * Ta, nc = (long)B * L * L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv + na + nc; i += (long)gridDim.x * blockDim.x) {
    uint8_t v;
    uint8_t* dst;
    long n, j;
    if (i < nv) { j = i; v = rgb[j * ld_rgb] != 0.f; dst = vm; n = nv; }
    else if (i < nv + na) { j = i - nv; v = audio[j * ld_aud] != 0.f; dst = am; n = na; }
    else {
      j = i - nv - na;
      const long b = j / ((long)L * L), r = j - b * L * L;
      const int qi = (int)(r / L), kj = (int)(r - (long)qi * L);
      v = (trg[b * L + kj] != pad) && kj <= qi;
      dst = cm; n = nc;
    }
    for (int c = 0; c < copies; ++c) dst[c * n + j] = v;
  }
}

// the same masks from captions[:, :-1] of a (B, L + 1) caption batch, plus the shifted copies the step works on and the
// step's device counters (one thread)
__global__ void batch_head_kernel(const float* __restrict__ rgb, long ld_rgb, const float* __restrict__ audio, long ld_aud,
                                  const int64_t* __restrict__ cap, long ld_cap, int B, int Tv, int Ta, int L, int64_t pad,
                                  int copies, uint8_t* __restrict__ vm, uint8_t* __restrict__ am, uint8_t* __restrict__ cm,
                                  int64_t* __restrict__ trg_in, int64_t* __restrict__ trg_y, int64_t* bump64, int32_t* bump32_a,
                                  int32_t* bump32_b) {
  const long nv = (long)B * Tv, na = (long)B * Ta, nc = (long)B * L * L, nt = (long)B * L;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (bump64) *bump64 += 1;
    if (bump32_a) *bump32_a += 1;
    if (bump32_b) *bump32_b += 1;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv + na + nc + nt; i += (long)gridDim.x * blockDim.x) {
    uint8_t v;
    uint8_t* dst;
    long n, j;
    if (i < nv) { j = i; v = rgb[j * ld_rgb] != 0.f; dst = vm; n = nv; }
    else if (i < nv + na) { j = i - nv; v = audio[j * ld_aud] != 0.f; dst = am; n = na; }
    else if (i < nv + na + nc) {
      j = i - nv - na;
      const long b = j / ((long)L * L), r = j - b * L * L;
      const int qi = (int)(r / L), kj = (int)(r - (long)qi * L);
      v = (cap[b * ld_cap + kj] != pad) && kj <= qi;
      dst = cm; n = nc;
    } else {
      j = i - nv - na - nc;
      const long b = j / L, l = j - b * L;
      trg_in[j] = cap[b * ld_cap + l];
      trg_y[j] = cap[b * ld_cap + l + 1];
      continue;
    }
    for (int c = 0; c < copies; ++c) dst[c * n + j] = v;
  }
}

inline unsigned grid_for(long total, int block = 256, int cap = 2048) {
  long g = (total + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" int bmhrl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, int64_t ldy,
                                   float* y_f32, float* mean, float* rstd, int64_t rows, int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32) && rows > 0 && D > 0);
  dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
  const bool vec = D % 4 == 0 && D <= 1024 && ldy % 4 == 0 &&
                   ((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_f32) & 15) == 0) && (((uintptr_t)y_bf16 & 7) == 0);
  if (vec) {
    const int nv = (D + 255) / 256;
#define LN_FWDV(NV_, RPW_, GRID_) hipLaunchKernelGGL((ln_fwd_vec_kernel<NV_, RPW_>), GRID_, block, 0, S_(stream), x, gamma, beta, \
                                                    (bf16_t*)y_bf16, (long)ldy, y_f32, mean, rstd, (long)rows, D)
    if (D <= 128) {                                      // two rows per wave
      dim3 grid2((unsigned)((rows + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK)));
      LN_FWDV(1, 2, grid2);
    } else if (nv <= 1) LN_FWDV(1, 1, grid); else if (nv <= 2) LN_FWDV(2, 1, grid); else LN_FWDV(4, 1, grid);
#undef LN_FWDV
    return hip_status(hipGetLastError());
  }
  hipLaunchKernelGGL(ln_fwd_kernel, grid, block, 0, S_(stream), x, gamma, beta, (bf16_t*)y_bf16, (long)ldy, y_f32, mean,
                     rstd, (long)rows, D);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_layernorm_bwd_ws(const float* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                                      int64_t rows, int32_t D, float* workspace, int64_t workspace_floats,
                                      bmhrl_stream_t stream);
extern "C" int64_t bmhrl_layernorm_bwd_workspace(int64_t rows, int32_t D);

// the one-launch form: parameter gradients by one atomic per column and block
static int ln_bwd_direct(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                         const float* dx_add, float* dgamma, float* dbeta, int64_t rows, int32_t D, bmhrl_stream_t stream) {
  // ~512 blocks (2048 waves): fills the chip and keeps the dgamma/dbeta atomics at ~512 per column
  int rpw = (int)((rows + 2047) / 2048);
  if (rpw < 1) rpw = 1;
  if (rpw > 32) rpw = 32;
  if (bmhrl_deterministic()) rpw = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);    // (rows no vector form serves: one block)
  const long waves = (rows + rpw - 1) / rpw;
  dim3 grid((unsigned)((waves + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(256);
  if (!bmhrl_deterministic() && D % 4 == 0 &&
      ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dx_add) & 15) == 0)) {
    // ~1024 waves (256 blocks): measured optimum on MI355X between streaming parallelism and the dgamma/dbeta atomics
    // (one per column per block, all blocks onto the same D addresses: 1024 blocks cost 2x the time of 256)
    static const int target = getenv("BMHRL_LN_WAVES") ? atoi(getenv("BMHRL_LN_WAVES")) : 1024;
    int rv = (int)((rows + target - 1) / target);
    if (rv < 1) rv = 1;
    if (rv > 32) rv = 32;
    const long wv = (rows + rv - 1) / rv;
    dim3 gridv((unsigned)((wv + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    const int nv = (D + 255) / 256;
#define LN_BWDV(NV_) hipLaunchKernelGGL((ln_bwd_vec_kernel<NV_, false>), gridv, block, 0, S_(stream), dy, x, gamma, mean, rstd, dx, \
                                        dx_add, dgamma, dbeta, (long)rows, D, rv, (float*)nullptr)
    if (nv <= 1) LN_BWDV(1); else if (nv <= 2) LN_BWDV(2); else LN_BWDV(4);
#undef LN_BWDV
    return hip_status(hipGetLastError());
  }
  const int nc = (D + 63) / 64;
#define LN_BWD(NC_) hipLaunchKernelGGL(ln_bwd_kernel<NC_>, grid, block, 0, S_(stream), dy, x, gamma, mean, rstd, dx, dx_add, \
                                       dgamma, dbeta, (long)rows, D, rpw)
  if (nc <= 1) LN_BWD(1); else if (nc <= 2) LN_BWD(2); else if (nc <= 5) LN_BWD(5); else if (nc <= 8) LN_BWD(8); else LN_BWD(16);
#undef LN_BWD
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                   const float* rstd, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                                   int64_t rows, int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows > 0 && D > 0 && D <= 64 * LN_MAXC);
  // (BMHRL_DETERMINISTIC: without a workspace the ordered form is ONE block -- correct, slow on long inputs; callers that
  //  care pass a workspace: bmhrl_layernorm_bwd_ws, as bmhrl_amd.ops.layernorm_bwd does)
  return ln_bwd_direct(dy, x, gamma, mean, rstd, dx, dx_add, dgamma, dbeta, rows, D, stream);
}

// Two-stage variant: ~4 waves per SIMD stream the rows, column sums go through `workspace`.
static void ln_bwd_ws_plan(int64_t rows, int D, int* rows_per_wave, long* blocks) {
  // waves: wide rows want few blocks (each ends with a 2*D-float reduction + store), narrow rows many rows in flight;
  // measured r01 (tests/bench_ln_bwd.py): 4096 x 1024 best at 1024 waves (16.8 us), 12800 x 128 at 4096 (10.6 us)
  static const int forced = getenv("BMHRL_LN_WS_WAVES") ? atoi(getenv("BMHRL_LN_WS_WAVES")) : 0;
  int target = forced > 0 ? forced : (1 << 20) / D;
  if (forced <= 0) target = target < 1024 ? 1024 : (target > 4096 ? 4096 : target);
  int rv = (int)((rows + target - 1) / target);
  if (rv < 1) rv = 1;
  if (rv > 32) rv = 32;
  const long wv = (rows + rv - 1) / rv;
  *rows_per_wave = rv;
  *blocks = (wv + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
}

extern "C" int64_t bmhrl_layernorm_bwd_workspace(int64_t rows, int32_t D) {
  if (rows <= 0 || D <= 0) return 0;
  int rv; long blocks;
  ln_bwd_ws_plan(rows, D, &rv, &blocks);
  return blocks * 2 * (int64_t)D;
}

extern "C" int bmhrl_layernorm_bwd_ws(const float* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                                      int64_t rows, int32_t D, float* workspace, int64_t workspace_floats,
                                      bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows > 0 && D > 0 && D <= 64 * LN_MAXC);
  const bool vec = D % 4 == 0 && D <= 1024 &&
                   ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dx_add) & 15) == 0);
  if (!vec || !workspace || (!dgamma && !dbeta) || workspace_floats < bmhrl_layernorm_bwd_workspace(rows, D))
    return ln_bwd_direct(dy, x, gamma, mean, rstd, dx, dx_add, dgamma, dbeta, rows, D, stream);
  int rv; long blocks;
  ln_bwd_ws_plan(rows, D, &rv, &blocks);
  dim3 gridv((unsigned)blocks), block(256);
  // a null dgamma / dbeta only skips the final add: the partial kernel fills both halves of the workspace rows it owns
  float* dg = dgamma ? dgamma : dbeta;
  float* db = dbeta ? dbeta : dgamma;
  const int nv = (D + 255) / 256;
  // few blocks (the caption-side rows: 480 x 300 -> 120 blocks): the same-address atomics are short then, and cheaper
  // than a second launch on the step's serial chain
  static const long atomic_max = getenv("BMHRL_LN_ATOMIC_BLOCKS") ? atol(getenv("BMHRL_LN_ATOMIC_BLOCKS")) : 128;
  if (blocks <= atomic_max && !bmhrl_deterministic()) {
#define LN_BWDA(NV_) hipLaunchKernelGGL((ln_bwd_vec_kernel<NV_, false>), gridv, block, 0, S_(stream), dy, x, gamma, mean, rstd, dx, \
                                        dx_add, dgamma, dbeta, (long)rows, D, rv, (float*)nullptr)
    if (nv <= 1) LN_BWDA(1); else if (nv <= 2) LN_BWDA(2); else LN_BWDA(4);
#undef LN_BWDA
    return hip_status(hipGetLastError());
  }
#define LN_BWDP(NV_) hipLaunchKernelGGL((ln_bwd_vec_kernel<NV_, true>), gridv, block, 0, S_(stream), dy, x, gamma, mean, rstd, dx, \
                                        dx_add, dg, db, (long)rows, D, rv, workspace)
  if (nv <= 1) LN_BWDP(1); else if (nv <= 2) LN_BWDP(2); else LN_BWDP(4);
#undef LN_BWDP
  const int cg = (2 * D + 63) / 64;
  int rg = (int)((256 + cg - 1) / cg);                         // ~256 blocks in total
  const int max_rg = (int)((blocks + 3) / 4);
  if (rg > max_rg) rg = max_rg;
  if (rg < 1 || bmhrl_deterministic()) rg = 1;           // (deterministic: one block sums a column's partials, in order)
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cg, rg), block, 0, S_(stream), workspace, (int)blocks, D, dgamma, dbeta);
  return hip_status(hipGetLastError());
}

// `groups` independent LayerNorms of rows_per_group rows each, laid out back to back (x, y, mean, rstd: group-major; gamma,
// beta, dgamma, dbeta: (groups, D)) -- the worker and the manager half of a paired fusion block -- as ONE launch
extern "C" int bmhrl_layernorm_fwd_groups(const float* x, const float* gamma, const float* beta, void* y_bf16, int64_t ldy,
                                          float* y_f32, float* mean, float* rstd, int64_t rows_per_group, int32_t D,
                                          int32_t groups, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32) && rows_per_group > 0 && D > 0 && groups >= 1 && groups <= 65535);
  const int64_t rows = rows_per_group;
  const bool vec = D % 4 == 0 && D <= 1024 && ldy % 4 == 0 && D > 128 &&
                   ((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y_f32) & 15) == 0) && (((uintptr_t)y_bf16 & 7) == 0);
  if (!vec || groups == 1) {
    for (int g = 0; g < groups; ++g) {
      const int rc = bmhrl_layernorm_fwd(x + g * rows * D, gamma + (long)g * D, beta + (long)g * D,
                                         y_bf16 ? (char*)y_bf16 + 2 * g * rows * ldy : nullptr, ldy, y_f32 ? y_f32 + g * rows * D : nullptr,
                                         mean ? mean + g * rows : nullptr, rstd ? rstd + g * rows : nullptr, rows, D, stream);
      if (rc) return rc;
    }
    return 0;
  }
  dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), (unsigned)groups), block(256);
  const int nv = (D + 255) / 256;
#define LN_FWDG(NV_) hipLaunchKernelGGL((ln_fwd_vec_kernel<NV_, 1>), grid, block, 0, S_(stream), x, gamma, beta, (bf16_t*)y_bf16, \
                                        (long)ldy, y_f32, mean, rstd, (long)rows, D)
  if (nv <= 1) LN_FWDG(1); else if (nv <= 2) LN_FWDG(2); else LN_FWDG(4);
#undef LN_FWDG
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_layernorm_bwd_groups(const float* dy, const float* x, const float* gamma, const float* mean,
                                          const float* rstd, float* dx, const float* dx_add, float* dgamma, float* dbeta,
                                          int64_t rows_per_group, int32_t D, int32_t groups, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows_per_group > 0 && D > 0 && D <= 64 * LN_MAXC && groups >= 1 &&
                  groups <= 65535);
  const int64_t rows = rows_per_group;
  int rv; long blocks;
  ln_bwd_ws_plan(rows, D, &rv, &blocks);
  static const long atomic_max = getenv("BMHRL_LN_ATOMIC_BLOCKS") ? atol(getenv("BMHRL_LN_ATOMIC_BLOCKS")) : 128;
  const bool vec = D % 4 == 0 && D <= 1024 &&
                   ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dx_add) & 15) == 0);
  if (!vec || groups == 1 || blocks > atomic_max || bmhrl_deterministic()) {
    for (int g = 0; g < groups; ++g) {
      const int rc = bmhrl_layernorm_bwd(dy + g * rows * D, x + g * rows * D, gamma + (long)g * D, mean + g * rows, rstd + g * rows,
                                         dx + g * rows * D, dx_add ? dx_add + g * rows * D : nullptr,
                                         dgamma ? dgamma + (long)g * D : nullptr, dbeta ? dbeta + (long)g * D : nullptr, rows, D, stream);
      if (rc) return rc;
    }
    return 0;
  }
  dim3 gridv((unsigned)blocks, (unsigned)groups), block(256);
  const int nv = (D + 255) / 256;
#define LN_BWDG(NV_) hipLaunchKernelGGL((ln_bwd_vec_kernel<NV_, false>), gridv, block, 0, S_(stream), dy, x, gamma, mean, rstd, dx, \
                                        dx_add, dgamma, dbeta, (long)rows, D, rv, (float*)nullptr)
  if (nv <= 1) LN_BWDG(1); else if (nv <= 2) LN_BWDG(2); else LN_BWDG(4);
#undef LN_BWDG
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_add_posenc(const float* a, const float* b, const float* pe, float* out, void* out_bf16, int64_t ldob,
                                int32_t B, int32_t S, int32_t D, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(a && pe && out && B > 0 && S > 0 && D > 0);
  const long total = (long)B * S * D;
  const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)pe | (uintptr_t)out;
  if (D % 4 == 0 && (al & 15) == 0 && (!out_bf16 || (ldob % 4 == 0 && ((uintptr_t)out_bf16 & 7) == 0))) {
    hipLaunchKernelGGL(add_posenc_vec4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, S_(stream), a, b, pe, out, (bf16_t*)out_bf16,
                       (long)ldob, S, D / 4, total / 4, dropout_p, seed, seed_dev);
    return hip_status(hipGetLastError());
  }
  hipLaunchKernelGGL(add_posenc_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), a, b, pe, out, (bf16_t*)out_bf16,
                     (long)ldob, S, D, total, dropout_p, seed, seed_dev);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_embed_posenc(const int64_t* tok, const int64_t* tok2, float mix, const float* table, const float* pe,
                                  float* emb_out, float* out, int32_t B, int32_t L, int32_t D, float scale,
                                  float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(tok && table && pe && out && B > 0 && L > 0 && D > 0);
  const long total = (long)B * L * D;
  hipLaunchKernelGGL(embed_posenc_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), tok, tok2, mix, table, pe,
                     emb_out, out, L, D, total, scale, dropout_p, seed, seed_dev);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_embed_bwd(const int64_t* tok, const int64_t* tok2, float mix, const float* dC, float* dtable,
                               int32_t B, int32_t L, int32_t D, float scale, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(tok && dC && dtable && B > 0 && L > 0 && D > 0);
  const long total = (long)B * L * D;
  if (bmhrl_deterministic())
    hipLaunchKernelGGL(embed_bwd_ordered_kernel, dim3((unsigned)((D + 63) / 64)), dim3(64), 0, S_(stream), tok, tok2, mix, dC, dtable,
                       D, (long)B * L, scale);
  else
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), tok, tok2, mix, dC, dtable, D,
                       total, scale);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_cast_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, float scale,
                               float dropout_p, uint64_t seed, const uint64_t* seed_dev, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldy >= cols && ldx >= cols);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, S_(stream), x, (long)ldx, (bf16_t*)y,
                     (long)ldy, (long)rows, cols, scale, dropout_p, seed, seed_dev);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_cast_bf16_copies(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols,
                                      int32_t copies, int64_t copy_stride, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldy >= cols && ldx >= cols && copies >= 1 && (copies == 1 || copy_stride >= rows * ldy));
  hipLaunchKernelGGL(cast_bf16_copies_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, S_(stream), x, (long)ldx, (bf16_t*)y,
                     (long)ldy, (long)rows, cols, copies, (long)copy_stride);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_cast_colsum_bf16_groups(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols,
                                             float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                             float* colsum, int64_t group_rows, int64_t colsum_stride, bmhrl_stream_t stream);

extern "C" int bmhrl_cast_split3_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part, int32_t lo_slot,
                                      int64_t rows, int32_t cols, const float* x2, int64_t ldx2, int32_t cols2,
                                      bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= 3 * part && (lo_slot == 1 || lo_slot == 2));
  if (!x2) cols2 = 0;
  BMHRL_CHECK_ARG(cols2 >= 0 && part >= cols + cols2 && (!x2 || ldx2 >= cols2));
  hipLaunchKernelGGL(cast_split3_kernel, dim3(grid_for(rows * (cols + cols2))), dim3(256), 0, S_(stream), x, (long)ldx, (bf16_t*)y,
                     (long)ldy, (long)part, lo_slot, (long)rows, cols, x2, (long)ldx2, cols2);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_cast_colsum_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols, float scale,
                                      float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* colsum,
                                      bmhrl_stream_t stream) {
  return bmhrl_cast_colsum_bf16_groups(x, ldx, y, ldy, rows, cols, scale, dropout_p, seed, seed_dev, colsum, rows, 0, stream);
}

extern "C" int bmhrl_cast_colsum_bf16_groups(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows_all, int32_t cols,
                                             float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                             float* colsum, int64_t group_rows, int64_t colsum_stride, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && y && colsum && rows_all > 0 && cols > 0 && ldy >= cols && ldx >= cols);
  BMHRL_CHECK_ARG(group_rows > 0 && rows_all % group_rows == 0);
  const int groups = (int)(rows_all / group_rows);
  const int64_t rows = group_rows;                     // the block shape is chosen per group
  const int col_blocks = (cols + 255) / 256;
  // ~512 blocks in total, but at most 128 row blocks: every block ends with one atomic per column, and atomics onto the
  // same address serialise (a 128-wide matrix cut into 512 row blocks spent 13 of its 16 us there)
  int row_blocks = 512 / col_blocks;
  row_blocks = row_blocks < 1 ? 1 : (row_blocks > 128 ? 128 : row_blocks);
  if (bmhrl_deterministic()) row_blocks = 1;             // one block per column block and group: one add per address
  int rpb = (int)((rows + row_blocks - 1) / row_blocks);
  if (rpb < 16) rpb = 16;
  const int bpg = (int)((rows + rpb - 1) / rpb);
  dim3 grid((unsigned)col_blocks, (unsigned)(bpg * groups)), block(256);
  int cgs = 64;                                         // column groups per block: the smallest power of two covering it
  while (cgs > 1 && (cgs / 2) * 4 >= (cols < 256 ? cols : 256)) cgs /= 2;
  hipLaunchKernelGGL(cast_colsum_kernel, grid, block, 0, S_(stream), x, (long)ldx, (bf16_t*)y, (long)ldy, (long)rows_all, cols,
                     scale, dropout_p, seed, seed_dev, colsum, rpb, cgs, (long)group_rows, bpg, (long)colsum_stride);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_cast_segments(const int64_t* segments, int32_t n_segments, int32_t n_blocks, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(segments && n_segments > 0 && n_blocks > 0);
  hipLaunchKernelGGL(cast_segments_kernel, dim3((unsigned)n_blocks), dim3(256), 0, S_(stream), segments, n_segments);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_colsum_bf16(const void* dY, int64_t ld, float* db, int32_t accumulate, int64_t rows, int32_t cols,
                                 bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dY && db && rows > 0 && cols > 0);
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(db, 0, sizeof(float) * cols, S_(stream));
    if (e != hipSuccess) return (int)e;
  }
  BMHRL_CHECK_ARG(ld % 8 == 0 && ((uintptr_t)dY & 15) == 0);
  const int col_blocks = (cols + 511) / 512;
  int rpb = (int)((rows * col_blocks + 511) / 512);      // aim at ~512 blocks in total
  if (rpb < 16) rpb = 16;
  if (bmhrl_deterministic()) rpb = (int)rows;            // one block per column block: one add per address
  dim3 grid((unsigned)col_blocks, (unsigned)((rows + rpb - 1) / rpb)), block(256);
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, block, 0, S_(stream), (const bf16_t*)dY, (long)ld, db, (long)rows, cols, rpb, 0l);
  return hip_status(hipGetLastError());
}

// the same for `groups` row groups laid out back to back, sums of group g ADDED at db + g * db_stride: one launch
extern "C" int bmhrl_colsum_bf16_groups(const void* dY, int64_t ld, float* db, int64_t rows_per_group, int32_t cols,
                                        int32_t groups, int64_t db_stride, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dY && db && rows_per_group > 0 && cols > 0 && groups >= 1 && groups <= 65535);
  BMHRL_CHECK_ARG(ld % 8 == 0 && ((uintptr_t)dY & 15) == 0);
  const int col_blocks = (cols + 511) / 512;
  int rpb = (int)((rows_per_group * groups * col_blocks + 511) / 512);
  if (rpb < 16) rpb = 16;
  if (bmhrl_deterministic()) rpb = (int)rows_per_group;
  dim3 grid((unsigned)col_blocks, (unsigned)((rows_per_group + rpb - 1) / rpb), (unsigned)groups), block(256);
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, block, 0, S_(stream), (const bf16_t*)dY, (long)ld, db, (long)rows_per_group, cols, rpb,
                     (long)db_stride);
  return hip_status(hipGetLastError());
}

static int tail_table(const bmhrl_fusion_tail_params* groups, int32_t n_groups, bool bwd, TailTable& T) {
  BMHRL_CHECK_ARG(groups && (n_groups == 1 || n_groups == 2));
  for (int i = 0; i < n_groups; ++i) {
    const bmhrl_fusion_tail_params& g = groups[i];
    BMHRL_CHECK_ARG(g.gamma_ca && g.beta_ca && g.gamma_cv && g.beta_cv && g.a_v);
    (void)bwd;
    T.g[i] = g;
  }
  if (n_groups == 1) T.g[1] = T.g[0];
  return 0;
}

extern "C" int bmhrl_fusion_tail_fwd(const float* ca, const float* cv, const bmhrl_fusion_tail_params* groups, int32_t n_groups,
                                     int64_t rows_per_group, int32_t D, float* out, float* stats, void* out_bf16, int64_t ldob,
                                     bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(ca && cv && out && stats && rows_per_group > 0 && D > 0 && D <= TAIL_MAXD);
  TailTable T;
  if (int rc = tail_table(groups, n_groups, false, T)) return rc;
  const long rows = rows_per_group * n_groups;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  BMHRL_CHECK_ARG(!out_bf16 || ldob >= D);
  if (D <= 320) hipLaunchKernelGGL(fusion_tail_fwd_kernel<5>, grid, block, 0, S_(stream), ca, cv, T, (long)rows_per_group, n_groups, D, out, stats, (bf16_t*)out_bf16, (long)ldob);
  else hipLaunchKernelGGL(fusion_tail_fwd_kernel<8>, grid, block, 0, S_(stream), ca, cv, T, (long)rows_per_group, n_groups, D, out, stats, (bf16_t*)out_bf16, (long)ldob);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_fusion_tail_bwd(const float* dout, const float* dout1, int64_t ldd0, int64_t ldd1, const float* ca,
                                     const float* cv, const float* stats, const bmhrl_fusion_tail_params* groups, int32_t n_groups,
                                     int64_t rows_per_group, int32_t D, float* dca, float* dcv, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dout && ca && cv && stats && dca && dcv && rows_per_group > 0 && D > 0 && D <= TAIL_MAXD);
  TailTable T;
  if (int rc = tail_table(groups, n_groups, true, T)) return rc;
  if (ldd0 <= 0) ldd0 = D;
  if (ldd1 <= 0) ldd1 = ldd0;
  if (!dout1) dout1 = dout + rows_per_group * ldd0;          // (one tensor holding both groups back to back)
  BMHRL_CHECK_ARG(ldd0 >= D && ldd1 >= D);
  dim3 grid((unsigned)((rows_per_group + TAIL_RPB - 1) / TAIL_RPB), (unsigned)n_groups), block(256);
  if (D <= 320) hipLaunchKernelGGL(fusion_tail_bwd_kernel<5>, grid, block, 0, S_(stream), dout, dout1, (long)ldd0, (long)ldd1, ca, cv, stats, T, (long)rows_per_group, n_groups, D, dca, dcv);
  else hipLaunchKernelGGL(fusion_tail_bwd_kernel<8>, grid, block, 0, S_(stream), dout, dout1, (long)ldd0, (long)ldd1, ca, cv, stats, T, (long)rows_per_group, n_groups, D, dca, dcv);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_gate_fwd(const float* cv, const float* ca, const float* a_v, float* out, void* out_bf16, int64_t ldob,
                              int64_t rows, int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(cv && ca && a_v && out && rows > 0 && D > 0);
  const long total = rows * D;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), cv, ca, a_v, out, (bf16_t*)out_bf16,
                     (long)ldob, D, total);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_gate_bwd(const float* dout, const float* cv, const float* ca, const float* a_v, float* dcv, float* dca,
                              float* da_v, int64_t rows, int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dout && cv && ca && a_v && dcv && dca && rows > 0 && D > 0);
  const long total = rows * D;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(bmhrl_deterministic() ? 1u : grid_for(total, 256, 256)), dim3(256), 0, S_(stream), dout,
                     cv, ca, a_v, dcv, dca, da_v, total);       // (deterministic: one block = one add into da_v)
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_expand_goals_index(const int32_t* seg, int32_t* src, int32_t B, int32_t L, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(seg && src && B > 0 && L > 0);
  hipLaunchKernelGGL(expand_goals_index_kernel, dim3((B + 63) / 64), dim3(64), 0, S_(stream), seg, src, B, L);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_expand_goals(const int32_t* seg, const float* x, int32_t* src, float* out, void* out_bf16, int64_t ldob,
                                  int32_t B, int32_t L, int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(seg && x && src && out && B > 0 && L > 0 && L <= EG_MAXL && D > 0 && (!out_bf16 || ldob >= D));
  hipLaunchKernelGGL(expand_goals_kernel<false>, dim3((unsigned)B), dim3(256), 0, S_(stream), seg, x, src, out,
                     (bf16_t*)out_bf16, (long)ldob, B, L, D, 1.f, 1.f, 0ull, (const uint64_t*)nullptr, (float*)nullptr);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_expand_goals_explore(const int32_t* seg, const float* x, int32_t* src, float* out, void* out_bf16,
                                          int64_t ldob, int32_t B, int32_t L, int32_t D, float mean_factor, float std_factor,
                                          uint64_t seed, const uint64_t* seed_dev, float* noise_out, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(seg && x && src && out && B > 0 && L > 0 && L <= EG_MAXL && D > 0 && D <= EG_MAXD && (!out_bf16 || ldob >= D));
  BMHRL_CHECK_ARG(mean_factor != 0.f && std_factor != 0.f && (int64_t)B * L * D < (1ll << 24));   // float element counts
  hipLaunchKernelGGL(expand_goals_kernel<true>, dim3((unsigned)B), dim3(256), 0, S_(stream), seg, x, src, out,
                     (bf16_t*)out_bf16, (long)ldob, B, L, D, mean_factor, std_factor, (uint64_t)seed, seed_dev, noise_out);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_gather_rows(const float* x, const int32_t* src, float* out, void* out_bf16, int64_t ldob, int64_t rows,
                                 int32_t D, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && src && out && rows > 0 && D > 0);
  const long total = rows * D;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), x, src, out, (bf16_t*)out_bf16,
                     (long)ldob, D, total);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_scatter_add_rows(const float* dout, const int32_t* src, float* dx, int64_t rows, int32_t D,
                                      bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dout && src && dx && rows > 0 && D > 0);
  const long total = rows * D;
  if (bmhrl_deterministic())
    hipLaunchKernelGGL(scatter_add_rows_ordered_kernel, dim3((unsigned)((D + 63) / 64)), dim3(64), 0, S_(stream), dout, src, dx, D,
                       (long)rows);
  else
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), dout, src, dx, D, total);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int32_t step, const int32_t* step_dev,
                               float grad_scale, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || step_dev));
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, S_(stream), param, grad, exp_avg, exp_avg_sq,
                     (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, step_dev);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_adam_segments(const int64_t* segments, int32_t n_segments, int32_t n_blocks, float* param, const float* grad,
                                   float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int32_t step, const int32_t* step_dev, float grad_scale,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(segments && n_segments > 0 && n_blocks > 0 && param && grad && exp_avg && exp_avg_sq && (step >= 1 || step_dev));
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_segments_kernel, dim3((unsigned)n_blocks), dim3(256), 0, S_(stream), segments, n_segments, param, grad,
                     exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, step_dev);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_make_masks(const float* rgb, int64_t ld_rgb, const float* audio, int64_t ld_aud, const int64_t* trg, int32_t B,
                                int32_t Tv, int32_t Ta, int32_t L, int64_t pad_idx, int32_t copies, uint8_t* v_mask,
                                uint8_t* a_mask, uint8_t* c_mask, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(rgb && audio && trg && v_mask && a_mask && c_mask && B > 0 && Tv > 0 && Ta > 0 && L > 0 && copies >= 1);
  const long total = (long)B * (Tv + Ta + (long)L * L);
  hipLaunchKernelGGL(make_masks_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), rgb, (long)ld_rgb, audio, (long)ld_aud, trg,
                     B, Tv, Ta, L, pad_idx, copies, v_mask, a_mask, c_mask);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_batch_head(const float* rgb, int64_t ld_rgb, const float* audio, int64_t ld_aud, const int64_t* captions,
                                int64_t ld_cap, int32_t B, int32_t Tv, int32_t Ta, int32_t L, int64_t pad_idx, int32_t copies,
                                uint8_t* v_mask, uint8_t* a_mask, uint8_t* c_mask, int64_t* trg_in, int64_t* trg_y,
                                int64_t* bump64, int32_t* bump32_a, int32_t* bump32_b, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(rgb && audio && captions && v_mask && a_mask && c_mask && trg_in && trg_y);
  BMHRL_CHECK_ARG(B > 0 && Tv > 0 && Ta > 0 && L > 0 && copies >= 1 && ld_cap >= L + 1);
  const long total = (long)B * (Tv + Ta + (long)L * L + L);
  hipLaunchKernelGGL(batch_head_kernel, dim3(grid_for(total)), dim3(256), 0, S_(stream), rgb, (long)ld_rgb, audio, (long)ld_aud,
                     captions, (long)ld_cap, B, Tv, Ta, L, pad_idx, copies, v_mask, a_mask, c_mask, trg_in, trg_y, bump64, bump32_a,
                     bump32_b);
  return hip_status(hipGetLastError());
}

extern "C" const char* bmhrl_hip_arch(void) { return "gfx950"; }
// 2: bmhrl_gemm_desc.colsum, bmhrl_cast_colsum_bf16, bmhrl_cast_segments
// 3: bmhrl_layernorm_bwd_ws (+ _workspace), bmhrl_rnn_wavefront / bmhrl_rnn_layer; attention outputs 16-byte aligned, ldo % 8 == 0
// 7: bmhrl_adam_segments   8: bmhrl_gemm_desc.colsum_sb1 / bias_sb1, fp16 attention entry points   9: bmhrl_make_masks
// 10: bmhrl_softmax_bwd_rows, bmhrl_cast_split3_bf16 (+ split shadows in the segment tables), DSCORE honours the mask,
//     bmhrl_sample_tokens row_offset, bmhrl_attention_max_keys, bmhrl_token_loss_reduce, bmhrl_fusion_tail_fwd / _bwd
// 11: bmhrl_batch_head, bmhrl_smooth_kl_bwd loss_scale2, bmhrl_layernorm_fwd_groups / _bwd_groups, bmhrl_colsum_bf16_groups,
//     bmhrl_cast_bf16_copies
extern "C" int bmhrl_hip_abi_version(void) { return 17; }
// the ONE reading of BMHRL_DETERMINISTIC (common.h): the host side asks here instead of parsing the variable again
extern "C" int bmhrl_deterministic_enabled(void) { return bmhrl_deterministic() ? 1 : 0; }
