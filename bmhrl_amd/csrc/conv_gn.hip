// Conv1d(padding='same') + GroupNorm input projection of the reference's DETR-mode agent (model/det_bmhrl_agent.py:79-86,
// 169-174: three Conv1d(d_model, d_model, kernel 3 / 6 / 9) + GroupNorm(32) blocks over the time axis of the video
// features) for gfx950.  The convolution is a GEMM over an unfolded operand -- row (b, t) of the operand holds the k time
// steps t - left .. t - left + k - 1 of sample b side by side (zeros outside the clip; 'same' padding puts the extra step of an
// even kernel on the right: left = (k - 1) / 2) --, so its forward, weight gradient and data gradient run on bmhrl_gemm; the
// kernels here build the operand (fp32 -> bf16), fold the operand's gradient back onto the time axis, and do the GroupNorm.
// Layout: activations (B, T, C) row major (the reference transposes to (B, C, T) for its Conv1d / GroupNorm modules and back).
#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

#define S_(s) ((hipStream_t)(s))

__global__ __launch_bounds__(256) void unfold1d_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long ldo, int B, int T,
                                                       int C, int k, int left) {
  // one thread per 4 columns of the operand row (b, t): columns j * C + c
  const long groups = (long)B * T * k * (C / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4)) * 4;
    const long r = i / (C / 4);
    const int j = (int)(r % k);
    const long bt = r / k;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int ts = t + j - left;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ts >= 0 && ts < T) v = *reinterpret_cast<const f32x4*>(x + ((long)b * T + ts) * C + c4);
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(out + bt * ldo + (long)j * C + c4) = o;
  }
}

// dx[b][t][c] = sum_j dU[(b, t - j + left)][j * C + c] over the rows that exist: every output element is owned by one thread
// (no atomics), the k terms are added in order of j
__global__ __launch_bounds__(256) void fold1d_kernel(const float* __restrict__ du, long ldu, float* __restrict__ dx, int B, int T, int C,
                                                     int k, int left) {
  const long groups = (long)B * T * (C / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4)) * 4;
    const long bt = i / (C / 4);
    const int t = (int)(bt % T), b = (int)(bt / T);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
      const int tr = t - j + left;
      if (tr >= 0 && tr < T) a += *reinterpret_cast<const f32x4*>(du + ((long)b * T + tr) * ldu + (long)j * C + c4);
    }
    *reinterpret_cast<f32x4*>(dx + bt * C + c4) = a;
  }
}

// GroupNorm over (T, C / G) per (sample, group): block = (b, g); two passes over the group (mean, then centred variance: the
// group is T * C / G elements, a few KB .. a few hundred KB: L2-resident on the second pass), then the affine output
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int T, int C,
                                                            int G, float eps) {
  __shared__ float red[16];
  const int g = blockIdx.x % G, b = blockIdx.x / G;
  const int cpg = C / G;
  const long n = (long)T * cpg;
  const float* xb = x + (long)b * T * C + g * cpg;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += xb[(i / cpg) * C + i % cpg];
  const float mean = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float d = xb[(i / cpg) * C + i % cpg] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum(q, red) / (float)n + eps);
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  float* yb = y + (long)b * T * C + g * cpg;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cpg);
    const long o = (i / cpg) * C + c;
    yb[o] = (xb[o] - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c];
  }
}

// backward: with xh = (x - mean) rstd and dxh = dy gamma over the group (n elements):
//   dx = rstd (dxh - mean(dxh) - xh mean(dxh xh)) ;  dgamma[c] += sum_t dy xh ;  dbeta[c] += sum_t dy   (one atomic per channel and block)
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int T, int C, int G) {
  __shared__ float red[16];
  __shared__ float s_dg[256], s_db[256];
  const int g = blockIdx.x % G, b = blockIdx.x / G;
  const int cpg = C / G;
  const long n = (long)T * cpg;
  const long base = (long)b * T * C + g * cpg;
  const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
  float s1 = 0.f, s2 = 0.f;
  // a thread always visits the same channel when 256 % cpg == 0 (cpg = 32 at the reference's width): its dgamma / dbeta terms
  // stay in registers; otherwise (odd widths) they go through the per-channel atomics element by element
  const bool fixed = 256 % cpg == 0;
  float dg = 0.f, db = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cpg);
    const long o = base + (i / cpg) * C + c;
    const float xh = (x[o] - mean) * rstd, d = dy[o], dxh = d * gamma[g * cpg + c];
    s1 += dxh;
    s2 += dxh * xh;
    if (fixed) { dg += d * xh; db += d; }
    else {
      if (dgamma) atomicAdd(dgamma + g * cpg + c, d * xh);
      if (dbeta) atomicAdd(dbeta + g * cpg + c, d);
    }
  }
  s1 = block_sum(s1, red) / (float)n;
  s2 = block_sum(s2, red) / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = (int)(i % cpg);
    const long o = base + (i / cpg) * C + c;
    const float xh = (x[o] - mean) * rstd, dxh = dy[o] * gamma[g * cpg + c];
    dx[o] = rstd * (dxh - s1 - xh * s2);
  }
  if (fixed) {
    s_dg[threadIdx.x] = dg;
    s_db[threadIdx.x] = db;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
      float a = 0.f, c2 = 0.f;
      for (int k = threadIdx.x; k < 256; k += cpg) { a += s_dg[k]; c2 += s_db[k]; }
      if (dgamma) atomicAdd(dgamma + g * cpg + threadIdx.x, a);
      if (dbeta) atomicAdd(dbeta + g * cpg + threadIdx.x, c2);
    }
  }
}

inline unsigned grid_for(long items) {
  long b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int bmhrl_unfold1d_bf16(const float* x, void* out, int64_t ldo, int32_t B, int32_t T, int32_t C, int32_t k, int32_t left,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && out && B > 0 && T > 0 && C > 0 && C % 4 == 0 && k >= 1 && left >= 0 && left < k && ldo >= (int64_t)k * C && ldo % 4 == 0);
  BMHRL_CHECK_ARG((((uintptr_t)x & 15) | ((uintptr_t)out & 7)) == 0);
  hipLaunchKernelGGL(unfold1d_kernel, dim3(grid_for((long)B * T * k * (C / 4))), dim3(256), 0, S_(stream), x, (bf16_t*)out, (long)ldo, B, T,
                     C, k, left);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_fold1d(const float* du, int64_t ldu, float* dx, int32_t B, int32_t T, int32_t C, int32_t k, int32_t left,
                            bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(du && dx && B > 0 && T > 0 && C > 0 && C % 4 == 0 && k >= 1 && left >= 0 && left < k && ldu >= (int64_t)k * C && ldu % 4 == 0);
  BMHRL_CHECK_ARG((((uintptr_t)du | (uintptr_t)dx) & 15) == 0);
  hipLaunchKernelGGL(fold1d_kernel, dim3(grid_for((long)B * T * (C / 4))), dim3(256), 0, S_(stream), du, (long)ldu, dx, B, T, C, k, left);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int32_t B, int32_t T, int32_t C, int32_t G, float eps, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && gamma && beta && y && mean && rstd && B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0);
  hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3((unsigned)(B * G)), dim3(256), 0, S_(stream), x, gamma, beta, y, mean, rstd, T, C, G, eps);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                                   float* dgamma, float* dbeta, int32_t B, int32_t T, int32_t C, int32_t G, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dy && x && gamma && mean && rstd && dx && B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0 && C / G <= 256);
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3((unsigned)(B * G)), dim3(256), 0, S_(stream), dy, x, gamma, mean, rstd, dx, dgamma, dbeta,
                     T, C, G);
  return hip_status(hipGetLastError());
}
