// Per-token loss step for gfx950: log-softmax over the vocabulary, label-smoothing KL / biased KL (forward row
// sums and the gradient w.r.t. the logits, including the path through the reward amplitude), categorical sampling
// and the REINFORCE-with-baseline terms.  All HBM-bound: one 256-thread block per (b, l) row of V log-probs; the
// smoothed target distribution is never materialised (closed form over the <= 3 special columns of a row).
//
// Reference: model/bm_hrl_agent.py:463-466 (log-softmax), loss/label_smoothing.py:12-32, loss/biased_kl.py:22-53,69-81,
// epoch_loops/captioning_bmrl_loops.py:271-334,409-416.
#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

__global__ void log_softmax_kernel(float* __restrict__ x, long ld, int V) {
  __shared__ float red[16];
  float* r = x + (long)blockIdx.x * ld;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < V; c += blockDim.x) m = fmaxf(m, r[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s += __expf(r[c] - m);
  s = block_sum(s, red);
  const float lse = m + __logf(s);
  for (int c = threadIdx.x; c < V; c += blockDim.x) r[c] -= lse;
}

// Rows of up to 256 * 4 * NV columns with 16-byte alignment: the row is read ONCE with 16-byte loads and stays in registers
// (the vocabulary row of 10 172 logits: 10 float4 per thread) -- one read and one write instead of three strided passes.
template <int NV>
__global__ __launch_bounds__(256) void log_softmax_vec_kernel(float* __restrict__ x, long ld, int V) {
  __shared__ float red[16];
  float* r = x + (long)blockIdx.x * ld;
  f32x4 v[NV];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    v[i] = c < V ? *reinterpret_cast<const f32x4*>(r + c) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
  }
  m = block_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __expf(v[i][j] - m);          // exp(-inf) = 0 for the columns past V
  s = block_sum(s, red);
  const float lse = m + __logf(s);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    if (c < V) *reinterpret_cast<f32x4*>(r + c) = v[i] - f32x4{lse, lse, lse, lse};
  }
}

template <int NV>
__global__ __launch_bounds__(256) void log_softmax_bwd_vec_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                                  long ld, bf16_t* __restrict__ gb, long ldg, int V) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* d = dlogp + row * ld;
  const float* lp = logp + row * ld;
  f32x4 dv[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    dv[i] = c < V ? *reinterpret_cast<const f32x4*>(d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (dv[i][0] + dv[i][1]) + (dv[i][2] + dv[i][3]);
  }
  s = block_sum(s, red);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    if (c < V) {
      const f32x4 l = *reinterpret_cast<const f32x4*>(lp + c);
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(dv[i][j] - __expf(l[j]) * s);
      *reinterpret_cast<bf16x4*>(gb + row * ldg + c) = o;
    }
  }
}

// The target distribution of one row, as the reference builds it:
//   dist[v] = u ; dist[t] = keep*(1-amp) ; dist[pad] = 0 ; dist[a] += keep*amp ; (pad row -> 0) ; (+1e-8 if biased)
struct RowTarget {
  int t, a;          // target token, sampled token (-1 = plain label smoothing)
  bool zero_row;     // pad row that the reference zeroes
  float u, keep, amp, eps;
  int pad;
  __device__ float at(int c) const {  // dist (+eps) at column c
    if (zero_row) return eps;
    float v = (c == pad) ? 0.f : (c == t ? keep * (1.f - amp) : u);
    if (c == a) v += keep * amp;
    return v + eps;
  }
};

__device__ __forceinline__ float xlogx(float d) { return d > 0.f ? d * __logf(d) : 0.f; }

// pad rows are zeroed only if the SUM of padded flat indices is positive: row r > 0 always qualifies itself; row 0
// qualifies only if some other row is padded too.
__device__ bool pad_row_zeroed(const int64_t* trg, long row, long rows, int pad, int zero_pad_rows) {
  if (trg[row] != pad) return false;
  if (zero_pad_rows >= 0) return zero_pad_rows != 0;
  if (row > 0) return true;
  for (long r = 1; r < rows; ++r)
    if (trg[r] == pad) return true;
  return false;
}

__device__ RowTarget make_target(const float* lp, const int64_t* trg, const int64_t* btrg, const float* score,
                                 const float* n_row, float smoothing, int pad, int zero_pad_rows, long row, long rows,
                                 int V, float* raw_amp) {
  RowTarget T;
  T.t = (int)trg[row];
  T.a = btrg ? (int)btrg[row] : -1;
  T.u = smoothing / (V - 2);
  T.keep = 1.f - smoothing;
  T.pad = pad;
  T.eps = btrg ? 1e-8f : 0.f;
  T.zero_row = pad_row_zeroed(trg, row, rows, pad, zero_pad_rows);
  T.amp = 0.f;
  if (btrg) {
    const float raw = score[row] * __expf(lp[T.a]) * n_row[row];
    if (raw_amp) *raw_amp = raw;
    T.amp = fminf(fmaxf(raw, 0.f), 1.f);
  }
  return T;
}

__global__ void smooth_kl_fwd_kernel(const float* __restrict__ logp, long ld, const int64_t* __restrict__ trg,
                                     const int64_t* __restrict__ btrg, const float* __restrict__ score,
                                     const float* __restrict__ n_row, float smoothing, int pad, int zero_pad_rows,
                                     float* __restrict__ row_loss, float* __restrict__ amp_out, long rows, int V) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* lp = logp + row * ld;
  float s1 = 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s1 += lp[c];
  s1 = block_sum(s1, red);
  if (threadIdx.x != 0) return;
  const RowTarget T = make_target(lp, trg, btrg, score, n_row, smoothing, pad, zero_pad_rows, row, rows, V, nullptr);
  if (amp_out) amp_out[row] = T.amp;
  // special columns: unique({t, pad, a}); every other column holds d0
  int sp[3], ns = 0;
  sp[ns++] = T.t;
  if (T.pad != T.t) sp[ns++] = T.pad;
  if (T.a >= 0 && T.a != T.t && T.a != T.pad) sp[ns++] = T.a;
  const float d0 = T.zero_row ? T.eps : T.u + T.eps;
  float loss = 0.f, lp_special = 0.f;
  for (int i = 0; i < ns; ++i) {
    const float d = T.at(sp[i]);
    loss += xlogx(d) - d * lp[sp[i]];
    lp_special += lp[sp[i]];
  }
  loss += (float)(V - ns) * xlogx(d0) - d0 * (s1 - lp_special);
  row_loss[row] = loss;
}

// The unreduced (rows, V) divergence itself -- what the reference's criteria return (loss/label_smoothing.py:32,
// loss/biased_kl.py:52) -- for callers that look at single entries (analyze_bmhrl_div); the training path only ever sums it.
__global__ void smooth_kl_full_kernel(const float* __restrict__ logp, long ld, const int64_t* __restrict__ trg,
                                      const int64_t* __restrict__ btrg, const float* __restrict__ score,
                                      const float* __restrict__ n_row, float smoothing, int pad, int zero_pad_rows,
                                      float* __restrict__ out, long rows, int V) {
  const long row = blockIdx.x;
  const float* lp = logp + row * ld;
  __shared__ RowTarget sT;
  if (threadIdx.x == 0) sT = make_target(lp, trg, btrg, score, n_row, smoothing, pad, zero_pad_rows, row, rows, V, nullptr);
  __syncthreads();
  const RowTarget T = sT;
  for (int c = threadIdx.x; c < V; c += blockDim.x) {
    const float d = T.at(c);
    out[row * V + c] = xlogx(d) - d * lp[c];
  }
}

__global__ void smooth_kl_bwd_kernel(const float* __restrict__ logp, long ld, const int64_t* __restrict__ trg,
                                     const int64_t* __restrict__ btrg, const float* __restrict__ score,
                                     const float* __restrict__ n_row, float smoothing, int pad, int zero_pad_rows,
                                     const float* __restrict__ loss_scale, const float* __restrict__ loss_scale2,
                                     int wrt_logits, bf16_t* __restrict__ gb, long ldg, float* __restrict__ gf, long rows, int V) {
  const long row = blockIdx.x;
  const float* lp = logp + row * ld;
  float raw = 0.f;
  const RowTarget T = make_target(lp, trg, btrg, score, n_row, smoothing, pad, zero_pad_rows, row, rows, V, &raw);
  const float scale = loss_scale2 ? loss_scale[0] * loss_scale2[0] : loss_scale[0];
  // d rowloss / d logp_v = -dist'_v, plus the amplitude path on column a:
  //   d rowloss/d amp = keep * [ (log d'_a + 1 - logp_a) - (t != pad) * (log d'_t + 1 - logp_t) ],  d amp/d logp_a = raw
  float extra_a = 0.f;
  if (T.a >= 0 && !T.zero_row && raw >= 0.f && raw <= 1.f) {
    const float da = T.at(T.a);
    float g = __logf(da) + 1.f - lp[T.a];
    if (T.t != T.pad) {
      const float dt = T.at(T.t);
      g -= __logf(dt) + 1.f - lp[T.t];
    }
    extra_a = T.keep * g * raw;
  }
  int ns = 1 + (T.pad != T.t) + (T.a >= 0 && T.a != T.t && T.a != T.pad);
  const float d0 = T.zero_row ? T.eps : T.u + T.eps;
  float dsum = (float)(V - ns) * d0 + T.at(T.t);
  if (T.pad != T.t) dsum += T.at(T.pad);
  if (T.a >= 0 && T.a != T.t && T.a != T.pad) dsum += T.at(T.a);
  const float G = wrt_logits ? (-dsum + extra_a) : 0.f;  // sum_v d rowloss / d logp_v (log-softmax backward term)
  for (int c = threadIdx.x; c < V; c += blockDim.x) {
    float g = -T.at(c) - (wrt_logits ? __expf(lp[c]) * G : 0.f);
    if (c == T.a) g += extra_a;
    g *= scale;
    if (gb) gb[row * ldg + c] = (bf16_t)g;
    if (gf) gf[row * (long)V + c] = g;
  }
}

// d rowloss / d log(raw amplitude) of every row (0 where the clamp is active, on pad rows the reference zeroes, without a sampled
// token): what the backward adds to column a of the row itself.  The manager branch of biased_kl() needs it per row: its
// amplitude holds the PRODUCT of the sampled tokens' probabilities over a segment, so the same quantity also flows to the
// other tokens of the segment (epoch_loops/captioning_bmrl_loops.py:299-317).
__global__ void smooth_kl_amp_grad_kernel(const float* __restrict__ logp, long ld, const int64_t* __restrict__ trg,
                                          const int64_t* __restrict__ btrg, const float* __restrict__ score,
                                          const float* __restrict__ n_row, float smoothing, int pad, int zero_pad_rows,
                                          float* __restrict__ out, long rows, int V) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float* lp = logp + row * ld;
  float raw = 0.f;
  const RowTarget T = make_target(lp, trg, btrg, score, n_row, smoothing, pad, zero_pad_rows, row, rows, V, &raw);
  float e = 0.f;
  if (T.a >= 0 && !T.zero_row && raw >= 0.f && raw <= 1.f) {
    float g = __logf(T.at(T.a)) + 1.f - lp[T.a];
    if (T.t != T.pad) g -= __logf(T.at(T.t)) + 1.f - lp[T.t];
    e = T.keep * g * raw;
  }
  out[row] = e;
}

__global__ void log_softmax_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp, long ld,
                                       bf16_t* __restrict__ gb, long ldg, int V) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* d = dlogp + row * ld;
  const float* lp = logp + row * ld;
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s += d[c];
  s = block_sum(s, red);
  for (int c = threadIdx.x; c < V; c += blockDim.x) gb[row * ldg + c] = (bf16_t)(d[c] - __expf(lp[c]) * s);
}

// The warmstart step's whole loss tail in ONE pass over the logits (r03): log-softmax in place (the row stays in registers,
// as in log_softmax_vec_kernel), the label-smoothing row sum (smooth_kl_fwd_kernel), the gradient w.r.t. the logits as the
// bf16 operand of the head's backward (smooth_kl_bwd_kernel with wrt_logits: same expressions, same values) and the loops'
// reduction loss = weight * sum(rows) / n_tokens: every block counts the tokens itself (rows is a few hundred) and adds its row
// to a fixed-point sum; the block that arrives last writes the loss and re-arms the four sync words (deterministic).
// `dloss` is the gradient the caller will hand to backward() (a constant 1 in the trainer): the gradient is final here.
template <int NV>
__global__ __launch_bounds__(256) void head_loss_kernel(float* __restrict__ x, long ld, const int64_t* __restrict__ trg,
                                                        float smoothing, int pad, const float* __restrict__ weight, float factor,
                                                        const float* __restrict__ dloss, float* __restrict__ row_loss,
                                                        float* __restrict__ out, bf16_t* __restrict__ gb, long ldg,
                                                        unsigned* __restrict__ sync, long rows, int V) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  float* r = x + row * ld;
  f32x4 v[NV];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    v[i] = c < V ? *reinterpret_cast<const f32x4*>(r + c) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
  }
  m = block_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __expf(v[i][j] - m);
  s = block_sum(s, red);
  const float lse = m + __logf(s);
  const RowTarget T = make_target(r, trg, nullptr, nullptr, nullptr, smoothing, pad, -1, row, rows, V, nullptr);
  float s1 = 0.f, lp_t = 0.f, lp_pad = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    if (c < V) {
      v[i] = v[i] - f32x4{lse, lse, lse, lse};
      *reinterpret_cast<f32x4*>(r + c) = v[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1 += v[i][j];
        if (c + j == T.t) lp_t = v[i][j];
        if (c + j == T.pad) lp_pad = v[i][j];
      }
    }
  }
  float cnt = 0.f;
  for (long i = threadIdx.x; i < rows; i += 256) cnt += trg[i] != pad ? 1.f : 0.f;
  // one exchange for the four sums (lp_t / lp_pad: one thread holds the value, the others 0)
  s1 = wave_sum(s1); lp_t = wave_sum(lp_t); lp_pad = wave_sum(lp_pad); cnt = wave_sum(cnt);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const int wv = threadIdx.x >> 6;
    red[wv] = s1; red[4 + wv] = lp_t; red[8 + wv] = lp_pad; red[12 + wv] = cnt;
  }
  __syncthreads();
  s1 = (red[0] + red[1]) + (red[2] + red[3]);
  lp_t = (red[4] + red[5]) + (red[6] + red[7]);
  lp_pad = (red[8] + red[9]) + (red[10] + red[11]);
  cnt = (red[12] + red[13]) + (red[14] + red[15]);
  const float w = (weight ? weight[0] : 1.f) / (cnt * factor);
  const float scale = dloss ? w * dloss[0] : w;
  // gradient: the expressions of smooth_kl_bwd_kernel without a sampled token
  const int ns = 1 + (T.pad != T.t);
  const float d0 = T.zero_row ? T.eps : T.u + T.eps;
  float dsum = (float)(V - ns) * d0 + T.at(T.t);
  if (T.pad != T.t) dsum += T.at(T.pad);
  const float G = -dsum;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (threadIdx.x + 256 * i);
    if (c < V) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g = -T.at(c + j) - __expf(v[i][j]) * G;
        g *= scale;
        o[j] = (bf16_t)g;
      }
      *reinterpret_cast<bf16x4*>(gb + row * ldg + c) = o;
    }
  }
  if (threadIdx.x != 0) return;
  // row sum: the closed form of smooth_kl_fwd_kernel
  float loss = xlogx(T.at(T.t)) - T.at(T.t) * lp_t, lp_special = lp_t;
  if (T.pad != T.t) {
    loss += xlogx(T.at(T.pad)) - T.at(T.pad) * lp_pad;
    lp_special += lp_pad;
  }
  loss += (float)(V - ns) * xlogx(d0) - d0 * (s1 - lp_special);
  row_loss[row] = loss;
  // sum over the rows without a second launch and without a fence (a release fence writes this XCD's L2 back: 480 of them
  // cost 16 us): the rows are added as 2^-32 fixed point with a 64-bit integer atomic -- associative, so the sum does not
  // depend on the order the blocks arrive in -- and the block whose counter increment comes last reads the total.  An
  // atomic that has RETURNED has been performed where all XCDs see it, so a block's increment follows its own addition.
  // A row loss that is not finite (a NaN / Inf logit: training has diverged) has no fixed-point image -- the conversion of a
  // NaN or an out-of-range double to an integer is undefined and would leave a finite garbage sum -- so it is reported through
  // the fourth sync word instead (bit 0 NaN, bit 1 +Inf or beyond the fixed-point range, bit 2 -Inf ...), which the last block
  // turns into what the sum of the float rows would have been: the loss the caller monitors shows the divergence.
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(sync);
  constexpr float FIXED_MAX = 1048576.f;                       // 2^20 per row: 2^52 in fixed point, room for 2^11 such rows
  unsigned bad = 0u;
  if (loss != loss) bad = 1u;
  else if (loss > FIXED_MAX) bad = 2u;
  else if (loss < -FIXED_MAX) bad = 4u;
  if (bad) {
    const unsigned f0 = atomicOr(sync + 3, bad);               // (returned: performed before this block's arrival below)
    asm volatile("" : : "v"(f0) : "memory");
    loss = 0.f;
  }
  const long long fixed = (long long)((double)loss * 4294967296.0);
  const unsigned long long before = atomicAdd(acc, (unsigned long long)fixed);
  asm volatile("" : : "v"((unsigned)before) : "memory");
  if (atomicAdd(sync + 2, 1u) != (unsigned)(rows - 1)) return;
  const long long total = (long long)atomicAdd(acc, 0ull);
  const unsigned flags = atomicExch(sync + 3, 0u);
  float sum = (float)((double)total * (1.0 / 4294967296.0));
  if (flags & 1u) sum = NAN;
  else if ((flags & 6u) == 6u) sum = NAN;                      // +Inf + -Inf
  else if (flags & 2u) sum = INFINITY;
  else if (flags & 4u) sum = -INFINITY;
  out[0] = sum * w;
  out[1] = w;
  atomicExch(acc, 0ull);
  atomicExch(sync + 2, 0u);
}

// Inverse-CDF categorical sample with one uniform per row (or arg-max), block per row.
// loss = weight * sum(row_loss) / n_tokens and scale = weight / n_tokens (what the backward multiplies every row by), with
// n_tokens = #(trg != pad): the reduction epoch_loops/captioning_bmrl_loops.py:1156-1158 (warmstart) / :846-847,859 (RL, with
// `factor` = 4/20) does with torch.sum / n_tokens.  One block; rows = B*L is a few hundred.
__global__ void token_loss_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ trg, long rows, int64_t pad,
                                         const float* __restrict__ weight, float factor, float* __restrict__ loss,
                                         float* __restrict__ scale) {
  __shared__ float s_sum[256];
  __shared__ float s_cnt[256];
  float a = 0.f, c = 0.f;
  for (long i = threadIdx.x; i < rows; i += blockDim.x) {
    a += row_loss[i];
    c += trg[i] != pad ? 1.f : 0.f;
  }
  s_sum[threadIdx.x] = a;
  s_cnt[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float w = (weight ? weight[0] : 1.f) / (s_cnt[0] * factor);
    loss[0] = s_sum[0] * w;
    scale[0] = w;
  }
}

__global__ void sample_kernel(const float* __restrict__ logp, long ld, int64_t* __restrict__ out, float* __restrict__ p_out,
                              int V, int greedy, uint64_t seed, const uint64_t* __restrict__ seed_dev, long row_offset) {
  if (seed_dev) seed += seed_dev[0];          // a device word advanced per step: fresh samples under graph replay
  __shared__ float part[256];
  __shared__ int best_i[256];
  const long row = blockIdx.x;
  const float* lp = logp + row * ld;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int chunk = (V + nt - 1) / nt;
  const int c0 = tid * chunk, c1 = min(V, c0 + chunk);
  if (greedy) {
    float bv = -INFINITY;
    int bi = V;
    for (int c = c0; c < c1; ++c)
      if (lp[c] > bv) { bv = lp[c]; bi = c; }
    part[tid] = bv;
    best_i[tid] = bi;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < nt; ++i)
        if (part[i] > bv) { bv = part[i]; bi = best_i[i]; }  // first maximum wins, like torch.argmax
      out[row] = bi;
      if (p_out) p_out[row] = __expf(lp[bi]);
    }
    return;
  }
  float s = 0.f;
  for (int c = c0; c < c1; ++c) s += __expf(lp[c]);
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    for (int i = 0; i < nt; ++i) total += part[i];
    const float u = uniform01(seed, (uint64_t)(row + row_offset)) * total;   // (row_offset: this rank's first row of the global batch)
    float acc = 0.f;
    int t = 0;
    while (t < nt - 1 && acc + part[t] <= u) acc += part[t++];
    int c = t * chunk;
    const int ce = min(V, c + chunk);
    int pick = ce - 1;
    for (; c < ce; ++c) {
      acc += __expf(lp[c]);
      if (acc > u) { pick = c; break; }
    }
    if (pick < 0) pick = 0;
    out[row] = pick;
    if (p_out) p_out[row] = __expf(lp[pick]);
  }
}

__global__ void reinforce_kernel(const float* __restrict__ pred, long ld, int is_logp, const int64_t* __restrict__ action,
                                 const float* __restrict__ value, const float* __restrict__ critic_value,
                                 float* __restrict__ row_policy, float* __restrict__ row_value, long rows) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float raw = pred[row * ld + action[row]];
  const float pa = fminf(fmaxf(is_logp ? __expf(raw) : raw, 1e-5f), 1.f - 1e-5f);
  const float adv = value[row] - critic_value[row];
  row_policy[row] = -adv * __logf(pa);
  row_value[row] = adv * adv;
}

// d(mean(row_policy) + mean(row_value)) * gscale: the probability gradient has one non-zero per row (the clamp passes
// gradient on the closed interval, like torch.clamp); value / critic gradients are +-2 adv / rows.
__global__ void reinforce_bwd_kernel(const float* __restrict__ probs, long ld, const int64_t* __restrict__ action,
                                     const float* __restrict__ value, const float* __restrict__ critic_value,
                                     const float* __restrict__ gscale, float* __restrict__ dprobs,
                                     float* __restrict__ dvalue, float* __restrict__ dcritic, long rows, int V) {
  const long row = blockIdx.x;
  const float g = gscale[0] / (float)rows;
  const float adv = value[row] - critic_value[row];
  const int a = (int)action[row];
  const float pa = probs[row * ld + a];
  const float ga = (pa >= 1e-5f && pa <= 1.f - 1e-5f) ? -adv / pa * g : 0.f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) dprobs[row * (long)V + c] = (c == a) ? ga : 0.f;
  if (threadIdx.x == 0) {
    if (dvalue) dvalue[row] = 2.f * adv * g;
    if (dcritic) dcritic[row] = -2.f * adv * g;
  }
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" int bmhrl_log_softmax(float* logits, int64_t ld, int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logits && rows > 0 && V > 0 && ld >= V);
  if (V % 4 == 0 && ld % 4 == 0 && ((uintptr_t)logits & 15) == 0 && V <= 256 * 4 * 12) {
    const int nv = (V + 1023) / 1024;
#define LSM(NV_) hipLaunchKernelGGL(log_softmax_vec_kernel<NV_>, dim3((unsigned)rows), dim3(256), 0, S_(stream), logits, (long)ld, V)
    if (nv <= 1) LSM(1); else if (nv <= 2) LSM(2); else if (nv <= 4) LSM(4); else if (nv <= 8) LSM(8); else LSM(12);
#undef LSM
    return hip_status(hipGetLastError());
  }
  hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), logits, (long)ld, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_smooth_kl_fwd(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg,
                                   const float* score, const float* n_row, float smoothing, int32_t pad_idx,
                                   int32_t zero_pad_rows, float* row_loss, float* amp_out, int64_t rows, int32_t V,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logp && trg && row_loss && rows > 0 && V > 2);
  BMHRL_CHECK_ARG(!biased_trg || (score && n_row));
  hipLaunchKernelGGL(smooth_kl_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), logp, (long)ld, trg, biased_trg,
                     score, n_row, smoothing, pad_idx, zero_pad_rows, row_loss, amp_out, (long)rows, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_smooth_kl_full(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg, const float* score,
                                    const float* n_row, float smoothing, int32_t pad_idx, int32_t zero_pad_rows, float* out,
                                    int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logp && trg && out && rows > 0 && V > 2);
  BMHRL_CHECK_ARG(!biased_trg || (score && n_row));
  hipLaunchKernelGGL(smooth_kl_full_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), logp, (long)ld, trg, biased_trg, score,
                     n_row, smoothing, pad_idx, zero_pad_rows, out, (long)rows, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_token_loss_reduce(const float* row_loss, const int64_t* trg, int64_t rows, int64_t pad_idx, const float* weight,
                                      float factor, float* loss, float* scale, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(row_loss && trg && loss && scale && rows > 0 && factor > 0.f);
  hipLaunchKernelGGL(token_loss_reduce_kernel, dim3(1), dim3(256), 0, S_(stream), row_loss, trg, (long)rows, pad_idx, weight, factor,
                     loss, scale);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_head_loss(float* logits, int64_t ld, const int64_t* trg, float smoothing, int32_t pad_idx, const float* weight,
                               float factor, const float* dloss, float* row_loss, float* loss_scale, void* dlogits_bf16,
                               int64_t ldg, uint32_t* counter, int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logits && trg && row_loss && loss_scale && dlogits_bf16 && counter && rows > 0 && V > 2 && factor > 0.f);
  BMHRL_CHECK_ARG(V % 4 == 0 && ld % 4 == 0 && ld >= V && ldg % 4 == 0 && ldg >= V && V <= 256 * 4 * 12);
  BMHRL_CHECK_ARG((((uintptr_t)logits & 15) | ((uintptr_t)dlogits_bf16 & 7) | ((uintptr_t)counter & 7)) == 0);
  const int nv = (V + 1023) / 1024;
#define HL(NV_) hipLaunchKernelGGL(head_loss_kernel<NV_>, dim3((unsigned)rows), dim3(256), 0, S_(stream), logits, (long)ld, trg,    \
                                   smoothing, pad_idx, weight, factor, dloss, row_loss, loss_scale, (bf16_t*)dlogits_bf16, (long)ldg, \
                                   counter, (long)rows, V)
  if (nv <= 2) HL(2); else if (nv <= 4) HL(4); else if (nv <= 8) HL(8); else HL(12);
#undef HL
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_smooth_kl_bwd(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg,
                                   const float* score, const float* n_row, float smoothing, int32_t pad_idx,
                                   int32_t zero_pad_rows, const float* loss_scale, const float* loss_scale2, int32_t wrt_logits,
                                   void* dlogits_bf16, int64_t ldg, float* dlogits_f32, int64_t rows, int32_t V,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logp && trg && loss_scale && (dlogits_bf16 || dlogits_f32) && rows > 0 && V > 2);
  BMHRL_CHECK_ARG(!biased_trg || (score && n_row));
  hipLaunchKernelGGL(smooth_kl_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), logp, (long)ld, trg, biased_trg,
                     score, n_row, smoothing, pad_idx, zero_pad_rows, loss_scale, loss_scale2, wrt_logits, (bf16_t*)dlogits_bf16,
                     (long)ldg, dlogits_f32, (long)rows, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_smooth_kl_amp_grad(const float* logp, int64_t ld, const int64_t* trg, const int64_t* biased_trg,
                                        const float* score, const float* n_row, float smoothing, int32_t pad_idx,
                                        int32_t zero_pad_rows, float* out, int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logp && trg && biased_trg && score && n_row && out && rows > 0 && V > 2);
  hipLaunchKernelGGL(smooth_kl_amp_grad_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, S_(stream), logp, (long)ld,
                     trg, biased_trg, score, n_row, smoothing, pad_idx, zero_pad_rows, out, (long)rows, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_log_softmax_bwd(const float* dlogp, const float* logp, int64_t ld, void* dlogits_bf16, int64_t ldg,
                                     int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(dlogp && logp && dlogits_bf16 && rows > 0 && V > 0 && ld >= V && ldg >= V);
  if (V % 4 == 0 && ld % 4 == 0 && ldg % 4 == 0 && V <= 256 * 4 * 12 &&
      ((((uintptr_t)dlogp | (uintptr_t)logp) & 15) == 0) && (((uintptr_t)dlogits_bf16 & 7) == 0)) {
    const int nv = (V + 1023) / 1024;
#define LSMB(NV_) hipLaunchKernelGGL(log_softmax_bwd_vec_kernel<NV_>, dim3((unsigned)rows), dim3(256), 0, S_(stream), dlogp, logp, \
                                     (long)ld, (bf16_t*)dlogits_bf16, (long)ldg, V)
    if (nv <= 1) LSMB(1); else if (nv <= 2) LSMB(2); else if (nv <= 4) LSMB(4); else if (nv <= 8) LSMB(8); else LSMB(12);
#undef LSMB
    return hip_status(hipGetLastError());
  }
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), dlogp, logp, (long)ld,
                     (bf16_t*)dlogits_bf16, (long)ldg, V);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_sample_tokens(const float* logp, int64_t ld, int64_t* out, float* p_out, int64_t rows, int32_t V,
                                   int32_t greedy, uint64_t seed, const uint64_t* seed_dev, int64_t row_offset,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(logp && out && rows > 0 && V > 0 && row_offset >= 0);
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), logp, (long)ld, out, p_out, V, greedy, seed,
                     seed_dev, (long)row_offset);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_reinforce_fwd(const float* pred, int64_t ld, int32_t is_logp, const int64_t* action, const float* value,
                                   const float* critic_value, float* row_policy, float* row_value, int64_t rows, int32_t V,
                                   bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(pred && action && value && critic_value && row_policy && row_value && rows > 0 && V > 0);
  hipLaunchKernelGGL(reinforce_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, S_(stream), pred, (long)ld, is_logp,
                     action, value, critic_value, row_policy, row_value, (long)rows);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_reinforce_bwd(const float* probs, int64_t ld, const int64_t* action, const float* value,
                                   const float* critic_value, const float* gscale, float* dprobs, float* dvalue,
                                   float* dcritic, int64_t rows, int32_t V, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(probs && action && value && critic_value && gscale && dprobs && rows > 0 && V > 0);
  hipLaunchKernelGGL(reinforce_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, S_(stream), probs, (long)ld, action, value,
                     critic_value, gscale, dprobs, dvalue, dcritic, (long)rows, V);
  return hip_status(hipGetLastError());
}
