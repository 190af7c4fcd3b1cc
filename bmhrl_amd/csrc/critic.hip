// Frozen segment critic (SegmentCritic, model/bm_hrl_agent.py:186-215) for gfx950: LSTM(4) -> AReLU -> GRU(2) -> AReLU ->
// Linear(600 -> 1) -> sigmoid > threshold, in fp32 (the labels are a hard threshold, so the arithmetic stays fp32:
// input projections on the f32-input MFMA v_mfma_f32_32x32x2_f32, which is a k-ordered fmaf chain, recurrences on VALU).
//
// The recurrence is inherently serial in time; it is expressed as one small launch per (layer, time step) that the
// host puts on a side stream / into the step's HIP graph.  Each launch: every block owns 4 hidden units (all their
// gate rows of W_hh, 16 x H floats) and all batch rows; h_{t-1} and the weight rows are staged in LDS (16-byte
// reads, the weight row is a broadcast across the 16 batch lanes); the gate pre-activations are exchanged through
// LDS and 4 x 16 threads apply the cell equations.
#include <cstdlib>
#include "common.h"
#include "../../include/bmhrl_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ fp32 GEMM (MFMA)
// C[M,N] = A[M,K] . W[N,K]^T + b1[N] (+ b2[N]);  one wave per 32x32 tile, 4 waves (2x2) per block; K % 4 == 0.
// v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; a lane loads 4 consecutive k
// of its row (16 bytes) and feeds elements {h, 2+h} to two MFMAs.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long lda, const float* __restrict__ W,
                                                        long ldw, const float* __restrict__ b1, const float* __restrict__ b2,
                                                        float* __restrict__ C, long ldc, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
  if (m0 >= M || n0 >= N) return;
  const float* ap = A + (long)min(m0 + r32, M - 1) * lda;
  const float* wp = W + (long)min(n0 + r32, N - 1) * ldw;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k);
    const f32x4 w = *reinterpret_cast<const f32x4*>(wp + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[1] : a[0], h ? w[1] : w[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[3] : a[2], h ? w[3] : w[2], acc, 0, 0, 0);
  }
  const int n = n0 + r32;
  if (n < N) {
    const float bias = (b1 ? b1[n] : 0.f) + (b2 ? b2[n] : 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M) C[(long)m * ldc + n] = acc[r] + bias;
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float arelu_(float x, float alpha, float beta) {
  // AReLU: relu(x) * (1 + sigmoid(beta)) - relu(-x) * clamp(alpha, .01, .99)   (model/bm_hrl_agent.py:13-23)
  return x > 0.f ? x * (1.f + sigmoidf_(beta)) : x * fminf(fmaxf(alpha, 0.01f), 0.99f);
}

constexpr int UB = 4;          // hidden units per block
constexpr int MAXH = 640;      // LDS rows hold up to 640 floats (H = 2 * d_model_caps = 600)

// One LSTM (GATES = 4: i,f,g,o) or GRU (GATES = 3: r,z,n) time step of one layer.
//   xproj: (B*L, GATES*H) = W_ih x + b_ih (+ b_hh for the LSTM)  row index b*L + t
//   GRU: bhh (3H) is added to the hidden projection (the n gate needs r * (W_hn h + b_hn))
template <int GATES>
__global__ __launch_bounds__(256) void rnn_step_kernel(const float* __restrict__ xproj, const float* __restrict__ whh,
                                                        const float* __restrict__ bhh, const float* __restrict__ h_prev,
                                                        const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                        float* __restrict__ c_out, float* __restrict__ seq_out,
                                                        const float* __restrict__ ar_alpha, const float* __restrict__ ar_beta,
                                                        int B, int L, int H, int t) {
  __shared__ __attribute__((aligned(16))) float sh_h[16][MAXH + 4];
  __shared__ __attribute__((aligned(16))) float sh_w[16][MAXH + 4];
  __shared__ float sh_g[16][17];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * UB;
  const int b0 = blockIdx.y * 16;
  const int slot = tid >> 4, bl = tid & 15;          // gate-row slot (gate = slot / UB, unit = slot % UB), batch lane
  const int gate = slot / UB, unit = u0 + slot % UB;
  const bool row_ok = gate < GATES && unit < H;
  // stage h_{t-1} of 16 batch rows and the 16 (12 for the GRU) weight rows
  for (int i = tid; i < 16 * (H / 4); i += 256) {
    const int r = i / (H / 4), c4 = (i % (H / 4)) * 4;
    f32x4 hv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
    if (b0 + r < B && t > 0) hv = *reinterpret_cast<const f32x4*>(h_prev + (long)(b0 + r) * H + c4);
    const int g = r / UB, u = u0 + r % UB;
    if (g < GATES && u < H) wv = *reinterpret_cast<const f32x4*>(whh + ((long)g * H + u) * H + c4);
    *reinterpret_cast<f32x4*>(&sh_h[r][c4]) = hv;
    *reinterpret_cast<f32x4*>(&sh_w[r][c4]) = wv;
  }
  __syncthreads();
  float acc = 0.f;
  if (row_ok && t > 0) {
    for (int k = 0; k < H; k += 4) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(&sh_w[slot][k]);
      const f32x4 hv = *reinterpret_cast<const f32x4*>(&sh_h[bl][k]);
      acc = fmaf(w[0], hv[0], acc); acc = fmaf(w[1], hv[1], acc); acc = fmaf(w[2], hv[2], acc); acc = fmaf(w[3], hv[3], acc);
    }
  }
  if (row_ok && GATES == 3) acc += bhh[gate * H + unit];
  sh_g[slot][bl] = acc;
  __syncthreads();
  if (tid < UB * 16) {
    const int u = u0 + (tid >> 4), b = b0 + bl;
    if (u < H && b < B) {
      const int ul = tid >> 4;
      const float* xp = xproj + ((long)b * L + t) * (GATES * H);
      float hn;
      if (GATES == 4) {
        const float gi = sigmoidf_(xp[u] + sh_g[ul][bl]);
        const float gf = sigmoidf_(xp[H + u] + sh_g[UB + ul][bl]);
        const float gg = tanhf(xp[2 * H + u] + sh_g[2 * UB + ul][bl]);
        const float go = sigmoidf_(xp[3 * H + u] + sh_g[3 * UB + ul][bl]);
        const float c = gf * (t > 0 ? c_prev[(long)b * H + u] : 0.f) + gi * gg;
        c_out[(long)b * H + u] = c;
        hn = go * tanhf(c);
      } else {
        const float r = sigmoidf_(xp[u] + sh_g[ul][bl]);
        const float z = sigmoidf_(xp[H + u] + sh_g[UB + ul][bl]);
        const float n = tanhf(xp[2 * H + u] + r * sh_g[2 * UB + ul][bl]);
        hn = (1.f - z) * n + z * (t > 0 ? sh_h[bl][u] : 0.f);
      }
      h_out[(long)b * H + u] = hn;
      seq_out[((long)b * L + t) * H + u] = ar_alpha ? arelu_(hn, ar_alpha[0], ar_beta[0]) : hn;
    }
  }
}

// ---- layer wavefront.  Cell (layer l, time t) needs (l-1, t) and (l, t-1) only, so launch s runs every cell with
// l + t == s at once (blockIdx.z = layer): L + n_layers - 1 = 35 launches of up to 6 x 150 blocks instead of 6 x (1 + 30)
// small ones in a row.  The input projection of the cell (W_ih x_t + b_ih [+ b_hh]) moves into the step: same k-ordered
// fmaf chain and the same association (chain + (b_ih + b_hh), then + the recurrent chain) as the GEMM + step pair above.
constexpr int MAX_RNN_LAYERS = 8;
struct RnnTable { bmhrl_rnn_layer l[MAX_RNN_LAYERS]; };

// `T` (chunk): layer l trails layer l - 1 by T time steps (cell (l, t) runs in launch t + T * l), so that at t % T == 0 the
// inputs of T steps are ready and W_ih is read ONCE for them (projections kept in P.xproj); the recurrent half of a cell
// still streams W_hh every step.  T = 1 is the plain diagonal.  Weight bytes per step of the whole stack:
// 30 * 34.5 MB (W_hh) + 30 / T * 31.7 MB (W_ih).
template <int SPT>
__global__ __launch_bounds__(256) void rnn_wave_kernel(const RnnTable tab, int n_layers, int B, int L, int H, int s, int T) {
  // A block owns UBW = 4 * SPT hidden units (all their gate rows: NS = 16 * SPT row slots) and 16 batch rows.  x_t / h_{t-1}
  // of the batch rows go through LDS; the weight rows do not (a row belongs to one slot, nothing in it is reused).  A
  // thread serves SPT slots, so every x value read from LDS feeds SPT weight rows (LDS reads, not FMAs, bound the cell).
  constexpr int UBW = UB * SPT, NS = 16 * SPT;
  __shared__ __attribute__((aligned(16))) float sh_h[16][MAXH + 4];
  __shared__ float sh_g[NS][17], sh_x[NS][17];
  const int layer = blockIdx.z, t = s - T * layer;
  if (t < 0 || t >= L) return;                      // uniform for the block
  const bmhrl_rnn_layer& P = tab.l[layer];
  const int GATES = P.gates;
  const int tid = threadIdx.x;
  // blockIdx.y = role * (batch blocks) + batch block.  Role 0 runs the cell (and, on the first step of a chunk, the input
  // projection of that step); roles 1 .. T-1 exist on chunk steps only and project step t + role of the chunk into P.xproj:
  // the T projections of a chunk run side by side instead of one after the other inside the cell's block.
  const int nb16 = (B + 15) / 16;
  const int role = (int)blockIdx.y / nb16;
  const int u0 = blockIdx.x * UBW, b0 = ((int)blockIdx.y - role * nb16) * 16;
  const int sg = tid >> 4, bl = tid & 15;
  bool row_ok[SPT];
  long wrow[SPT];
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    const int slot = sg + 16 * q, gate = slot / UBW, unit = u0 + slot % UBW;
    row_ok[q] = gate < GATES && unit < H;
    wrow[q] = row_ok[q] ? (long)gate * H + unit : 0;
  }
  constexpr int NP = (MAXH / 4 + 15) / 16;              // 16-byte pieces of a weight row per lane (10)
  // The 16 lanes of a slot split its weight row: lane j owns the pieces j, j + 16, ... (coalesced 256-byte reads, all of a
  // lane's pieces requested at once: one memory latency per row, not a chain of them) ...
  auto load_rows = [&](const float* __restrict__ w, const int K, f32x4 (&wv)[SPT][NP]) {
    const int nf = K / 4;
#pragma unroll
    for (int q = 0; q < SPT; ++q)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int f = bl + 16 * i;
        wv[q][i] = f < nf ? *reinterpret_cast<const f32x4*>(w + wrow[q] * K + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  // ... accumulates its slice against all 16 batch rows (x from LDS), and the 16 x 16 partial sums of a slot are
  // reduce-scattered with 15 shuffles so that lane j ends with batch row j: out[q] = sum_k w_q[k] * sh_h[bl][k]
  // (fixed summation order; it differs from a k-ascending chain by fp32 rounding only)
  auto apply_rows = [&](const f32x4 (&wv)[SPT][NP], const int K, float (&out)[SPT]) {
    const int nf = K / 4;
    float acc[SPT][16];
#pragma unroll
    for (int q = 0; q < SPT; ++q)
#pragma unroll
      for (int b = 0; b < 16; ++b) acc[q][b] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = bl + 16 * i;
      if (f < nf) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(&sh_h[b][4 * f]);
#pragma unroll
          for (int q = 0; q < SPT; ++q) {
            acc[q][b] = fmaf(wv[q][i][0], xv[0], acc[q][b]); acc[q][b] = fmaf(wv[q][i][1], xv[1], acc[q][b]);
            acc[q][b] = fmaf(wv[q][i][2], xv[2], acc[q][b]); acc[q][b] = fmaf(wv[q][i][3], xv[3], acc[q][b]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) {     // after the step with mask m a lane keeps the batch rows whose bit m equals its own
        const bool hi = (bl & m) != 0;
#pragma unroll
        for (int b = 0; b < m; ++b) {
          const float send = hi ? acc[q][b] : acc[q][b + m];
          const float keep = hi ? acc[q][b + m] : acc[q][b];
          acc[q][b] = keep + __shfl_xor(send, m, 64);
        }
      }
      out[q] = acc[q][0];
    }
  };
  auto stage = [&](const float* __restrict__ src, const long row_stride, const int K, const bool live) {
    for (int i = tid; i < 16 * (K / 4); i += 256) {
      const int r = i / (K / 4), c4 = (i % (K / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live && b0 + r < B) v = *reinterpret_cast<const f32x4*>(src + (long)(b0 + r) * row_stride + c4);
      *reinterpret_cast<f32x4*>(&sh_h[r][c4]) = v;
    }
  };
  // pass 1: input projection of this cell -- or, with T > 1 and t % T == 0, of the next T cells of the layer
  const int K = P.in_dim;
  const bool b_ok = b0 + bl < B;
  float bias_x[SPT], xp[SPT];
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    bias_x[q] = row_ok[q] ? P.b_ih[wrow[q]] + (GATES == 4 ? P.b_hh[wrow[q]] : 0.f) : 0.f;
    xp[q] = 0.f;
  }
  if (role > 0) {                                       // projection of step t + role of the chunk that starts at t
    if (T == 1 || t % T != 0 || t + role >= L) return;  // (uniform for the block)
    f32x4 wv[SPT][NP];
    load_rows(P.w_ih, K, wv);
    stage(P.in_seq + (long)(t + role) * P.in_ld, (long)L * P.in_ld, K, true);
    __syncthreads();
    float v[SPT];
    apply_rows(wv, K, v);
#pragma unroll
    for (int q = 0; q < SPT; ++q)
      if (row_ok[q] && b_ok) P.xproj[((long)(b0 + bl) * L + t + role) * (GATES * H) + wrow[q]] = v[q] + bias_x[q];
    return;
  }
  if (T == 1 || t % T == 0) {
    f32x4 wv[SPT][NP];
    load_rows(P.w_ih, K, wv);
    stage(P.in_seq + (long)t * P.in_ld, (long)L * P.in_ld, K, true);
    __syncthreads();
    float v[SPT];
    apply_rows(wv, K, v);                               // (every lane takes part in the shuffles)
#pragma unroll
    for (int q = 0; q < SPT; ++q) xp[q] = v[q] + bias_x[q];
    __syncthreads();
  } else if (b_ok) {
#pragma unroll
    for (int q = 0; q < SPT; ++q)                       // stored by this thread at the chunk's first step
      if (row_ok[q]) xp[q] = P.xproj[((long)(b0 + bl) * L + t) * (GATES * H) + wrow[q]];
  }
  // pass 2: h_{t-1}.  The recurrent weight rows are requested BEFORE h is staged: they depend on nothing, and their
  // latency (they come from the Infinity Cache: the six layers' 31.7 MB do not stay in an XCD's L2 between launches) then
  // passes under the h load -> LDS -> barrier chain instead of following it.
  float acc[SPT];
#pragma unroll
  for (int q = 0; q < SPT; ++q) acc[q] = 0.f;
  if (t > 0) {                                         // t is uniform
    f32x4 wv[SPT][NP];
    load_rows(P.w_hh, H, wv);
    stage(P.h[(t + 1) & 1], H, H, true);
    __syncthreads();
    apply_rows(wv, H, acc);
  } else {
    stage(P.h[(t + 1) & 1], H, H, false);
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    if (row_ok[q] && GATES == 3) acc[q] += P.b_hh[wrow[q]];
    sh_g[sg + 16 * q][bl] = acc[q];
    sh_x[sg + 16 * q][bl] = xp[q];
  }
  __syncthreads();
  if (tid < UBW * 16) {
    const int ul = tid >> 4, u = u0 + ul, b = b0 + bl;
    if (u < H && b < B) {
      float hn;
      if (GATES == 4) {
        const float gi = sigmoidf_(sh_x[ul][bl] + sh_g[ul][bl]);
        const float gf = sigmoidf_(sh_x[UBW + ul][bl] + sh_g[UBW + ul][bl]);
        const float gg = tanhf(sh_x[2 * UBW + ul][bl] + sh_g[2 * UBW + ul][bl]);
        const float go = sigmoidf_(sh_x[3 * UBW + ul][bl] + sh_g[3 * UBW + ul][bl]);
        const float c = gf * (t > 0 ? P.c[(t + 1) & 1][(long)b * H + u] : 0.f) + gi * gg;
        P.c[t & 1][(long)b * H + u] = c;
        hn = go * tanhf(c);
      } else {
        const float r = sigmoidf_(sh_x[ul][bl] + sh_g[ul][bl]);
        const float z = sigmoidf_(sh_x[UBW + ul][bl] + sh_g[UBW + ul][bl]);
        const float n = tanhf(sh_x[2 * UBW + ul][bl] + r * sh_g[2 * UBW + ul][bl]);
        hn = (1.f - z) * n + z * (t > 0 ? sh_h[bl][u] : 0.f);
      }
      P.h[t & 1][(long)b * H + u] = hn;
      P.seq_out[((long)b * L + t) * H + u] = P.arelu_alpha ? arelu_(hn, P.arelu_alpha[0], P.arelu_beta[0]) : hn;
    }
  }
}

// ---- the same wavefront cell on the f32 matrix pipe (r03, default at chunk 1).  The VALU cell above is bound by its LDS
// reads (one x value per two FMAs) and by the serial order load -> project -> load -> recur; here a block owns RW_UNITS = 16
// hidden units and 16 batch rows and runs 16 waves: wave (k half, pass, gate) multiplies the 16 rows of ONE gate of ONE
// matrix (pass 0: W_ih . x_t, pass 1: W_hh . h_{t-1}) over one half of k with v_mfma_f32_16x16x4_f32 (fp32 in, fp32
// accumulate: IEEE products and sums, only the order of a row's sum differs from the k-ascending chain).  Lane (r = lane &
// 15, q = lane >> 4) loads 16 bytes of weight row r at k = 16 step + 4 q straight from global memory (four lanes cover 64
// contiguous bytes) and reads the 16 bytes of batch row r at the same k from LDS: element e of both feeds MFMA e of the
// step, so the k mapping of A and B agree.  ALL of a wave's weight pieces (<= RW_STEPS) are requested before anything else:
// one memory latency per launch, passing under the x / h staging.  Every weight byte is read once per launch, an x value
// read from LDS feeds 16 rows; the matrix pipe's 256 flop / clock / CU (the VALU's packed rate) is the floor: ~4 us.
constexpr int RW_UNITS = 16, RW_LD = MAXH + 4;      // row stride = 4 (mod 32) banks: the 16-byte reads of the 16 batch rows are 2-way
constexpr int RW_STEPS = (MAXH / 16 + 1) / 2;       // 16-column steps of one k half

__global__ __launch_bounds__(1024) void rnn_wave_mfma_kernel(const RnnTable tab, int B, int L, int H, int s) {
  __shared__ __attribute__((aligned(16))) float sh_in[2][16][RW_LD];   // x_t | h_{t-1}, zero behind the last column
  __shared__ float sh_g[2][2][4][RW_UNITS][17];                        // [k half][pass][gate][unit][batch row]
  const int layer = blockIdx.z, t = s - layer;
  if (t < 0 || t >= L) return;                                         // uniform for the block
  const bmhrl_rnn_layer& P = tab.l[layer];
  const int GATES = P.gates, K1 = P.in_dim;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = wave >> 3, pass = (wave >> 2) & 1, gate = wave & 3;
  const int u0 = blockIdx.x * RW_UNITS, b0 = blockIdx.y * 16;
  const bool busy = gate < GATES && (pass == 0 || t > 0);              // uniform for the wave
  const int K = pass == 0 ? K1 : H, nsteps = (K + 15) / 16, nh = (nsteps + 1) / 2;
  const int step0 = half * nh, n = busy ? min(nh, nsteps - step0) : 0;
  const int r = lane & 15, kq = lane >> 4;
  const bool row_ok = u0 + r < H;
  const float* wrow = (pass == 0 ? P.w_ih : P.w_hh) + ((long)gate * H + (row_ok ? u0 + r : 0)) * K + 16 * step0 + 4 * kq;
  f32x4 piece[RW_STEPS];
#pragma unroll
  for (int i = 0; i < RW_STEPS; ++i)
    piece[i] = (i < n && row_ok && 16 * (step0 + i) + 4 * kq < K) ? *reinterpret_cast<const f32x4*>(wrow + 16 * i)
                                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int C4 = RW_LD / 4;
  for (int i = tid; i < 2 * 16 * C4; i += 1024) {
    const int img = i / (16 * C4), rem = i - img * (16 * C4), rr = rem / C4, c4 = (rem - rr * C4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b0 + rr < B) {
      if (img == 0) {
        if (c4 < K1) v = *reinterpret_cast<const f32x4*>(P.in_seq + ((long)(b0 + rr) * L + t) * P.in_ld + c4);
      } else if (t > 0 && c4 < H) {
        v = *reinterpret_cast<const f32x4*>(P.h[(t + 1) & 1] + (long)(b0 + rr) * H + c4);
      }
    }
    *reinterpret_cast<f32x4*>(&sh_in[img][rr][c4]) = v;
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* xb = &sh_in[pass][r][16 * step0 + 4 * kq];
#pragma unroll
  for (int i = 0; i < RW_STEPS; ++i) {
    if (i < n) {                                                       // uniform
      const f32x4 x = *reinterpret_cast<const f32x4*>(xb + 16 * i);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(piece[i][0], x[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(piece[i][1], x[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(piece[i][2], x[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(piece[i][3], x[3], acc, 0, 0, 0);
    }
  }
  if (gate < GATES) {                                                  // accumulator: unit 4 q + i, batch row r
#pragma unroll
    for (int i = 0; i < 4; ++i) sh_g[half][pass][gate][4 * kq + i][r] = acc[i];
  }
  __syncthreads();
  if (tid < 256) {
    const int ul = tid >> 4, bl = tid & 15, u = u0 + ul, b = b0 + bl;
    if (u < H && b < B) {
      float hn;
      if (GATES == 4) {                                                // same association as the VALU cell: (x part + (b_ih + b_hh)) + h part
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pre[g] = ((sh_g[0][0][g][ul][bl] + sh_g[1][0][g][ul][bl]) + (P.b_ih[(long)g * H + u] + P.b_hh[(long)g * H + u])) +
                   (sh_g[0][1][g][ul][bl] + sh_g[1][1][g][ul][bl]);
        const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
        const float c = gf * (t > 0 ? P.c[(t + 1) & 1][(long)b * H + u] : 0.f) + gi * gg;
        P.c[t & 1][(long)b * H + u] = c;
        hn = go * tanhf(c);
      } else {
        float xp[3], hp[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          xp[g] = (sh_g[0][0][g][ul][bl] + sh_g[1][0][g][ul][bl]) + P.b_ih[(long)g * H + u];
          hp[g] = (sh_g[0][1][g][ul][bl] + sh_g[1][1][g][ul][bl]) + P.b_hh[(long)g * H + u];
        }
        const float rg = sigmoidf_(xp[0] + hp[0]);
        const float z = sigmoidf_(xp[1] + hp[1]);
        const float nn = tanhf(xp[2] + rg * hp[2]);
        hn = (1.f - z) * nn + z * sh_in[1][bl][u];                     // (zero at t = 0)
      }
      P.h[t & 1][(long)b * H + u] = hn;
      P.seq_out[((long)b * L + t) * H + u] = P.arelu_alpha ? arelu_(hn, P.arelu_alpha[0], P.arelu_beta[0]) : hn;
    }
  }
}

// labels[row] = sigmoid(lin_w . x[row] + lin_b) > thr ; also the raw score (for tests / callers)
__global__ void critic_head_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                   float thr, float* __restrict__ score, int32_t* __restrict__ labels, long rows, int H) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int k = lane; k < H; k += 64) acc = fmaf(x[row * H + k], w[k], acc);
  acc = wave_sum(acc) + b[0];
  if (lane == 0) {
    if (score) score[row] = acc;
    if (labels) labels[row] = sigmoidf_(acc) > thr ? 1 : 0;
  }
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" int bmhrl_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias1,
                              const float* bias2, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                              bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0);
  BMHRL_CHECK_ARG((((uintptr_t)A | (uintptr_t)W) & 15) == 0);
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), block(256);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, S_(stream), A, (long)lda, W, (long)ldw, bias1, bias2, C, (long)ldc, M, N, K);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_rnn_step(int32_t gates, const float* xproj, const float* whh, const float* bhh, const float* h_prev,
                              const float* c_prev, float* h_out, float* c_out, float* seq_out, const float* arelu_alpha,
                              const float* arelu_beta, int32_t B, int32_t L, int32_t H, int32_t t, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG((gates == 3 || gates == 4) && xproj && whh && h_out && seq_out && B > 0 && L > 0 && t >= 0 && t < L);
  BMHRL_CHECK_ARG(H > 0 && H <= MAXH && H % 4 == 0 && (t == 0 || h_prev) && (gates == 3 ? bhh != nullptr : c_out != nullptr));
  BMHRL_CHECK_ARG((arelu_alpha == nullptr) == (arelu_beta == nullptr));
  dim3 grid((unsigned)((H + UB - 1) / UB), (unsigned)((B + 15) / 16)), block(256);
  if (gates == 4)
    hipLaunchKernelGGL(rnn_step_kernel<4>, grid, block, 0, S_(stream), xproj, whh, bhh, h_prev, c_prev, h_out, c_out, seq_out,
                       arelu_alpha, arelu_beta, B, L, H, t);
  else
    hipLaunchKernelGGL(rnn_step_kernel<3>, grid, block, 0, S_(stream), xproj, whh, bhh, h_prev, c_prev, h_out, c_out, seq_out,
                       arelu_alpha, arelu_beta, B, L, H, t);
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_rnn_wavefront(const bmhrl_rnn_layer* layers, int32_t n_layers, int32_t B, int32_t L, int32_t H,
                                   int32_t chunk, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(layers && n_layers > 0 && n_layers <= MAX_RNN_LAYERS && B > 0 && L > 0 && H > 0 && H <= MAXH && H % 4 == 0);
  BMHRL_CHECK_ARG(chunk >= 1 && chunk <= 16);
  RnnTable tab;
  for (int i = 0; i < n_layers; ++i) {
    const bmhrl_rnn_layer& p = layers[i];
    BMHRL_CHECK_ARG((p.gates == 3 || p.gates == 4) && p.w_ih && p.w_hh && p.b_ih && p.b_hh && p.in_seq && p.seq_out && p.h[0] && p.h[1]);
    BMHRL_CHECK_ARG(p.in_dim > 0 && p.in_dim <= MAXH && p.in_dim % 4 == 0 && p.in_ld % 4 == 0 && p.in_ld >= p.in_dim);
    BMHRL_CHECK_ARG(p.gates == 3 || (p.c[0] && p.c[1]));
    BMHRL_CHECK_ARG(chunk == 1 || p.xproj != nullptr);
    BMHRL_CHECK_ARG((p.arelu_alpha == nullptr) == (p.arelu_beta == nullptr));
    BMHRL_CHECK_ARG((((uintptr_t)p.w_ih | (uintptr_t)p.w_hh | (uintptr_t)p.in_seq | (uintptr_t)p.h[0] | (uintptr_t)p.h[1]) & 15) == 0);
    tab.l[i] = p;
  }
  static const bool mfma = !(getenv("BMHRL_RNN_MFMA") && atoi(getenv("BMHRL_RNN_MFMA")) == 0);
  if (chunk == 1 && mfma) {                             // the matrix-pipe cell: one launch per diagonal, 16 waves per block
    const dim3 grid((unsigned)((H + RW_UNITS - 1) / RW_UNITS), (unsigned)((B + 15) / 16), (unsigned)n_layers), block(1024);
    for (int s = 0; s < L + n_layers - 1; ++s)
      hipLaunchKernelGGL(rnn_wave_mfma_kernel, grid, block, 0, S_(stream), tab, B, L, H, s);
    return hip_status(hipGetLastError());
  }
  static const int spt = getenv("BMHRL_RNN_SPT") ? atoi(getenv("BMHRL_RNN_SPT")) : 2;     // gate rows per thread (tuning aid: 1)
  static const bool vargrid = getenv("BMHRL_RNN_VARGRID") && atoi(getenv("BMHRL_RNN_VARGRID")) != 0;
  const int ubw = UB * (spt == 1 ? 1 : 2);
  const unsigned nb16 = (unsigned)((B + 15) / 16);
  for (int s = 0; s < L + chunk * (n_layers - 1); ++s) {
    // launches in which the layers start a chunk (every layer trails the one below by `chunk` steps, so they all do in the
    // same launches) carry the extra projection roles
    // (the same grid in every launch; the extra roles of a non-chunk launch exit at once.  BMHRL_RNN_VARGRID=1 launches the
    // role blocks on chunk steps only -- the r02 form, kept as a diagnostic switch: see DESIGN.md section 10)
    const unsigned roles = (vargrid && !(chunk > 1 && s % chunk == 0)) ? 1u : (unsigned)chunk;
    dim3 grid((unsigned)((H + ubw - 1) / ubw), nb16 * roles, (unsigned)n_layers), block(256);
    if (spt == 1) hipLaunchKernelGGL(rnn_wave_kernel<1>, grid, block, 0, S_(stream), tab, n_layers, B, L, H, s, chunk);
    else hipLaunchKernelGGL(rnn_wave_kernel<2>, grid, block, 0, S_(stream), tab, n_layers, B, L, H, s, chunk);
  }
  return hip_status(hipGetLastError());
}

extern "C" int bmhrl_critic_head(const float* x, const float* w, const float* b, float threshold, float* score,
                                 int32_t* labels, int64_t rows, int32_t H, bmhrl_stream_t stream) {
  BMHRL_CHECK_ARG(x && w && b && (score || labels) && rows > 0 && H > 0);
  hipLaunchKernelGGL(critic_head_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(stream), x, w, b, threshold, score,
                     labels, (long)rows, H);
  return hip_status(hipGetLastError());
}
