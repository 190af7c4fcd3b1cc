// Native tuning / validation harness for bmhrl_gemm (no Python: starts in milliseconds on a fresh GPU box).
//
//   gemm_bench check          layouts x shapes (incl. ragged edges, batches, split-K) against an fp64 CPU dot product on
//                             sampled outputs
//   gemm_bench time [iters]   the hot path's shapes (tests/bench_gemm.py), HIP events around back-to-back launches
//
// BMHRL_GEMM_NOGLDS=1 in the environment selects the register-staged main loop (A/B of the two loops in one session).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/bmhrl_hip.h"

#define CK(x)                                                                             \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static int pad8(int n) { return (n + 7) & ~7; }

struct Shape {
  int M, N, K, at, bt, nb;
  int out_bf16, split;
  const char* tag;
};

struct Run {
  std::vector<uint16_t> A, B;
  uint16_t *dA = nullptr, *dB = nullptr, *dCb = nullptr;
  float* dC = nullptr;
  long lda, ldb, a_sb, b_sb;
};

static void make(const Shape& s, Run& r, unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  r.lda = s.at ? pad8(s.M) : pad8(s.K);
  r.ldb = s.bt ? pad8(s.N) : pad8(s.K);
  const long a_rows = s.at ? s.K : s.M, b_rows = s.bt ? s.K : s.N;
  r.a_sb = a_rows * r.lda;
  r.b_sb = b_rows * r.ldb;
  r.A.assign((size_t)s.nb * r.a_sb, 0);
  r.B.assign((size_t)s.nb * r.b_sb, 0);
  const int a_cols = s.at ? s.M : s.K, b_cols = s.bt ? s.N : s.K;
  for (int b = 0; b < s.nb; ++b) {
    for (long i = 0; i < a_rows; ++i)
      for (int j = 0; j < a_cols; ++j) r.A[(size_t)b * r.a_sb + i * r.lda + j] = f2bf(ud(rng));
    for (long i = 0; i < b_rows; ++i)
      for (int j = 0; j < b_cols; ++j) r.B[(size_t)b * r.b_sb + i * r.ldb + j] = f2bf(ud(rng));
  }
  CK(hipMalloc(&r.dA, r.A.size() * 2));
  CK(hipMalloc(&r.dB, r.B.size() * 2));
  CK(hipMemcpy(r.dA, r.A.data(), r.A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(r.dB, r.B.data(), r.B.size() * 2, hipMemcpyHostToDevice));
  const size_t n = (size_t)s.nb * s.M * s.N;
  if (s.out_bf16) { CK(hipMalloc(&r.dCb, n * 2)); CK(hipMemset(r.dCb, 0, n * 2)); }
  else { CK(hipMalloc(&r.dC, n * 4)); CK(hipMemset(r.dC, 0, n * 4)); }
}
static void release(Run& r) { (void)hipFree(r.dA); (void)hipFree(r.dB); (void)hipFree(r.dC); (void)hipFree(r.dCb); }

// `one` mode: the epilogue of the residual blocks (out-projection / second feed-forward product): fp32 C = dropout(acc + bias) + residual
static int g_epi = 0;                  // 0 plain, 1 bias + dropout 0.1 + residual, 2 the same + a bf16 copy of the output
static float *g_bias = nullptr, *g_res = nullptr;
static uint16_t* g_cb = nullptr;

static int launch(const Shape& s, const Run& r, hipStream_t st) {
  bmhrl_gemm_desc d;
  memset(&d, 0, sizeof d);
  d.M = s.M; d.N = s.N; d.K = s.K; d.batch1 = s.nb; d.batch2 = 1;
  d.A = r.dA; d.lda = r.lda; d.a_sb1 = r.a_sb; d.a_trans = s.at;
  d.B = r.dB; d.ldb = r.ldb; d.b_sb1 = r.b_sb; d.b_trans = s.bt;
  d.C = r.dC; d.ldc = s.N; d.c_sb1 = (long)s.M * s.N;
  d.Cb = r.dCb; d.ldcb = s.N; d.cb_sb1 = (long)s.M * s.N;
  d.epilogue = BMHRL_EPI_LINEAR; d.alpha = 1.f;
  d.allow_split_k = s.split;
  if (g_epi && !s.out_bf16) {
    d.bias = g_bias; d.residual = g_res; d.ldr = s.N; d.r_sb1 = (long)s.M * s.N; d.dropout_p = 0.1f; d.seed = 77;
    if (g_epi == 2) { d.Cb = g_cb; d.ldcb = s.N; d.cb_sb1 = (long)s.M * s.N; }
  }
  return bmhrl_gemm(&d, st);
}

static bool verify(const Shape& s, const Run& r) {
  const size_t n = (size_t)s.nb * s.M * s.N;
  std::vector<float> C(n);
  if (s.out_bf16) {
    std::vector<uint16_t> t(n);
    CK(hipMemcpy(t.data(), r.dCb, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) C[i] = bf2f(t[i]);
  } else {
    CK(hipMemcpy(C.data(), r.dC, n * 4, hipMemcpyDeviceToHost));
  }
  std::mt19937 rng(7);
  double worst = 0;
  const int samples = 600;
  for (int it = 0; it < samples; ++it) {
    const int b = rng() % s.nb;
    int m = rng() % s.M, nn = rng() % s.N;
    if (it < 8) { m = (it & 1) ? s.M - 1 : 0; nn = (it & 2) ? s.N - 1 : 0; }            // the corners
    if (it >= 8 && it < 40) { m = std::min(s.M - 1, (int)(rng() % 4) + (s.M / 128) * 128); nn = std::min(s.N - 1, (int)(rng() % 4) + (s.N / 128) * 128); }
    double acc = 0, mag = 0;
    for (int k = 0; k < s.K; ++k) {
      const float a = bf2f(s.at ? r.A[(size_t)b * r.a_sb + (long)k * r.lda + m] : r.A[(size_t)b * r.a_sb + (long)m * r.lda + k]);
      const float bb = bf2f(s.bt ? r.B[(size_t)b * r.b_sb + (long)k * r.ldb + nn] : r.B[(size_t)b * r.b_sb + (long)nn * r.ldb + k]);
      acc += (double)a * bb;
      mag += std::fabs((double)a * bb);
    }
    const double got = C[((size_t)b * s.M + m) * s.N + nn];
    const double tol = (s.out_bf16 ? 4e-3 : 2e-6) * std::max(mag, 1.0) + (s.out_bf16 ? 4e-3 * std::fabs(acc) : 0);
    worst = std::max(worst, std::fabs(got - acc) / tol);
  }
  const bool ok = worst <= 1.0;
  printf("  %-22s M%-6d N%-6d K%-6d at%d bt%d nb%-2d %s%s  worst err / tol %.3f  %s\n", s.tag, s.M, s.N, s.K, s.at, s.bt, s.nb,
         s.out_bf16 ? "bf16" : "f32 ", s.split ? " splitK" : "", worst, ok ? "ok" : "FAIL");
  return ok;
}

static double time_shape(const Shape& s, const Run& r, int iters) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(s, r, st);
  CK(hipStreamSynchronize(st));
  std::vector<float> ms;
  for (int rep = 0; rep < 5; ++rep) {
    if (s.split) CK(hipMemsetAsync(r.dC, 0, (size_t)s.nb * s.M * s.N * 4, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(s, r, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / iters);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipStreamDestroy(st));
  return ms[ms.size() / 2] * 1e3;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "check";
  if (mode == "check") {
    const Shape shapes[] = {
        {256, 256, 128, 0, 0, 1, 0, 0, "small nn"},        {300, 200, 192, 0, 0, 1, 1, 0, "ragged MN nn"},
        {300, 200, 192, 0, 1, 1, 0, 0, "ragged MN nt"},    {300, 200, 192, 1, 0, 1, 0, 0, "ragged MN tn"},
        {300, 200, 192, 1, 1, 1, 1, 0, "ragged MN tt"},    {130, 70, 64, 0, 0, 3, 0, 0, "batched 64x64 nn"},
        {130, 70, 64, 1, 1, 3, 0, 0, "batched 64x64 tt"},  {70, 130, 128, 0, 1, 2, 1, 0, "batched nt"},
        {70, 130, 128, 1, 0, 2, 0, 0, "batched tn"},       {480, 300, 1024, 0, 0, 1, 0, 0, "C out proj"},
        {4096, 1024, 1024, 0, 0, 1, 0, 0, "V out"},        {4096, 1024, 3072, 0, 1, 1, 0, 0, "V qkv dx"},
        {1024, 1024, 4096, 1, 1, 1, 0, 1, "V dW splitK"},  {3072, 128, 12800, 1, 1, 1, 0, 1, "A qkv dW splitK"},
        {128, 1024, 12800, 1, 1, 1, 0, 1, "A out dW splitK"}, {12800, 128, 1024, 0, 0, 1, 0, 0, "A out"},
        {480, 1024, 300, 0, 0, 1, 1, 0, "ragged K (old loop)"}, {257, 129, 100, 1, 1, 1, 0, 0, "ragged all tt"},
        {4096, 3072, 1024, 0, 0, 1, 1, 0, "V qkv"},        {800, 800, 256, 0, 0, 16, 1, 0, "attn S"},
        {800, 256, 800, 1, 1, 16, 1, 0, "attn dV (K=800: old loop)"},
        // 256 tiles of 128 x 128: the eight-wave kernel (gemm_glds8_kernel), every operand layout, ragged edges, a batch
        {2048, 2048, 512, 0, 0, 1, 1, 0, "8 waves nn"},   {2048, 2048, 512, 0, 1, 1, 0, 0, "8 waves nt"},
        {2048, 2048, 512, 1, 0, 1, 0, 0, "8 waves tn"},   {2048, 2048, 512, 1, 1, 1, 1, 0, "8 waves tt"},
        {2000, 1990, 192, 0, 0, 1, 0, 0, "8 waves ragged nn"}, {2000, 1990, 192, 1, 1, 1, 0, 0, "8 waves ragged tt"},
        {1000, 1000, 128, 0, 1, 4, 1, 0, "8 waves batch nt"},
    };
    bool all_ok = true;
    for (const Shape& s : shapes) {
      Run r;
      make(s, r, 1000u + s.M + 3 * s.N + 7 * s.K);
      const int rc = launch(s, r, 0);
      CK(hipDeviceSynchronize());
      if (rc != 0) { printf("  %s: launch rc %d\n", s.tag, rc); all_ok = false; release(r); continue; }
      all_ok &= verify(s, r);
      release(r);
    }
    printf(all_ok ? "ALL OK\n" : "FAILURES\n");
    return all_ok ? 0 : 1;
  }
  if (mode == "time") {
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const Shape shapes[] = {
        {4096, 3072, 1024, 0, 0, 1, 1, 0, "V qkv fwd"},      {4096, 1024, 1024, 0, 0, 1, 0, 0, "V out/ffn fwd"},
        {12800, 3072, 128, 0, 0, 1, 1, 0, "A qkv fwd"},      {12800, 128, 1024, 0, 0, 1, 0, 0, "A out fwd"},
        {4096, 2048, 1024, 0, 0, 1, 1, 0, "KV(V) fwd"},      {4096, 1024, 3072, 0, 1, 1, 0, 0, "V qkv dx"},
        {4096, 1024, 1024, 0, 1, 1, 1, 0, "V out dx"},       {12800, 128, 3072, 0, 1, 1, 0, 0, "A qkv dx"},
        {1024, 1024, 4096, 1, 1, 1, 0, 1, "V dW"},           {3072, 1024, 4096, 1, 1, 1, 0, 1, "V qkv dW"},
        {3072, 128, 12800, 1, 1, 1, 0, 1, "A qkv dW"},       {128, 1024, 12800, 1, 1, 1, 0, 1, "A out dW"},
        {800, 800, 256, 0, 0, 64, 1, 0, "attn S (A self)"},  {480, 1024, 1024, 0, 0, 1, 1, 0, "C 1024 fwd"},
        {480, 300, 1024, 0, 0, 1, 0, 0, "C out proj fwd"},   {8192, 8192, 8192, 0, 0, 1, 1, 0, "8192^3 nn"},
        {4096, 4096, 4096, 0, 0, 1, 1, 0, "4096^3 nn"},
    };
    for (const Shape& s : shapes) {
      Run r;
      make(s, r, 5u);
      if (launch(s, r, 0) != 0) { printf("%s: launch failed\n", s.tag); release(r); continue; }
      CK(hipDeviceSynchronize());
      const double us = time_shape(s, r, iters);
      const double gf = 2.0 * s.M * s.N * (double)s.K * s.nb / 1e9;
      printf("%-18s M=%6d N=%6d K=%6d at=%d bt=%d nb=%2d %s %8.1f us %7.1f TF/s\n", s.tag, s.M, s.N, s.K, s.at, s.bt, s.nb,
             s.out_bf16 ? "bf16" : "f32 ", us, gf / us * 1e3);
      fflush(stdout);
      release(r);
    }
    return 0;
  }
  if (mode == "one" && argc >= 9) {      // one M N K at bt nb out_bf16 [iters]   (with the trace build: BMHRL_GEMM_TRACE=1 prints stamps)
    Shape s{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), 0, "one"};
    const int iters = argc > 9 ? atoi(argv[9]) : 20;
    g_epi = argc > 10 ? atoi(argv[10]) : 0;
    Run r;
    make(s, r, 5u);
    if (g_epi) {
      const size_t n = (size_t)s.nb * s.M * s.N;
      CK(hipMalloc(&g_bias, (size_t)s.N * 4)); CK(hipMemset(g_bias, 0, (size_t)s.N * 4));
      CK(hipMalloc(&g_res, n * 4)); CK(hipMemset(g_res, 0, n * 4));
      CK(hipMalloc(&g_cb, n * 2));
    }
    if (launch(s, r, 0) != 0) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    const double us = time_shape(s, r, iters);
    const double gf = 2.0 * s.M * s.N * (double)s.K * s.nb / 1e9;
    printf("M=%d N=%d K=%d at=%d bt=%d nb=%d %s epi%d %8.1f us %7.1f TF/s\n", s.M, s.N, s.K, s.at, s.bt, s.nb, s.out_bf16 ? "bf16" : "f32 ",
           g_epi, us, gf / us * 1e3);
    release(r);
    return 0;
  }
  fprintf(stderr, "usage: gemm_bench check | time [iters] | one M N K at bt nb out_bf16 [iters]\n");
  return 2;
}
