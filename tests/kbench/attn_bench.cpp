// Native tuning / validation harness for the attention forward kernels (no Python, no torch: starts in milliseconds on a
// fresh GPU box).  Links against bmhrl_amd/csrc/libbmhrl_hip.so and calls the C ABI only.
//
//   attn_bench check            every (form, split) on small + odd shapes against an fp64 CPU restatement of
//                               model/multihead_attention.py:7-31 (sampled rows), incl. masks, fully masked rows, a forced rescale
//   attn_bench time [iters]     launch times of the reference shapes for every split (HIP events around back-to-back launches)
//   attn_bench one form B H Sq Sk code maskmode iters
//
// Build: hipcc -O2 -std=c++17 tests/kbench/attn_bench.cpp -o tests/kbench/attn_bench -Lbmhrl_amd/csrc -lbmhrl_hip -Wl,-rpath,'$ORIGIN/../../bmhrl_amd/csrc'
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

extern "C" int bmhrl_attention_config(int32_t head_dim, int32_t code);
extern "C" int64_t bmhrl_attention_shared128_bwd_workspace(int32_t B, int32_t H, int32_t Sk);
extern "C" int bmhrl_attention_shared128_bwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, const void* dCx, int64_t lddo,
                                             const float* row_max, const float* row_sum, const float* delta, const uint8_t* mask,
                                             int64_t mask_sb, void* dQp, int64_t lddq, float* dX, int64_t lddx, int32_t accumulate_dx,
                                             float* workspace, int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale,
                                             bmhrl_stream_t stream);
extern "C" int bmhrl_attn_delta(const void* dO, int64_t lddo, const void* O, int64_t ldo, float* delta, float scale, int32_t B,
                                int32_t H, int32_t Sq, int32_t dk, bmhrl_stream_t stream);

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
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

struct Case {
  int form;  // 256: bmhrl_attention_fwd; 128: bmhrl_attention_shared128_fwd
  int B, H, Sq, Sk;
  int maskmode;  // 0 none, 1 all ones, 2 padded suffix ((b*37) % (Sk/4) keys), 3 random 15 % + batch row 1 fully masked + suffix
  int spike;     // scale the keys of the second half of batch row 0 by 8 (forces the lazy rescale)
  int qmask;     // per-query mask (B, Sq, Sk): form 256 only
};

struct Buffers {
  std::vector<uint16_t> q, k, v;
  std::vector<uint8_t> mask;
  uint16_t *dq = nullptr, *dk = nullptr, *dv = nullptr, *dout = nullptr;
  uint8_t* dmask = nullptr;
  float *dmax = nullptr, *dsum = nullptr;
  long ldq, ldk, ldv, ldo;
};

static void make(const Case& c, Buffers& b, unsigned seed) {
  const int DK = c.form;
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, 1.f);
  b.ldq = (long)c.H * DK;
  b.ldo = b.ldq;
  b.ldk = c.form == 256 ? (long)c.H * DK : DK;
  b.ldv = b.ldk;
  b.q.resize((size_t)c.B * c.Sq * b.ldq);
  b.k.resize((size_t)c.B * c.Sk * b.ldk);
  for (auto& x : b.q) x = f2bf(nd(rng));
  for (size_t i = 0; i < b.k.size(); ++i) {
    float x = nd(rng);
    if (c.spike) {
      const long row = (long)(i / b.ldk);
      if (row < c.Sk && row >= c.Sk / 2) x *= 8.f;
    }
    b.k[i] = f2bf(x);
  }
  if (c.form == 256) {
    b.v.resize(b.k.size());
    for (auto& x : b.v) x = f2bf(nd(rng));
  }
  const long mrows = c.qmask ? c.Sq : 1;
  b.mask.assign((size_t)c.B * mrows * c.Sk, 1);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  for (int bb = 0; bb < c.B; ++bb)
    for (long r = 0; r < mrows; ++r)
      for (int kk = 0; kk < c.Sk; ++kk) {
        uint8_t m = 1;
        if (c.maskmode == 2 || c.maskmode == 3) {
          const int pad = (bb * 37) % std::max(1, c.Sk / 4);
          if (kk >= c.Sk - pad) m = 0;
        }
        if (c.maskmode == 3) {
          if (ud(rng) < 0.15f) m = 0;
          if (bb == 1) m = 0;
        }
        if (c.qmask && kk > r) m = 0;   // causal-like per-query mask
        b.mask[((size_t)bb * mrows + r) * c.Sk + kk] = m;
      }
  CK(hipMalloc(&b.dq, b.q.size() * 2));
  CK(hipMalloc(&b.dk, b.k.size() * 2));
  CK(hipMemcpy(b.dq, b.q.data(), b.q.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b.dk, b.k.data(), b.k.size() * 2, hipMemcpyHostToDevice));
  if (c.form == 256) {
    CK(hipMalloc(&b.dv, b.v.size() * 2));
    CK(hipMemcpy(b.dv, b.v.data(), b.v.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&b.dout, (size_t)c.B * c.Sq * b.ldo * 2));
  CK(hipMemset(b.dout, 0xff, (size_t)c.B * c.Sq * b.ldo * 2));
  CK(hipMalloc(&b.dmask, b.mask.size()));
  CK(hipMemcpy(b.dmask, b.mask.data(), b.mask.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&b.dmax, (size_t)c.B * c.H * c.Sq * 4));
  CK(hipMalloc(&b.dsum, (size_t)c.B * c.H * c.Sq * 4));
}
static void release(Buffers& b) {
  (void)hipFree(b.dq); (void)hipFree(b.dk); (void)hipFree(b.dv); (void)hipFree(b.dout); (void)hipFree(b.dmask);
  (void)hipFree(b.dmax); (void)hipFree(b.dsum);
}

static int launch(const Case& c, const Buffers& b, hipStream_t s) {
  const float scale = 1.f / 16.f;
  const uint8_t* m = c.maskmode == 0 && !c.qmask ? nullptr : b.dmask;
  if (c.form == 256)
    return bmhrl_attention_fwd(b.dq, b.ldq, b.dk, b.ldk, b.dv, b.ldv, b.dout, b.ldo, b.dmax, b.dsum, m,
                               c.qmask ? (long)c.Sq * c.Sk : c.Sk, c.qmask ? c.Sk : 0, c.B, c.H, c.Sq, c.Sk, 256, scale, 0.f, 0,
                               nullptr, s);
  return bmhrl_attention_shared128_fwd(b.dq, b.ldq, b.dk, b.ldk, b.dout, b.ldo, b.dmax, b.dsum, m, c.Sk, c.B, c.H, c.Sq, c.Sk,
                                       scale, s);
}

// fp64 restatement on sampled (b, h, q) rows; returns max |o - ref| / max |ref| and the worst log-sum-exp error
static bool verify(const Case& c, const Buffers& b, const char* tag) {
  const int DK = c.form;
  std::vector<uint16_t> out((size_t)c.B * c.Sq * b.ldo);
  std::vector<float> rmax((size_t)c.B * c.H * c.Sq), rsum(rmax.size());
  CK(hipMemcpy(out.data(), b.dout, out.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rmax.data(), b.dmax, rmax.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rsum.data(), b.dsum, rsum.size() * 4, hipMemcpyDeviceToHost));
  const float scale = 1.f / 16.f;
  const std::vector<uint16_t>& vv = c.form == 256 ? b.v : b.k;
  double worst = 0, worst_lse = 0, refmax = 0;
  std::vector<int> qs;
  for (int q = 0; q < c.Sq; q += std::max(1, c.Sq / 23)) qs.push_back(q);
  qs.push_back(c.Sq - 1);
  if (c.Sq > 40) { qs.push_back(31); qs.push_back(32); qs.push_back(33); }
  std::vector<double> sc(c.Sk), o(DK);
  for (int bb = 0; bb < c.B; ++bb)
    for (int h = 0; h < c.H; ++h) {
      if (c.B * c.H > 12 && !((bb * c.H + h) % 5 == 0 || bb == 1 || (bb == c.B - 1 && h == c.H - 1))) continue;
      for (int q : qs) {
        const uint16_t* qp = &b.q[((size_t)bb * c.Sq + q) * b.ldq + (size_t)h * DK];
        double mx = -1e300;
        for (int kk = 0; kk < c.Sk; ++kk) {
          const uint16_t* kp = &b.k[((size_t)bb * c.Sk + kk) * b.ldk + (c.form == 256 ? (size_t)h * DK : 0)];
          double s = 0;
          for (int d = 0; d < DK; ++d) s += (double)bf2f(qp[d]) * bf2f(kp[d]);
          s *= scale;
          const uint8_t m = (c.maskmode == 0 && !c.qmask) ? 1 : b.mask[((size_t)bb * (c.qmask ? c.Sq : 1) + (c.qmask ? q : 0)) * c.Sk + kk];
          if (!m) s = -1e9;
          sc[kk] = s;
          mx = std::max(mx, s);
        }
        double l = 0;
        for (int kk = 0; kk < c.Sk; ++kk) { sc[kk] = std::exp(sc[kk] - mx); l += sc[kk]; }
        std::fill(o.begin(), o.end(), 0.0);
        for (int kk = 0; kk < c.Sk; ++kk) {
          const uint16_t* vp = &vv[((size_t)bb * c.Sk + kk) * b.ldv + (c.form == 256 ? (size_t)h * DK : 0)];
          const double pk = sc[kk] / l;
          for (int d = 0; d < DK; ++d) o[d] += pk * bf2f(vp[d]);
        }
        const uint16_t* op = &out[((size_t)bb * c.Sq + q) * b.ldo + (size_t)h * DK];
        for (int d = 0; d < DK; ++d) {
          const float got = bf2f(op[d]);
          if (!std::isfinite(got)) { printf("  %s: non-finite output at b%d h%d q%d d%d\n", tag, bb, h, q, d); return false; }
          worst = std::max(worst, std::fabs(got - o[d]));
          refmax = std::max(refmax, std::fabs(o[d]));
        }
        const size_t si = ((size_t)bb * c.H + h) * c.Sq + q;
        const double lse = (double)rmax[si] + std::log((double)rsum[si]), lse_ref = mx + std::log(l);
        worst_lse = std::max(worst_lse, std::fabs(lse - lse_ref) / std::max(1.0, std::fabs(lse_ref)));
      }
    }
  const double rel = worst / std::max(refmax, 1e-30);
  const bool ok = rel < 1.2e-2 && worst_lse < 2e-5;
  printf("  %-34s max|o-ref|/max|ref| %.2e  lse rel err %.1e  %s\n", tag, rel, worst_lse, ok ? "ok" : "FAIL");
  return ok;
}

static double time_case(const Case& c, const Buffers& b, int iters) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch(c, b, s);
  CK(hipStreamSynchronize(s));
  std::vector<float> ms;
  for (int rep = 0; rep < 7; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch(c, b, s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t / iters);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipStreamDestroy(s));
  return ms[ms.size() / 2] * 1e3;   // us
}

static const int CODES256[] = {41, 22};
static const int CODES128[] = {41, 22, 24, 14};   // 14: the pair form (attention_pair.h)

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "check";
  bool all_ok = true;
  if (mode == "check") {
    const Case cases[] = {
        {256, 2, 2, 96, 100, 2, 0, 0},  {256, 3, 2, 33, 257, 3, 0, 0},   {256, 2, 4, 160, 200, 0, 0, 0},
        {256, 2, 2, 130, 64, 1, 1, 0},  {256, 2, 2, 128, 31, 3, 0, 0},   {256, 2, 1, 64, 96, 0, 0, 1},
        {256, 16, 4, 256, 800, 2, 0, 0}, {256, 16, 4, 800, 256, 2, 0, 0}, {256, 2, 4, 300, 1100, 2, 1, 0},
        {128, 2, 2, 96, 100, 2, 0, 0},  {128, 3, 2, 33, 257, 3, 0, 0},   {128, 2, 4, 160, 200, 0, 0, 0},
        {128, 2, 2, 130, 64, 1, 1, 0},  {128, 2, 2, 128, 31, 3, 0, 0},   {128, 16, 4, 256, 800, 2, 0, 0},
        {128, 16, 4, 800, 800, 3, 0, 0}, {128, 2, 4, 520, 2048, 2, 1, 0}, {128, 1, 4, 64, 5000, 2, 0, 0},
    };
    for (const Case& c : cases) {
      Buffers b;
      make(c, b, 1234u + c.Sq + 7 * c.Sk);
      printf("form %d  B%d H%d Sq%d Sk%d mask%d spike%d qmask%d\n", c.form, c.B, c.H, c.Sq, c.Sk, c.maskmode, c.spike, c.qmask);
      const int* codes = c.form == 256 ? CODES256 : CODES128;
      const int ncodes = c.form == 256 ? 2 : 4;
      for (int i = 0; i <= ncodes; ++i) {
        const int code = i < ncodes ? codes[i] : 0;
        bmhrl_attention_config(c.form, code);
        CK(hipMemset(b.dout, 0xff, (size_t)c.B * c.Sq * b.ldo * 2));
        const int rc = launch(c, b, 0);
        CK(hipDeviceSynchronize());
        if (rc != 0) { printf("  split %d: launch rc %d\n", code, rc); all_ok = false; continue; }
        char tag[64];
        snprintf(tag, sizeof tag, "split %d", code);
        all_ok &= verify(c, b, tag);
      }
      release(b);
    }
    printf(all_ok ? "ALL OK\n" : "FAILURES\n");
    return all_ok ? 0 : 1;
  }
  if (mode == "time") {
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const Case cases[] = {
        {128, 16, 4, 256, 800, 1, 0, 0}, {128, 16, 4, 256, 800, 2, 0, 0}, {128, 16, 4, 800, 800, 1, 0, 0},
        {128, 16, 4, 800, 800, 2, 0, 0}, {256, 16, 4, 800, 256, 1, 0, 0}, {256, 16, 4, 800, 256, 2, 0, 0},
        {256, 16, 4, 256, 256, 2, 0, 0}, {256, 16, 4, 256, 800, 1, 0, 0},
        {128, 8, 4, 1024, 2048, 2, 0, 0}, {128, 8, 4, 2048, 2048, 2, 0, 0}, {256, 8, 4, 2048, 1024, 2, 0, 0},
        {256, 8, 4, 1024, 1024, 2, 0, 0},
    };
    for (const Case& c : cases) {
      Buffers b;
      make(c, b, 99u);
      const double gf = 4.0 * c.B * c.H * (double)c.Sq * c.Sk * c.form / 1e9;
      printf("form %d B%d H%d Sq%d Sk%d mask%d (%.2f GF executed):", c.form, c.B, c.H, c.Sq, c.Sk, c.maskmode, gf);
      const int* codes = c.form == 256 ? CODES256 : CODES128;
      const int ncodes = c.form == 256 ? 2 : 4;
      for (int i = 0; i < ncodes; ++i) {
        bmhrl_attention_config(c.form, codes[i]);
        if (launch(c, b, 0) != 0) { printf("  [%d] n/a", codes[i]); continue; }
        CK(hipDeviceSynchronize());
        const double us = time_case(c, b, iters);
        printf("  [%d] %.1f us %.0f TF/s %.1f%%", codes[i], us, gf / us * 1e3, gf / us * 1e3 / 2500 * 100);
      }
      printf("\n");
      fflush(stdout);
      release(b);
    }
    return 0;
  }
  if (mode == "one" && argc >= 9) {
    Case c{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[8]), 0, 0};
    const int code = atoi(argv[7]);
    const int iters = argc > 9 ? atoi(argv[9]) : 20;
    Buffers b;
    make(c, b, 99u);
    bmhrl_attention_config(c.form, code);
    if (launch(c, b, 0) != 0) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    const double gf = 4.0 * c.B * c.H * (double)c.Sq * c.Sk * c.form / 1e9;
    const double us = time_case(c, b, iters);
    printf("form %d B%d H%d Sq%d Sk%d mask%d split %d: %.2f us  %.0f TF/s executed (%.1f %% of 2.5 PF)\n", c.form, c.B, c.H, c.Sq,
           c.Sk, c.maskmode, code, us, gf / us * 1e3, gf / us * 1e3 / 2500 * 100);
    release(b);
    return 0;
  }
  if (mode == "bwdcheck" || mode == "bwdtime") {
    // fused backward of the shared-128 form: forward (statistics, Cx) -> delta -> bmhrl_attention_shared128_bwd
    const bool timing = mode == "bwdtime";
    const Case small[] = {{128, 2, 2, 96, 100, 2, 0, 0}, {128, 3, 2, 33, 257, 3, 0, 0}, {128, 2, 4, 160, 200, 0, 0, 0},
                          {128, 2, 2, 130, 64, 1, 1, 0}, {128, 8, 4, 256, 300, 2, 0, 0}, {128, 2, 4, 290, 260, 3, 0, 0}};
    const Case big[] = {{128, 16, 4, 256, 800, 2, 0, 0}, {128, 16, 4, 800, 800, 2, 0, 0}, {128, 8, 4, 1024, 2048, 2, 0, 0}};
    const Case* cases = timing ? big : small;
    const int ncases = timing ? 3 : 6;
    for (int ci = 0; ci < ncases; ++ci) {
      const Case c = cases[ci];
      Buffers b;
      make(c, b, 4321u + c.Sq);
      const size_t nq = (size_t)c.B * c.Sq * b.ldq;
      std::vector<uint16_t> dcx(nq);
      std::mt19937 rng(77);
      std::normal_distribution<float> nd(0.f, 1.f);
      for (auto& x : dcx) x = f2bf(nd(rng));
      uint16_t *d_dcx, *d_dq;
      float *d_delta, *d_dx, *d_ws;
      CK(hipMalloc(&d_dcx, nq * 2));
      CK(hipMalloc(&d_dq, nq * 2));
      CK(hipMemcpy(d_dcx, dcx.data(), nq * 2, hipMemcpyHostToDevice));
      CK(hipMalloc(&d_delta, (size_t)c.B * c.H * c.Sq * 4));
      CK(hipMalloc(&d_dx, (size_t)c.B * c.Sk * 128 * 4));
      CK(hipMalloc(&d_ws, (size_t)bmhrl_attention_shared128_bwd_workspace(c.B, c.H, c.Sk) * 4));
      const float scale = 1.f / 16.f;
      const uint8_t* m = c.maskmode == 0 ? nullptr : b.dmask;
      auto run = [&](hipStream_t st) {
        bmhrl_attn_delta(d_dcx, b.ldq, b.dout, b.ldo, d_delta, 1.f, c.B, c.H, c.Sq, 128, st);
        return bmhrl_attention_shared128_bwd(b.dq, b.ldq, b.dk, b.ldk, d_dcx, b.ldq, b.dmax, b.dsum, d_delta, m, c.Sk, d_dq, b.ldq,
                                             d_dx, 128, 0, d_ws, c.B, c.H, c.Sq, c.Sk, scale, st);
      };
      bmhrl_attention_config(128, 0);
      if (launch(c, b, 0) != 0 || run(0) != 0) { printf("launch failed\n"); return 1; }
      CK(hipDeviceSynchronize());
      if (timing) {
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        std::vector<float> ms;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < 10; ++i) run(st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          ms.push_back(t / 10);
        }
        std::sort(ms.begin(), ms.end());
        const double gf = 14.0 * c.B * c.H * (double)c.Sq * c.Sk * 128 / 1e9;     // 7 products
        printf("bwd shared128 B%d H%d Sq%d Sk%d: %.1f us (delta + dq + dx + reduce), %.0f TF/s executed\n", c.B, c.H, c.Sq, c.Sk,
               ms[2] * 1e3, gf / (ms[2] * 1e3) * 1e3);
        continue;
      }
      // fp64 reference of dQp and dX
      std::vector<uint16_t> dq(nq);
      std::vector<float> dx((size_t)c.B * c.Sk * 128);
      CK(hipMemcpy(dq.data(), d_dq, nq * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(dx.data(), d_dx, dx.size() * 4, hipMemcpyDeviceToHost));
      double wq = 0, rq = 0, wx = 0, rx = 0;
      std::vector<double> S(c.Sk), P(c.Sk), dP(c.Sk), dXr((size_t)c.Sk * 128);
      for (int bb = 0; bb < c.B; ++bb) {
        std::fill(dXr.begin(), dXr.end(), 0.0);
        for (int h = 0; h < c.H; ++h)
          for (int q = 0; q < c.Sq; ++q) {
            const uint16_t* qp = &b.q[((size_t)bb * c.Sq + q) * b.ldq + (size_t)h * 128];
            const uint16_t* dp = &dcx[((size_t)bb * c.Sq + q) * b.ldq + (size_t)h * 128];
            double mx = -1e300;
            for (int k = 0; k < c.Sk; ++k) {
              const uint16_t* xp = &b.k[((size_t)bb * c.Sk + k) * b.ldk];
              double s0 = 0, d0 = 0;
              for (int d = 0; d < 128; ++d) { s0 += (double)bf2f(qp[d]) * bf2f(xp[d]); d0 += (double)bf2f(dp[d]) * bf2f(xp[d]); }
              const bool keep = c.maskmode == 0 || b.mask[(size_t)bb * c.Sk + k];
              S[k] = keep ? s0 * scale : -1e9;
              dP[k] = d0;
              mx = std::max(mx, S[k]);
            }
            double l = 0;
            for (int k = 0; k < c.Sk; ++k) { P[k] = std::exp(S[k] - mx); l += P[k]; }
            double delta = 0;
            for (int k = 0; k < c.Sk; ++k) { P[k] /= l; delta += P[k] * dP[k]; }
            double dqr[128] = {0};
            for (int k = 0; k < c.Sk; ++k) {
              const bool keep = c.maskmode == 0 || b.mask[(size_t)bb * c.Sk + k];
              const double ds = keep ? P[k] * (dP[k] - delta) * scale : 0.0;      // no gradient through masked_fill
              const uint16_t* xp = &b.k[((size_t)bb * c.Sk + k) * b.ldk];
              for (int d = 0; d < 128; ++d) {
                dqr[d] += ds * bf2f(xp[d]);
                dXr[(size_t)k * 128 + d] += P[k] * bf2f(dp[d]) + ds * bf2f(qp[d]);
              }
            }
            for (int d = 0; d < 128; ++d) {
              const double got = bf2f(dq[((size_t)bb * c.Sq + q) * b.ldq + (size_t)h * 128 + d]);
              wq = std::max(wq, std::fabs(got - dqr[d]));
              rq = std::max(rq, std::fabs(dqr[d]));
            }
          }
        for (size_t i = 0; i < dXr.size(); ++i) {
          wx = std::max(wx, std::fabs((double)dx[(size_t)bb * c.Sk * 128 + i] - dXr[i]));
          rx = std::max(rx, std::fabs(dXr[i]));
        }
      }
      const bool ok = wq <= 2e-2 * rq && wx <= 2e-2 * rx && std::isfinite(wq) && std::isfinite(wx);
      printf("bwd shared128 B%d H%d Sq%d Sk%d mask%d spike%d: dQp err %.2e of max %.2e, dX err %.2e of max %.2e  %s\n", c.B, c.H,
             c.Sq, c.Sk, c.maskmode, c.spike, wq, rq, wx, rx, ok ? "ok" : "FAIL");
      all_ok &= ok;
      (void)hipFree(d_dcx); (void)hipFree(d_dq); (void)hipFree(d_delta); (void)hipFree(d_dx); (void)hipFree(d_ws);
      release(b);
    }
    if (!timing) printf(all_ok ? "ALL OK\n" : "FAILURES\n");
    return all_ok ? 0 : 1;
  }
  fprintf(stderr, "usage: attn_bench check | time [iters] | one form B H Sq Sk code maskmode [iters] | bwdcheck | bwdtime\n");
  return 2;
}
