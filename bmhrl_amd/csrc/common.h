// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16, LDS transposed reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit MFMA operand type.  Every translation unit but two uses bfloat16; attention_f16.hip / attention128_f16.hip define
// BMHRL_F16_OPERANDS and get the SAME attention kernels with IEEE half operands (BASELINE configs[4]: "fp16/bf16 MFMA
// cross-attention"): the kernels only move 16-bit elements, hand them to the MFMA of the matching type (BMHRL_MFMA16) and
// convert fp32 -> operand type where P and the output are formed, so the type names below are the only switch.
#ifdef BMHRL_F16_OPERANDS
typedef _Float16 bf16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define BMHRL_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define BMHRL_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define WAVE 64
#define NEG_MASK (-1e9f)  // reference: masked_fill(mask == False, -1e9), model/multihead_attention.py:22

// argument checks of the C ABI: -22 (EINVAL); BMHRL_DEBUG_ARGS=1 in the environment names the failed condition on stderr
#include <cstdio>
#include <cstdlib>
static inline void bmhrl_arg_fail(const char* cond, const char* file, int line) {
  static const bool verbose = getenv("BMHRL_DEBUG_ARGS") != nullptr;
  if (verbose) fprintf(stderr, "bmhrl_hip: invalid argument: (%s) is false at %s:%d\n", cond, file, line);
}
#define BMHRL_CHECK_ARG(cond)                      \
  do {                                             \
    if (!(cond)) {                                 \
      bmhrl_arg_fail(#cond, __FILE__, __LINE__);   \
      return -22;                                  \
    }                                              \
  } while (0)

static inline int hip_status(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// BMHRL_DETERMINISTIC=1: every sum whose order depends on the scheduling of fp32 atomics takes an ordered path instead --
// no K split in the GEMMs, column sums / LayerNorm parameter gradients / the gate's scalar through one owner per address,
// embedding and goal scatter gradients row by row.  Same arithmetic up to the order of additions; slower (the step measured
// below in DESIGN.md); two runs of the same step then agree bit for bit (tests/test_split_backward_gpu.py).
static inline bool bmhrl_deterministic() {
  static const bool on = getenv("BMHRL_DETERMINISTIC") != nullptr && atoi(getenv("BMHRL_DETERMINISTIC")) != 0;
  return on;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; `red` is a 16-float LDS scratch.  All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Counter-based RNG: one 32-bit hash per (seed, element id).  Forward and backward regenerate the same
// dropout mask from the element id, so no mask tensor is stored.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  return (hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}
// Dropout decisions need far less than the 64-bit mixer above (three 64-bit multiplies per element, in every GEMM / cast /
// attention epilogue of the training step): a 32-bit finaliser (two rounds of xorshift-multiply, the "lowbias32" constants)
// over (seed, element id).  Forward and backward call the same function, so the mask is reproduced from the element id.
__device__ __forceinline__ uint32_t dropout_bits(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  x ^= ((uint32_t)(idx >> 32) + (uint32_t)(seed >> 32)) * 0x85EBCA77u;
  x ^= x >> 16; x *= 0x21F0AAADu;
  x ^= x >> 15; x *= 0x735A2D97u;
  x ^= x >> 15;
  return x;
}
// keep-scale of inverted dropout: 0 if dropped, 1/(1-p) if kept
__device__ __forceinline__ float dropout_scale(float p, uint64_t seed, uint64_t idx) {
  const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967040.f);
  return dropout_bits(seed, idx) < thr ? 0.f : 1.0f / (1.0f - p);
}

__device__ __forceinline__ bf16x8 zero_bf16x8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
  return z;
}

// ds_read_b64_tr_b16: per 16-lane group, reads a 4-row x 16-column block of 16-bit elements; lane 4q+p of the
// group passes the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q.
__device__ __forceinline__ bf16x4 lds_read_tr4(const bf16_t* p) {
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 join8(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
