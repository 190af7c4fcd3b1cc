// Head-dimension-128 attention forward in the absorbed-projection form (bmhrl_attention_shared128_fwd: V<-A cross
// attention and audio self attention -- every attention whose keys / values are the 128-wide audio stream).  Same kernel
// template as attention.hip (attention_fwd.h); a file of its own because it is built with the VGPR form of the MFMAs
// (-mllvm -amdgpu-mfma-vgpr-form) and without any accumulator-register operand: 256 registers per wave (two waves per SIMD)
// are then ONE file to the compiler -- the S^T accumulator is read by the softmax's VALU instructions in place (no
// v_accvgpr_read per score) and O^T needs no copies around its MFMAs.
#include "attention_bwd.h"

// attention128p.hip
bool bmhrl_attn128_pair_ok(int B, int H, int Sq, int Sk);
int bmhrl_attn128_pair_fwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo, float* row_max,
                           float* row_sum, const uint8_t* mask, int64_t mask_sb, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                           float scale, hipStream_t stream);

namespace {
int g_cfg128 = 0;   // (QW, KW) split: 0 = automatic
}  // namespace

void bmhrl_attn128_set_cfg(int code) { g_cfg128 = code; }

extern "C" int bmhrl_attention_shared128_fwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo,
                                             float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                                             int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale,
                                             bmhrl_stream_t stream) {
  return attention128_entry(Qp, ldq, X, ldx, ctx, ldo, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, g_cfg128,
                            (hipStream_t)stream, bmhrl_attn128_pair_fwd, bmhrl_attn128_pair_ok);
}

extern "C" int64_t bmhrl_attention_shared128_bwd_workspace(int32_t B, int32_t H, int32_t Sk) {
  return (int64_t)B * H * Sk * 128;       // fp32 elements: the per-head partials of dX
}

extern "C" int bmhrl_attention_shared128_bwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, const void* dCx, int64_t lddo,
                                             const float* row_max, const float* row_sum, const float* delta,
                                             const uint8_t* mask, int64_t mask_sb, void* dQp, int64_t lddq, float* dX,
                                             int64_t lddx, int32_t accumulate_dx, float* workspace, int32_t B, int32_t H,
                                             int32_t Sq, int32_t Sk, float scale, bmhrl_stream_t stream) {
  constexpr int DK = 128;
  BMHRL_CHECK_ARG(Qp && X && dCx && row_max && row_sum && delta && dQp);
  BMHRL_CHECK_ARG(B > 0 && H > 0 && Sq > 0 && Sk > 0);
  BMHRL_CHECK_ARG(ldq % 8 == 0 && ldx % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0);
  BMHRL_CHECK_ARG(ldq >= (int64_t)H * DK && ldx >= DK && lddo >= (int64_t)H * DK && lddq >= (int64_t)H * DK);
  BMHRL_CHECK_ARG((((uintptr_t)Qp | (uintptr_t)X | (uintptr_t)dCx | (uintptr_t)dQp) & 15) == 0);
  BMHRL_CHECK_ARG((int64_t)Sk * ldx * 2 < (1ll << 31));
  BMHRL_CHECK_ARG(dX == nullptr || (workspace != nullptr && lddx >= DK && lddx % 4 == 0 && ((uintptr_t)dX & 15) == 0));
  AttnBwdArgs a;
  a.Q = (const bf16_t*)Qp; a.ldq = ldq; a.X = (const bf16_t*)X; a.ldx = ldx; a.dO = (const bf16_t*)dCx; a.lddo = lddo;
  a.dQ = (bf16_t*)dQp; a.lddq = lddq; a.dXp = workspace;
  a.row_max = row_max; a.row_sum = row_sum; a.delta = delta; a.mask = mask; a.mask_sb = mask_sb;
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  // dQp: the forward's query-block split (4 x 1 with two workgroups per CU when there are enough 32-row blocks, else 2 x 2)
  const int64_t rows32 = (int64_t)B * H * ((Sq + 31) / 32);
  const bool wide = rows32 >= 1536;
  const int qw = wide ? 4 : 2;
  if (Sk > AttnCfg<DK, 4, 1, 4, true>::MAX_SK) return -22;
  a.q_tiles = (Sq + 32 * qw - 1) / (32 * qw);
  BMHRL_CHECK_ARG((int64_t)B * H * a.q_tiles * H * a.q_tiles < (1ll << 32));
  a.per_b = H * a.q_tiles;
  a.map_mode = (B % 8 == 0) ? 0 : 2;
  a.magic_perb = div_magic((unsigned)a.per_b);
  a.magic_qt = div_magic((unsigned)a.q_tiles);
  a.magic_h = div_magic((unsigned)H);
  dim3 grid((unsigned)(B * H * a.q_tiles));
  if (wide) hipLaunchKernelGGL((attn_bwd_dq128_kernel<4, 1, 4>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_dq128_kernel<2, 2, 4>), grid, dim3(256), 0, s, a);
  if (dX != nullptr) {
    const int kblocks = (Sk + 32 * DX_WAVES - 1) / (32 * DX_WAVES);
    hipLaunchKernelGGL((attn_bwd_dx128_kernel<4>), dim3((unsigned)(B * H * kblocks)), dim3(64 * DX_WAVES), 0, s, a);
    const long total = (long)B * Sk * 32;
    hipLaunchKernelGGL(attn_bwd_dx_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)workspace, dX,
                       (long)lddx, B, H, Sk, accumulate_dx);
  }
  return hip_status(hipGetLastError());
}
