// The "pair" form of the head-dimension-128 shared-key/value attention forward (attention_pair.h): a translation unit of its
// own because it is built WITHOUT -amdgpu-mfma-vgpr-form (attention128.hip has it): the kernel names the register class of
// every MFMA operand itself (O^T and Q'^T in the accumulator half of the 512-register file, S^T in arch VGPRs).
#include "attention_pair.h"

bool bmhrl_attn128_pair_ok(int B, int H, int Sq, int Sk) { return pair128_ok(B, H, Sq, Sk); }

int bmhrl_attn128_pair_fwd(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo, float* row_max,
                           float* row_sum, const uint8_t* mask, int64_t mask_sb, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                           float scale, hipStream_t stream) {
  if (!pair128_ok(B, H, Sq, Sk)) return -22;
  const hipError_t e = launch_pair128(Qp, ldq, X, ldx, ctx, ldo, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, g_attn_dbg,
                                      stream);
  attn_trace_dump("attn128 pair", Sq, Sk, stream, 12, 8);   // stamps 9 .. 12: inside the second iteration
  return hip_status(e);
}
