// IEEE-half operand variant of the head-dimension-128 (absorbed-projection, shared key / value image) attention forward; built
// like attention128.hip (VGPR form of the MFMAs, no accumulator-register operand in the file).  See attention_f16.hip.
#define BMHRL_F16_OPERANDS 1
#include "attention_fwd.h"

extern "C" int bmhrl_attention_shared128_fwd_f16(const void* Qp, int64_t ldq, const void* X, int64_t ldx, void* ctx, int64_t ldo,
                                                 float* row_max, float* row_sum, const uint8_t* mask, int64_t mask_sb,
                                                 int32_t B, int32_t H, int32_t Sq, int32_t Sk, float scale,
                                                 bmhrl_stream_t stream) {
  return attention128_entry(Qp, ldq, X, ldx, ctx, ldo, row_max, row_sum, mask, mask_sb, B, H, Sq, Sk, scale, 0,
                            (hipStream_t)stream);
}
