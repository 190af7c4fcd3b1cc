// IEEE-half operand variant of the head-dimension-256 attention forward (BASELINE configs[4]: "fp16/bf16 MFMA cross-attention"):
// the kernel template of attention_fwd.h compiled with BMHRL_F16_OPERANDS -- v_mfma_f32_32x32x16_f16, P and the output rounded
// to half instead of bfloat16, everything else (fp32 scores, softmax statistics, masks, dropout, launch shapes) identical.
#define BMHRL_F16_OPERANDS 1
#include "attention_fwd.h"

extern "C" int bmhrl_attention_fwd_f16(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                       void* O, int64_t ldo, float* row_max, float* row_sum, const uint8_t* mask,
                                       int64_t mask_sb, int64_t mask_sq, int32_t B, int32_t H, int32_t Sq, int32_t Sk,
                                       int32_t dk, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                       bmhrl_stream_t stream) {
  return attention256_entry(Q, ldq, K, ldk, V, ldv, O, ldo, row_max, row_sum, mask, mask_sb, mask_sq, B, H, Sq, Sk, dk, scale,
                            dropout_p, seed, seed_dev, 0, (hipStream_t)stream);
}
