"""ctypes binding of bmhrl_amd/csrc/libbmhrl_hip.so (C ABI: include/bmhrl_hip.h).

The library is required: loading raises if it is missing (no CPU or eager fallback exists).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BMHRL_HIP_LIB: load another build of the same library (kernel A/B comparisons in one run); never a different backend
LIB_PATH = os.environ.get("BMHRL_HIP_LIB") or os.path.join(_HERE, "csrc", "libbmhrl_hip.so")

i32, i64, u64, f32 = C.c_int32, C.c_int64, C.c_uint64, C.c_float
ptr = C.c_void_p


class RnnLayer(C.Structure):
    """bmhrl_rnn_layer of include/bmhrl_hip.h (field for field)"""
    _fields_ = [("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p),
                ("in_seq", C.c_void_p), ("in_ld", C.c_int64), ("in_dim", C.c_int32), ("gates", C.c_int32),
                ("seq_out", C.c_void_p), ("h", C.c_void_p * 2), ("c", C.c_void_p * 2),
                ("arelu_alpha", C.c_void_p), ("arelu_beta", C.c_void_p), ("xproj", C.c_void_p)]


class FusionTailParams(C.Structure):
    """bmhrl_fusion_tail_params of include/bmhrl_hip.h"""
    _fields_ = [(n, C.c_void_p) for n in ("gamma_ca", "beta_ca", "gamma_cv", "beta_cv", "a_v",
                                          "dgamma_ca", "dbeta_ca", "dgamma_cv", "dbeta_cv", "da_v")]


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32), ("batch1", i32), ("batch2", i32),
        ("A", ptr), ("lda", i64), ("a_sb1", i64), ("a_sb2", i64), ("a_trans", i32),
        ("B", ptr), ("ldb", i64), ("b_sb1", i64), ("b_sb2", i64), ("b_trans", i32),
        ("C", ptr), ("ldc", i64), ("c_sb1", i64), ("c_sb2", i64),
        ("Cb", ptr), ("ldcb", i64), ("cb_sb1", i64), ("cb_sb2", i64),
        ("epilogue", i32), ("alpha", f32), ("relu", i32), ("accumulate", i32), ("allow_split_k", i32),
        ("bias", ptr),
        ("residual", ptr), ("ldr", i64), ("r_sb1", i64), ("r_sb2", i64),
        ("mask", ptr), ("mask_sb1", i64), ("mask_sm", i64),
        ("rowvec", ptr), ("rowvec2", ptr), ("rv_sb1", i64), ("rv_sb2", i64),
        ("aux", ptr), ("ldaux", i64), ("aux_sb1", i64), ("aux_sb2", i64),
        ("dropout_p", f32), ("seed", u64), ("drop_sb1", i64), ("drop_sb2", i64), ("drop_sm", i64), ("seed_dev", ptr),
        ("colsum", ptr), ("colsum_sb2", i64), ("bias_sb2", i64), ("colsum_sb1", i64), ("bias_sb1", i64),
        ("split_ws", ptr), ("split_ws_elems", i64),
    ]


# name -> argtypes (every entry point of include/bmhrl_hip.h; tests check the .so exports all of them)
PROTOTYPES = {
    "bmhrl_gemm": [C.POINTER(GemmDesc), ptr],
    "bmhrl_gemm_group": [C.POINTER(GemmDesc), i32, ptr],
    "bmhrl_attention_fwd": [ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, i32, i32, i32, i32, f32,
                            f32, u64, ptr, ptr],
    "bmhrl_attention_shared128_fwd": [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i32, i32, i32, i32, f32, ptr],
    "bmhrl_attention_fwd_f16": [ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, i32, i32, i32, i32, f32,
                                f32, u64, ptr, ptr],
    "bmhrl_attention_shared128_fwd_f16": [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i32, i32, i32, i32, f32, ptr],
    "bmhrl_attention_config": [i32, i32],
    "bmhrl_attention_shared128_bwd": [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, i64, i32, ptr, i32, i32,
                                      i32, i32, f32, ptr],
    "bmhrl_small_attention_fwd": [ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i64, i32, i32, i32, i32, i32, f32, f32,
                                  u64, ptr, ptr],
    "bmhrl_small_attention_bwd": [ptr, i64, ptr, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr,
                                  ptr, i64, i64, i32, i32, i32, i32, i32, f32, ptr],
    "bmhrl_cast_memory": [ptr, ptr, ptr, i32, i32, i32, i32, ptr],
    "bmhrl_memory_attention": [i32, ptr, i64, ptr, i64, ptr, i64, i32, ptr, i64, i64, ptr, i64, ptr, i64, i32, i32, i32, i32, i32,
                               i32, f32, ptr],
    "bmhrl_softmax_rows": [ptr, i64, ptr, i64, i64, i32, i32, i64, ptr],
    "bmhrl_softmax_bwd_rows": [ptr, i64, ptr, i64, ptr, i64, i64, i32, f32, ptr, i64, i64, i32, i32, i32, i64, ptr],
    "bmhrl_attn_delta": [ptr, i64, ptr, i64, ptr, f32, i32, i32, i32, i32, ptr],
    "bmhrl_attention_bwd_scores256_ok": [i32, i32, i32, i64],
    "bmhrl_attention_bwd_scores256": [ptr, i64, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, i32, i32, i32, i32,
                                      f32, ptr],
    "bmhrl_layernorm_fwd": [ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_layernorm_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_layernorm_bwd_ws": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr, i64, ptr],
    "bmhrl_layernorm_fwd_groups": [ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i32, i32, ptr],
    "bmhrl_layernorm_bwd_groups": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, i32, ptr],
    "bmhrl_add_posenc": [ptr, ptr, ptr, ptr, ptr, i64, i32, i32, i32, f32, u64, ptr, ptr],
    "bmhrl_embed_posenc": [ptr, ptr, f32, ptr, ptr, ptr, ptr, i32, i32, i32, f32, f32, u64, ptr, ptr],
    "bmhrl_embed_bwd": [ptr, ptr, f32, ptr, ptr, i32, i32, i32, f32, ptr],
    "bmhrl_cast_bf16": [ptr, i64, ptr, i64, i64, i32, f32, f32, u64, ptr, ptr],
    "bmhrl_cast_split3_bf16": [ptr, i64, ptr, i64, i64, i32, i64, i32, ptr, i64, i32, ptr],
    "bmhrl_cast_colsum_bf16": [ptr, i64, ptr, i64, i64, i32, f32, f32, u64, ptr, ptr, ptr],
    "bmhrl_cast_colsum_bf16_groups": [ptr, i64, ptr, i64, i64, i32, f32, f32, u64, ptr, ptr, i64, i64, ptr],
    "bmhrl_cast_segments": [ptr, i32, i32, ptr],
    "bmhrl_colsum_bf16": [ptr, i64, ptr, i32, i64, i32, ptr],
    "bmhrl_colsum_bf16_groups": [ptr, i64, ptr, i64, i32, i32, i64, ptr],
    "bmhrl_cast_bf16_copies": [ptr, i64, ptr, i64, i64, i32, i32, i64, ptr],
    "bmhrl_gate_fwd": [ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "bmhrl_gate_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_fusion_tail_fwd": [ptr, ptr, ptr, i32, i64, i32, ptr, ptr, ptr, i64, ptr],
    "bmhrl_fusion_tail_bwd": [ptr, ptr, i64, i64, ptr, ptr, ptr, ptr, i32, i64, i32, ptr, ptr, ptr],
    "bmhrl_expand_goals_index": [ptr, ptr, i32, i32, ptr],
    "bmhrl_gather_rows": [ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "bmhrl_expand_goals": [ptr, ptr, ptr, ptr, ptr, i64, i32, i32, i32, ptr],
    "bmhrl_expand_goals_explore": [ptr, ptr, ptr, ptr, ptr, i64, i32, i32, i32, f32, f32, u64, ptr, ptr, ptr],
    "bmhrl_unfold1d_bf16": [ptr, ptr, i64, i32, i32, i32, i32, i32, ptr],
    "bmhrl_fold1d": [ptr, i64, ptr, i32, i32, i32, i32, i32, ptr],
    "bmhrl_groupnorm_fwd": [ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, f32, ptr],
    "bmhrl_groupnorm_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr],
    "bmhrl_scatter_add_rows": [ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_log_softmax": [ptr, i64, i64, i32, ptr],
    "bmhrl_smooth_kl_fwd": [ptr, i64, ptr, ptr, ptr, ptr, f32, i32, i32, ptr, ptr, i64, i32, ptr],
    "bmhrl_smooth_kl_full": [ptr, i64, ptr, ptr, ptr, ptr, f32, i32, i32, ptr, i64, i32, ptr],
    "bmhrl_smooth_kl_bwd": [ptr, i64, ptr, ptr, ptr, ptr, f32, i32, i32, ptr, ptr, i32, ptr, i64, ptr, i64, i32, ptr],
    "bmhrl_token_loss_reduce": [ptr, ptr, i64, i64, ptr, f32, ptr, ptr, ptr],
    "bmhrl_log_softmax_bwd": [ptr, ptr, i64, ptr, i64, i64, i32, ptr],
    "bmhrl_head_loss": [ptr, i64, ptr, f32, i32, ptr, f32, ptr, ptr, ptr, ptr, i64, ptr, i64, i32, ptr],
    "bmhrl_smooth_kl_amp_grad": [ptr, i64, ptr, ptr, ptr, ptr, f32, i32, i32, ptr, i64, i32, ptr],
    "bmhrl_sample_tokens": [ptr, i64, ptr, ptr, i64, i32, i32, u64, ptr, i64, ptr],
    "bmhrl_reinforce_fwd": [ptr, i64, i32, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_reinforce_bwd": [ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr],
    "bmhrl_gemm_f32": [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i32, i32, i32, ptr],
    "bmhrl_rnn_step": [i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr],
    "bmhrl_rnn_wavefront": [ptr, i32, i32, i32, i32, i32, ptr],
    "bmhrl_critic_head": [ptr, ptr, ptr, f32, ptr, ptr, i64, i32, ptr],
    "bmhrl_adam_step": [ptr, ptr, ptr, ptr, i64, f32, f32, f32, f32, f32, i32, ptr, f32, ptr],
    "bmhrl_make_masks": [ptr, i64, ptr, i64, ptr, i32, i32, i32, i32, i64, i32, ptr, ptr, ptr, ptr],
    "bmhrl_batch_head": [ptr, i64, ptr, i64, ptr, i64, i32, i32, i32, i32, i64, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "bmhrl_adam_segments": [ptr, i32, i32, ptr, ptr, ptr, ptr, f32, f32, f32, f32, f32, i32, ptr, f32, ptr],
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m bmhrl_amd.build` (hipcc --offload-arch=gfx950). "
            "bmhrl_amd has no CPU fallback.")
    # One HIP runtime per process: the stream handles passed in come from torch, so the library must bind to the
    # libamdhip64 torch itself uses (same SONAME as /opt/rocm's; whichever loads first wins).  Load torch's first.
    import torch
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.bmhrl_layernorm_bwd_workspace.argtypes = [i64, i32]
    lib.bmhrl_layernorm_bwd_workspace.restype = C.c_int64
    lib.bmhrl_attention_shared128_bwd_workspace.argtypes = [i32, i32, i32]
    lib.bmhrl_attention_shared128_bwd_workspace.restype = C.c_int64
    lib.bmhrl_attention_max_keys.restype = C.c_int
    lib.bmhrl_memory_attention_ok.argtypes = [i32, i32, i32]
    lib.bmhrl_memory_attention_ok.restype = C.c_int
    lib.bmhrl_small_attention_ok.argtypes = [i32, i32, i32]
    lib.bmhrl_small_attention_ok.restype = C.c_int
    lib.bmhrl_gemm_splits.argtypes = [i32, i32, i32, i32]
    lib.bmhrl_gemm_splits.restype = C.c_int
    lib.bmhrl_hip_arch.restype = C.c_char_p
    lib.bmhrl_hip_abi_version.restype = C.c_int
    lib.bmhrl_deterministic_enabled.restype = C.c_int
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "invalid argument" if rc < 0 else "hipError_t"
        raise HipError(f"{what} failed: {kind} {rc}")
