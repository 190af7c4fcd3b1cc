"""CPU checks of the C-ABI boundary: the library builds, loads and exports every symbol include/bmhrl_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bmhrl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmhrl_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from bmhrl_amd import build, _lib
    build.build(verbose=False)
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bmhrl_hip.h but not exported"
    assert set(_lib.PROTOTYPES) | {"bmhrl_hip_arch", "bmhrl_hip_abi_version", "bmhrl_deterministic_enabled", "bmhrl_layernorm_bwd_workspace",
                                    "bmhrl_attention_shared128_bwd_workspace", "bmhrl_attention_max_keys", "bmhrl_gemm_splits", "bmhrl_small_attention_ok",
                                    "bmhrl_memory_attention_ok"} == set(syms)
    assert lib.bmhrl_layernorm_bwd_workspace(4096, 1024) == 256 * 2 * 1024      # 4 rows per wave, 4 waves per block: 256 blocks
    assert lib.bmhrl_hip_arch() == b"gfx950"
    assert lib.bmhrl_hip_abi_version() == 17
    import os
    from bmhrl_amd import ops
    assert lib.bmhrl_deterministic_enabled() == int(os.environ.get("BMHRL_DETERMINISTIC", "0") not in ("", "0")) == int(ops.deterministic())
    assert lib.bmhrl_attention_max_keys() == 10112        # pure host query: the fused kernels' key limit
    # pure host query too: the video projections' weight gradients store every element once, the caption-side ones split K
    assert lib.bmhrl_gemm_splits(1024, 1024, 4096, 1) == 1 and lib.bmhrl_gemm_splits(128, 300, 480, 1) > 1
    assert lib.bmhrl_gemm_splits(0, 4, 4, 1) < 0
    assert lib.bmhrl_small_attention_ok(30, 30, 256) == 1 and lib.bmhrl_small_attention_ok(30, 33, 256) == 0
    assert lib.bmhrl_memory_attention_ok(30, 800, 128) == 1 and lib.bmhrl_memory_attention_ok(30, 1024, 128) == 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from bmhrl_amd import ops
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cast_bf16(x, 8, torch.zeros(4, 8, dtype=torch.bfloat16), 8, 4, 8)
