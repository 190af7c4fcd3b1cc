"""Builds bmhrl_amd/csrc/libbmhrl_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libbmhrl_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "attention128.hip", "attention128p.hip", "attention_bwd256.hip", "attention_fwd_sk256.hip", "attention_f16.hip", "attention128_f16.hip", "small_attention.hip",
           "memory_attention.hip", "elementwise.hip", "loss.hip", "critic.hip", "conv_gn.hip"]
HEADERS = ["common.h", "attention_fwd.h", "attention_bwd.h", "attention_pair.h"]
# attention.hip: the eight 16-register O^T accumulators are loop-carried vector PHIs; AMDGPUCodeGenPrepare would break
# them into 128 scalar (VGPR) PHIs, i.e. 128 accumulator<->VGPR copies per key tile around the MFMAs.
# gemm.hip: same for the MFMA tile accumulators of the main loop (1-3 % on the large shapes).
_VECTOR_PHIS = ["-mllvm", "-amdgpu-codegenprepare-break-large-phis=false"]
# attention128.hip (256 registers per wave, two waves per SIMD): MFMA results in arch VGPRs -- with no accumulator-register
# operand in the file the compiler treats the 256 registers as one file (see the file's header).
EXTRA_FLAGS = {"attention.hip": _VECTOR_PHIS, "gemm.hip": _VECTOR_PHIS, "attention_f16.hip": _VECTOR_PHIS,
               "attention128p.hip": _VECTOR_PHIS,
               "attention128.hip": _VECTOR_PHIS + ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "attention128_f16.hip": _VECTOR_PHIS + ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    deps.append(os.path.join(os.path.dirname(CSRC), "..", "include", "bmhrl_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c",
               os.path.join(CSRC, s), "-o", o] + EXTRA_FLAGS.get(s, [])
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
