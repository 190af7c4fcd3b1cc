#!/bin/bash
# Builds the native kernel harnesses of tests/kbench (tuning aids; see the .cpp headers).
set -e
cd "$(dirname "$0")/../.."
python -m bmhrl_amd.build > /dev/null
HIPCC=/opt/rocm/bin/hipcc
$HIPCC -O2 -std=c++17 tests/kbench/attn_bench.cpp -o tests/kbench/attn_bench -Lbmhrl_amd/csrc -lbmhrl_hip -Wl,-rpath,'$ORIGIN/../../bmhrl_amd/csrc' 2>&1 | grep -v hip-link || true
$HIPCC -O2 -std=c++17 tests/kbench/gemm_bench.cpp -o tests/kbench/gemm_bench -Lbmhrl_amd/csrc -lbmhrl_hip -Wl,-rpath,'$ORIGIN/../../bmhrl_amd/csrc' 2>&1 | grep -v hip-link || true
if [ "$1" = "trace" ]; then   # the attention kernels with cycle stamps (-DBMHRL_ATTN_TRACE), linked statically into a second binary
  mkdir -p tests/kbench/build
  F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-comment -DBMHRL_ATTN_TRACE -mllvm -amdgpu-codegenprepare-break-large-phis=false"
  $HIPCC $F -c bmhrl_amd/csrc/attention.hip -o tests/kbench/build/attention_t.o &
  $HIPCC $F -mllvm -amdgpu-mfma-vgpr-form -c bmhrl_amd/csrc/attention128.hip -o tests/kbench/build/attention128_t.o &
  $HIPCC $F -c bmhrl_amd/csrc/attention128p.hip -o tests/kbench/build/attention128p_t.o &
  $HIPCC $F -c bmhrl_amd/csrc/attention_fwd_sk256.hip -o tests/kbench/build/attention_fwd_sk256_t.o &
  $HIPCC $F -DBMHRL_GEMM_TRACE -c bmhrl_amd/csrc/gemm.hip -o tests/kbench/build/gemm_t.o &
  wait
  $HIPCC -O2 -std=c++17 -c tests/kbench/gemm_bench.cpp -o tests/kbench/build/gemm_bench.o
  $HIPCC --offload-arch=gfx950 tests/kbench/build/gemm_bench.o tests/kbench/build/gemm_t.o -o tests/kbench/gemm_bench_trace
  $HIPCC -O2 -std=c++17 -c tests/kbench/attn_bench.cpp -o tests/kbench/build/attn_bench.o
  $HIPCC --offload-arch=gfx950 tests/kbench/build/attn_bench.o tests/kbench/build/attention_t.o tests/kbench/build/attention128_t.o tests/kbench/build/attention128p_t.o tests/kbench/build/attention_fwd_sk256_t.o -o tests/kbench/attn_bench_trace
fi
