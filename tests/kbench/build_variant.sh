#!/bin/bash
# A second build of the library with extra compile flags on the attention files (tuning aid: A/B of kernel variants in one GPU
# session).  usage: tests/kbench/build_variant.sh NAME "-DBMHRL_ATTN_STAGGER=0"  ->  bmhrl_amd/csrc/variants/NAME/libbmhrl_hip.so
# (use it with BMHRL_HIP_LIB=... for the Python side, LD_LIBRARY_PATH=.../variants/NAME for tests/kbench/attn_bench)
set -e
cd "$(dirname "$0")/../.."
python -m bmhrl_amd.build > /dev/null
C=bmhrl_amd/csrc
mkdir -p $C/variants/$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -mllvm -amdgpu-codegenprepare-break-large-phis=false $2"
/opt/rocm/bin/hipcc $F -c $C/attention.hip -o $C/variants/$1/attention.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -c $C/attention128.hip -o $C/variants/$1/attention128.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$1/libbmhrl_hip.so $C/gemm.o $C/variants/$1/attention.o \
  $C/variants/$1/attention128.o $C/attention_f16.o $C/attention128_f16.o $C/elementwise.o $C/loss.o $C/critic.o
ls -la $C/variants/$1/libbmhrl_hip.so
