#!/bin/bash
# attn_bench variants with loop ablations compiled in (-DBMHRL_ABL=<mask>, see attention_fwd.h): timing aids, wrong results.
# usage: tests/kbench/build_abl.sh "0 1 2 4 8 15" ["extra hipcc flags"]
set -e
cd "$(dirname "$0")/../.."
HIPCC=/opt/rocm/bin/hipcc
mkdir -p tests/kbench/build
$HIPCC -O2 -std=c++17 -c tests/kbench/attn_bench.cpp -o tests/kbench/build/attn_bench.o
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-comment -mllvm -amdgpu-codegenprepare-break-large-phis=false $2"
for m in $1; do
  ( $HIPCC $F -DBMHRL_ABL=$m -c bmhrl_amd/csrc/attention.hip -o tests/kbench/build/attention_a$m.o &
    $HIPCC $F -DBMHRL_ABL=$m -mllvm -amdgpu-mfma-vgpr-form -c bmhrl_amd/csrc/attention128.hip -o tests/kbench/build/attention128_a$m.o &
    wait
    $HIPCC --offload-arch=gfx950 tests/kbench/build/attn_bench.o tests/kbench/build/attention_a$m.o tests/kbench/build/attention128_a$m.o -o tests/kbench/build/attn_abl_$m ) &
done
wait
ls -la tests/kbench/build/attn_abl_*
