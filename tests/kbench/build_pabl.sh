#!/bin/bash
# attn_bench variants of the PAIR kernel (attention_pair.h) with loop ablations compiled in (-DBMHRL_PABL=<mask>): timing aids,
# wrong results.  usage: tests/kbench/build_pabl.sh "0 1 2 4 8 16 32 64 128 255" ["extra hipcc flags"]
set -e
cd "$(dirname "$0")/../.."
HIPCC=/opt/rocm/bin/hipcc
mkdir -p tests/kbench/build
$HIPCC -O2 -std=c++17 -c tests/kbench/attn_bench.cpp -o tests/kbench/build/attn_bench.o
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-comment -mllvm -amdgpu-codegenprepare-break-large-phis=false -DBMHRL_ATTN_TRACE $2"
[ -f tests/kbench/build/attention_t.o ] || $HIPCC $F -c bmhrl_amd/csrc/attention.hip -o tests/kbench/build/attention_t.o
[ -f tests/kbench/build/attention128_t.o ] || $HIPCC $F -mllvm -amdgpu-mfma-vgpr-form -c bmhrl_amd/csrc/attention128.hip -o tests/kbench/build/attention128_t.o
for m in $1; do
  ( $HIPCC $F -DBMHRL_PABL=$m -c bmhrl_amd/csrc/attention128p.hip -o tests/kbench/build/attention128p_a$m.o
    $HIPCC --offload-arch=gfx950 tests/kbench/build/attn_bench.o tests/kbench/build/attention_t.o tests/kbench/build/attention128_t.o tests/kbench/build/attention128p_a$m.o -o tests/kbench/build/pair_abl_$m ) &
done
wait
ls tests/kbench/build/pair_abl_*
