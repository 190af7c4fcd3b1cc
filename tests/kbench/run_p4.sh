mkdir -p gpurun_out/p4
step() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/p4/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
step check 180 tests/kbench/attn_bench check && \
step time 120 tests/kbench/attn_bench time 30 && \
( for sh in "4096 1024 1024 0 0 1 0" "4096 1024 1024 0 0 1 1" "4096 3072 1024 0 0 1 1" "4096 1024 3072 0 1 1 0" "1024 1024 4096 1 1 1 0" "4096 4096 4096 0 0 1 1"; do timeout -k 10 60 env BMHRL_GEMM_TRACE=1 tests/kbench/gemm_bench_trace one $sh 3 2>&1 | tail -3; done > gpurun_out/p4/gemm_trace.log; true )
