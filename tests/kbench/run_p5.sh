mkdir -p gpurun_out/p5
step() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/p5/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
step check_w8 180 env BMHRL_GEMM_W8=2 tests/kbench/gemm_bench check && \
step check 180 tests/kbench/gemm_bench check && \
step time_w0 200 env BMHRL_GEMM_W8=0 tests/kbench/gemm_bench time 20 && \
step time_w1 200 env BMHRL_GEMM_W8=1 tests/kbench/gemm_bench time 20 && \
step time_w2 200 env BMHRL_GEMM_W8=2 tests/kbench/gemm_bench time 20 && \
( for sh in "4096 1024 1024 0 0 1 0" "4096 1024 1024 0 0 1 1" "4096 1024 3072 0 1 1 0"; do timeout -k 10 60 env BMHRL_GEMM_TRACE=1 tests/kbench/gemm_bench_trace one $sh 3 2>&1 | tail -3; done > gpurun_out/p5/gemm_trace.log; true )
