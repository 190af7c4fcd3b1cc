mkdir -p gpurun_out/p1
step() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/p1/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
step check 180 tests/kbench/attn_bench check && \
step time 120 tests/kbench/attn_bench time 30 && \
step trace 60 env BMHRL_ATTN_TRACE=1 tests/kbench/attn_bench_trace one 128 16 4 256 800 14 2 3 && \
step trace22 60 env BMHRL_ATTN_TRACE=1 tests/kbench/attn_bench_trace one 128 16 4 256 800 22 2 3 && \
step pytest_expl 500 python -m pytest tests/test_exploration_gpu.py tests/test_attn128_gpu.py -x -q
