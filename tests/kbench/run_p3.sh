mkdir -p gpurun_out/p3
step() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/p3/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
step check 180 tests/kbench/attn_bench check && \
step time 120 tests/kbench/attn_bench time 30 && \
( timeout -k 10 60 env BMHRL_ATTN_TRACE=1 tests/kbench/attn_bench_trace one 128 16 4 256 800 14 2 3 2>&1 | tail -4 > gpurun_out/p3/trace.log; true ) && \
step pytest 1000 python -m pytest tests -m gpu -x -q
tail -4 gpurun_out/p3/pytest.log
