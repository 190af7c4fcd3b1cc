mkdir -p gpurun_out/p9
step() { name=$1; shift; timeout -k 10 "$@" > gpurun_out/p9/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]; }
step check 180 tests/kbench/gemm_bench check && \
step time 200 tests/kbench/gemm_bench time 20 && \
step pytest 1000 python -m pytest tests -m gpu -x -q && \
step bench 300 python bench.py
tail -3 gpurun_out/p9/pytest.log; tail -c 600 gpurun_out/p9/bench.log
