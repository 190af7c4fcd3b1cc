#!/bin/bash
# XCD-aware tile order of the direct-to-LDS GEMM kernel (BMHRL_GEMM_XCD): correctness, isolated timings, the step
mkdir -p gpurun_out/q12
BMHRL_GEMM_XCD=1 timeout -k 10 200 tests/kbench/gemm_bench check > gpurun_out/q12/check.log 2>&1; echo "check rc=$?"; tail -2 gpurun_out/q12/check.log
for x in 0 1; do BMHRL_GEMM_XCD=$x timeout -k 10 200 tests/kbench/gemm_bench time > gpurun_out/q12/time_$x.log 2>&1; done
paste -d'|' <(cut -c1-80 gpurun_out/q12/time_0.log) <(cut -c62-92 gpurun_out/q12/time_1.log)
for x in 0 1 0 1; do
  BMHRL_GEMM_XCD=$x timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q12/bench_$x.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/q12/bench_$x.log") if x.startswith("{")]
print("xcd=$x", json.loads(l[-1])["ms_per_step"] if l else "FAILED")
PY
done
