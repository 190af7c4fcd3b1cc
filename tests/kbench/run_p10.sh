mkdir -p gpurun_out/p10
( for lib in "" "tests/kbench/build/twopass"; do for e in 0 1 2; do for sh in "4096 1024 1024 0 0 1 0" "12800 128 1024 0 0 1 0" "480 300 1024 0 0 1 0" "480 1024 1024 0 0 1 0"; do
  echo "lib[${lib:-default}] $(LD_LIBRARY_PATH=$lib timeout -k 10 60 tests/kbench/gemm_bench one $sh 30 $e 2>&1 | tail -1)"; done; done; done ) > gpurun_out/p10/epi.log 2>&1
cat gpurun_out/p10/epi.log
timeout -k 10 600 python -m pytest tests/test_decode_gpu.py tests/test_kernels_gpu.py tests/test_streams_gpu.py tests/test_agent_gpu.py -x -q > gpurun_out/p10/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/p10/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/p10/bench.log 2>&1; tail -c 300 gpurun_out/p10/bench.log | head -3
BMHRL_ORDERED_DX=0 python bench.py --no-cpu-baseline > gpurun_out/p10/bench_atomdx.log 2>&1
BMHRL_GEMM_W8=0 python bench.py --no-cpu-baseline > gpurun_out/p10/bench_w0.log 2>&1
python bench.py --no-cpu-baseline --no-exploration > gpurun_out/p10/bench_noexp.log 2>&1
for f in bench bench_atomdx bench_w0 bench_noexp; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/p10/$f.log').read().strip().split('\n')[-1]); print('$f', d['value'], d['ms_per_step'])
except Exception as e: print('$f', 'failed', e)
PY
done
