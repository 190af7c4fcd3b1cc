mkdir -p gpurun_out/p13
timeout -k 10 600 python -m pytest tests/test_detr_agent_gpu.py tests/test_detr_gpu.py tests/test_exploration_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/p13/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/p13/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/p13/bench.log 2>&1
python - <<PY
import json
d=json.loads(open('gpurun_out/p13/bench.log').read().strip().split('\n')[-1]); print('bench', d['value'], d['ms_per_step'])
PY
