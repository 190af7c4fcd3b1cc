#!/bin/bash
mkdir -p gpurun_out/q8
timeout -k 10 400 python -m pytest tests/test_detr_gpu.py tests/test_detr_agent_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -3
for m in 0 2 0 2 1; do
  BMHRL_FUSED_MEMATTN=$m timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q8/bench_$m.log 2>&1; echo "memattn=$m rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/q8/bench_$m.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("memattn=$m", d["value"], d["ms_per_step"])
PY
done
