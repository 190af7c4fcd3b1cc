#!/bin/bash
mkdir -p gpurun_out/q6
timeout -k 10 600 python -m pytest tests/test_split_backward_gpu.py -x -q > gpurun_out/q6/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/q6/pytest.log
grep -q "failed\|error" gpurun_out/q6/pytest.log && exit 1
for e in 1 0 1 0; do
  BMHRL_EARLY_ADAM=$e timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q6/bench_$e.log 2>&1; echo "early=$e rc=$?"
  python - <<PY
import json
l=[x for x in open("gpurun_out/q6/bench_$e.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("early=$e", d["value"], d["ms_per_step"])
PY
done
