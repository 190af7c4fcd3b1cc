#!/bin/bash
mkdir -p gpurun_out/q5
export PYTHONPATH=$PWD
for mode in noopt_side step; do
  timeout -k 10 200 python tests/probes/step_then_capture.py $mode > gpurun_out/q5/v_$mode.log 2>&1; echo "$mode rc=$?"
  grep -v "amdgpu.ids\|UserWarning\|run_backward\|Extension modules" gpurun_out/q5/v_$mode.log | grep "done\|captured\|^\[\|Error\|error" | tail -4
  rm -f core core.*
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/q5/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/q5/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q5/bench.log 2>&1; echo "bench rc=$?"; tail -c 400 gpurun_out/q5/bench.log | head -c 200
