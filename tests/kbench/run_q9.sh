#!/bin/bash
mkdir -p gpurun_out/q9
export BMHRL_FUSED_MEMATTN=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/q9/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/q9/pytest.log
bash tests/probes/step_sequence.sh > gpurun_out/q9/seq.log 2>&1; echo "seq rc=$?"; head -1 gpurun_out/seq/sequence.txt
