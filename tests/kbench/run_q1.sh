#!/bin/bash
# round 4: fused score backward of the d_k = 256 attentions -- parity, timing, then the whole suite and the bench line
mkdir -p gpurun_out/q1
timeout -k 10 300 python -m pytest tests/test_attn_bwd256_gpu.py -x -q > gpurun_out/q1/new.log 2>&1; echo "new rc=$?"
tail -5 gpurun_out/q1/new.log
grep -q "failed\|error" gpurun_out/q1/new.log && exit 1
timeout -k 10 200 python tests/bench_attn_bwd256.py > gpurun_out/q1/time.log 2>&1; echo "time rc=$?"
cat gpurun_out/q1/time.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/q1/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/q1/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/q1/bench.log 2>&1; echo "bench rc=$?"
tail -c 1500 gpurun_out/q1/bench.log
BMHRL_FUSED_SCORES_BWD=0 timeout -k 10 300 python bench.py > gpurun_out/q1/bench_off.log 2>&1; echo "bench_off rc=$?"
tail -c 600 gpurun_out/q1/bench_off.log
