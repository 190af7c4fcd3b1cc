#!/bin/bash
# round 4: gradients produced in the flat bucket (data-parallel path) -- tests, then the phased step against the one-phase step
mkdir -p gpurun_out/q3
timeout -k 10 900 python -m pytest tests/test_split_backward_gpu.py tests/test_ddp_gpu.py tests/test_streams_gpu.py -x -q > gpurun_out/q3/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/q3/pytest.log
grep -q "failed\|error" gpurun_out/q3/pytest.log && exit 1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
for cfg in "one:0:0:" "pg1:1:0:" "pg_split:1:1:0" "pg_split_homes:1:1:1"; do
  IFS=: read name pg split homes <<< "$cfg"
  ( [ "$pg" = 1 ] && export BMHRL_BENCH_FORCE_PG=1; [ "$split" = 1 ] && export BMHRL_SPLIT_BACKWARD=1 BMHRL_DIRECT_GRADS=0; [ -n "$homes" ] && export BMHRL_GRAD_HOMES=$homes;
    timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q3/bench_$name.log 2>&1; echo "$name rc=$?" )
  python - <<PY
import json
l=[x for x in open("gpurun_out/q3/bench_$name.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$name", d["value"], d["ms_per_step"])
PY
done
