#!/bin/bash
# in-step sweep of the GEMM tile-plan thresholds (one box)
mkdir -p gpurun_out/q10
run() { name=$1; shift; ( export "$@"; timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/q10/$name.log 2>&1 ); python - <<PY
import json
l=[x for x in open("gpurun_out/q10/$name.log") if x.startswith("{")]
print("$name", (json.loads(l[-1])["ms_per_step"] if l else "FAILED"))
PY
}
run base X=1
run bigmin128 BMHRL_GEMM_BIGMIN=128
run bigmin192 BMHRL_GEMM_BIGMIN=192
run bigmin400 BMHRL_GEMM_BIGMIN=400
run base2 X=1
run midmax256 BMHRL_GEMM_MIDMAX=256
run stages3 BMHRL_GEMM_STAGES=3
run base3 X=1
