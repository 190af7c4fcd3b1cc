#!/bin/bash
# Tuning aid: bench.py against several builds of the library (bmhrl_amd/csrc/variants/NAME.so), interleaved, one GPU session.
# usage: tests/bench_step_ab.sh NAME [NAME ...]
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for n in "$@"; do
    ms=$(BMHRL_HIP_LIB=$R/bmhrl_amd/csrc/variants/$n.so python $R/bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$n pass $rep: $ms ms/step"
  done
done
