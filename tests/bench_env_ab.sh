#!/bin/bash
# Tuning aid: bench.py under several environment settings, interleaved, one GPU session.
# usage: tests/bench_env_ab.sh "NAME=VAL ..." "NAME=VAL ..."   (an empty string = the defaults)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
  for e in "$@"; do
    ms=$(env $e python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$e] pass $rep: $ms ms/step"
  done
done
