#!/bin/bash
# runs every capture pattern in its own process; a host segfault of one pattern does not stop the others
out=${1:-gpurun_out/probes.log}
: > "$out"
for p in ${PROBES:-flat two_sides reuse pool_wrap nested nested_idle nested_cross}; do
  echo "=== $p" >> "$out"
  LD_PRELOAD=$PWD/tests/probes/libsegv_bt.so timeout -k 5 120 python -X faulthandler tests/probes/capture_forks.py $p >> "$out" 2>&1
  rc=$?
  echo "rc=$rc" >> "$out"
  if [ $rc -eq 124 ]; then echo "timeout: stopping" >> "$out"; exit 1; fi
done
exit 0
