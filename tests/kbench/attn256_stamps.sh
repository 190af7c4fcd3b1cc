#!/bin/bash
# cycle stamps of the d_k = 256 forward kernel at the A<-V shape (4 x 1 waves) and the video-self shape (2 x 2)
bash tests/kbench/build.sh trace > /dev/null 2>&1
mkdir -p gpurun_out/q7
BMHRL_ATTN_TRACE=1 timeout -k 10 120 tests/kbench/attn_bench_trace one 256 16 4 800 256 41 2 3 > gpurun_out/q7/av.log 2>&1; echo "rc=$?"; cat gpurun_out/q7/av.log
BMHRL_ATTN_TRACE=1 timeout -k 10 120 tests/kbench/attn_bench_trace one 256 16 4 256 256 22 2 3 > gpurun_out/q7/vv.log 2>&1; echo "rc=$?"; cat gpurun_out/q7/vv.log
