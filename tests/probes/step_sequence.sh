R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/seq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/log 2>&1 || { tail -3 $OUT/log; exit 1; }
cd $R && python3 tests/probes/step_sequence.py $OUT/t > $OUT/sequence.txt && head -1 $OUT/sequence.txt && grep -n "head_loss" -B3 -A3 $OUT/sequence.txt | head -30
rm -rf $OUT/t
