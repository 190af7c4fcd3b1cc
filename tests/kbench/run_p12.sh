R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/p12
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/benchtrace -- python3 $R/bench.py --no-cpu-baseline > $OUT/benchtrace.log 2>&1 || tail -3 $OUT/benchtrace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -- python3 $R/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline > $OUT/eager.log 2>&1 || tail -3 $OUT/eager.log
cd $R
python3 tests/probes/step_sequence.py $OUT/benchtrace > $OUT/step_sequence.txt 2> $OUT/step_sequence.err
python3 profiles/summarize.py $OUT/eager 10 > $OUT/eager_stats.md 2> $OUT/eager_stats.err
head -5 $OUT/step_sequence.txt; head -30 $OUT/eager_stats.md
rm -rf $OUT/benchtrace $OUT/eager
