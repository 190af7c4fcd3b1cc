#!/bin/bash
# Collects the per-round profile artefacts on the GPU box (copy the summaries into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the attention kernels alone and of the eager step
#   2. FETCH_SIZE / WRITE_SIZE of the head-dim-128 attention launch (separate counter-only passes)
#   3. the default bench.py line
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn128 -- python3 $R/tests/bench_one_attn128.py > $OUT/attn128.log 2>&1 || { tail -3 $OUT/attn128.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn256 -- python3 $R/tests/bench_one_attn.py > $OUT/attn256.log 2>&1 || { tail -3 $OUT/attn256.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tests/bench_one_attn128.py > $OUT/fetch.log 2>&1 || { tail -3 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tests/bench_one_attn128.py > $OUT/write.log 2>&1 || { tail -3 $OUT/write.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -- python3 $R/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline > $OUT/eager.log 2>&1 || { tail -3 $OUT/eager.log; exit 1; }
# the bench command itself under the tracer (graph mode): its last phase launches the roofline kernel 55 times in isolation
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/benchtrace -- python3 $R/bench.py --no-cpu-baseline > $OUT/benchtrace.log 2>&1 || { tail -3 $OUT/benchtrace.log; exit 1; }
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
python3 profiles/bench_trace_summary.py $OUT/benchtrace $OUT/benchtrace.log > $OUT/bench_kernel_stats.md
python3 profiles/summarize.py $OUT/eager 10 > $OUT/eager_stats.md
python3 - <<PY
import csv, glob, collections
for name in ("attn128", "attn256"):
    f = glob.glob("$OUT/" + name + "/**/*_kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:70], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print(name, k, "calls", len(v), "avg us %.1f" % (sum(v) / len(v)), "min %.1f" % min(v))
for name in ("fetch", "write"):
    tot = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd" in r["Kernel_Name"]:
                tot[(r["Counter_Name"], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))].append(float(r["Counter_Value"]))
    for k, v in tot.items():
        print(name, k, "dispatches", len(v), "mean %.0f" % (sum(v) / len(v)))
PY
