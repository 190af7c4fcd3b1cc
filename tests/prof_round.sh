#!/bin/bash
# Collects the per-round profile artefacts on the GPU box (copy the summaries into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the attention kernels alone, of the eager step and of the bench command itself
#   2. counter-only passes (separate runs, no trace domains) of the two cross-modal attention launches: MFMA busy cycles,
#      wave cycles, waits; FETCH_SIZE / WRITE_SIZE of the V<-A launch
#   3. the default bench.py line
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ATTN_SHAPE=va
run() { d=$1; shift; "$@" > $OUT/$d.log 2>&1 || { tail -3 $OUT/$d.log; exit 1; }; }
run attn128 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn128 -- python3 $R/tests/bench_one_attn128.py
SQ=800 SK=256 run attn256 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/attn256 -- python3 $R/tests/bench_one_attn.py
run fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tests/bench_one_attn128.py
run write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tests/bench_one_attn128.py
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS"
run pmc128a rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc128a -- python3 $R/tests/bench_one_attn128.py
run pmc128b rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc128b -- python3 $R/tests/bench_one_attn128.py
SQ=800 SK=256 run pmc256a rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc256a -- python3 $R/tests/bench_one_attn.py
SQ=800 SK=256 run pmc256b rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc256b -- python3 $R/tests/bench_one_attn.py
run eager rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -- python3 $R/bench.py --eager --steps 8 --warmup 2 --no-cpu-baseline
# the bench command itself under the tracer (graph mode): its last phase launches the roofline kernel in isolation
run benchtrace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/benchtrace -- python3 $R/bench.py --no-cpu-baseline
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
python3 bench.py --mode rl --no-cpu-baseline > $OUT/bench_rl.json 2> $OUT/bench_rl.err || { tail -3 $OUT/bench_rl.err; exit 1; }
python3 profiles/bench_trace_summary.py $OUT/benchtrace $OUT/benchtrace.log > $OUT/bench_kernel_stats.md
python3 profiles/summarize.py $OUT/eager 10 > $OUT/eager_stats.md
python3 - <<PY > $OUT/attn_summary.txt
import csv, glob, collections
for name in ("attn128", "attn256"):
    f = glob.glob("$OUT/" + name + "/**/*_kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:90], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        print(name, k, "calls", len(v), "avg us %.2f" % (sum(v) / len(v)), "min %.2f" % min(v))
for name in ("fetch", "write", "pmc128a", "pmc128b", "pmc256a", "pmc256b"):
    tot = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd" in r["Kernel_Name"]:
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(tot.items()):
        print(name, k, "dispatches", len(v), "mean %.0f" % (sum(v) / len(v)))
PY
cat $OUT/attn_summary.txt
