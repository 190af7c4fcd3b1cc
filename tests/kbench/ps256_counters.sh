#!/bin/bash
# counters of attn_bwd_ps256_kernel alone (tests/bench_attn_bwd256.py): kernel trace, HBM traffic, MFMA / wave cycles -- separate passes
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ps256; rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tests/bench_attn_bwd256.py > $OUT/trace.log 2>&1 || { tail -3 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tests/bench_attn_bwd256.py > $OUT/fetch.log 2>&1 || { tail -3 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tests/bench_attn_bwd256.py > $OUT/write.log 2>&1 || { tail -3 $OUT/write.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $OUT/pmc -- python3 $R/tests/bench_attn_bwd256.py > $OUT/pmc.log 2>&1 || { tail -3 $OUT/pmc.log; exit 1; }
cd $R
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "ps256" in r["Kernel_Name"]:
        agg[r["Grid_Size_X"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print("trace grid", k, "calls", len(v), "avg us %.2f" % (sum(v) / len(v)), "min %.2f" % min(v))
for name in ("fetch", "write", "pmc"):
    tot = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ps256" in r["Kernel_Name"]:
                tot[(r["Counter_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    for k, v in sorted(tot.items()):
        print(name, k, "dispatches", len(v), "mean %.0f" % (sum(v) / len(v)))
PY
cat $OUT/summary.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/pmc
