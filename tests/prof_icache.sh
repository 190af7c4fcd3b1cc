#!/bin/bash
# Instruction-fetch counters of the short kernels (run on the GPU box; counters only, separate passes): does the code a wave
# executes ONCE per launch (prologues, epilogues, unrolled loop copies) stall on instruction-cache misses?
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i "icache\|ifetch\|inst_cache\|SQC_" $OUT/counters.txt | head -60 > $OUT/icache_counters.txt
run() {   # name, counter set, command...
  name=$1; shift; set=$1; shift
  rocprofv3 --pmc $set --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; }
}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_WAVES"; do
  i=$((i+1))
  run gemm_f32_$i "$set" $R/tests/kbench/gemm_bench one 4096 1024 1024 0 0 1 0 5
  run gemm_bf16_$i "$set" $R/tests/kbench/gemm_bench one 4096 1024 1024 0 0 1 1 5
  run attn14_$i "$set" $R/tests/kbench/attn_bench one 128 16 4 256 800 14 2 5
  run attn22_$i "$set" $R/tests/kbench/attn_bench one 128 16 4 256 800 22 2 5
done
python3 - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$OUT/*_[12]")):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in sorted(tot): print(f"{os.path.basename(d):14s} {k[0]:60s} {k[1]:28s} {tot[k]/n[k]:14.0f} ({n[k]})")
PY
