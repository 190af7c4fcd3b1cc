#!/bin/bash
# per-launch durations of the critic's wavefront kernel (which launches are heavy), tests/bench_critic.py under the tracer
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/critic; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tests/bench_critic.py > $OUT/log 2>&1 || { tail -3 $OUT/log; exit 1; }
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/t/**/*_kernel_trace.csv", recursive=True)[0])))
for name, n in (("rnn_wave_mfma", 35), ("rnn_wave_kernel", 45)):   # the final replay of the chunk-1 (matrix pipe) / chunk-3 (VALU) form
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if name in r["Kernel_Name"])
    if len(ev) < n:
        continue
    last = ev[-n:]
    d = [(e - s) / 1e3 for s, e in last]
    g = [(last[i + 1][0] - last[i][1]) / 1e3 for i in range(n - 1)]
    print(name, "durations us:", " ".join(f"{x:.1f}" for x in d))
    print(name, "gaps us     :", " ".join(f"{x:.1f}" for x in g))
    print(name, "sum dur %.0f us, sum gaps %.0f us" % (sum(d), sum(g)))
PY
