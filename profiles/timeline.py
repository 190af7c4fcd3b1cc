#!/usr/bin/env python3
"""Busy / idle / overlap of the GPU inside the timed steps of a `rocprofv3 --kernel-trace --output-format csv` run.
usage: timeline.py <dir with *_kernel_trace.csv> <steps at the end of the run to analyse>"""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# a step ends with the Adam kernel
ends = [i for i, e in enumerate(ev) if "adam_" in e[2]]
first = ends[-steps - 1] + 1
seg = ev[first:ends[-1] + 1]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy, cur_s, cur_e, total = 0, None, None, 0
for s, e, _ in seg:
    total += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{steps} steps: wall {(t1 - t0) / steps / 1e6:.3f} ms/step, GPU busy (union) {busy / steps / 1e6:.3f} ms/step, "
      f"sum of kernel durations {total / steps / 1e6:.3f} ms/step, kernels/step {len(seg) / steps:.0f}")
gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-8:]), seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1)), reverse=True)
print("largest gaps (us, after kernel -> before kernel):")
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:8.1f}  {a}  ->  {b}")

# per-queue busy time and the top kernels of the busiest queue (the critical path of the captured graph)
import collections
qs = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= t1:
        qs[r["Queue_Id"]].append((s, e, r["Kernel_Name"]))
print("per queue: kernels, busy ms/step")
for q, v in sorted(qs.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    print(f"  queue {q}: {len(v) / steps:.0f} kernels/step, {sum(e - s for s, e, _ in v) / steps / 1e6:.3f} ms/step")
q0 = max(qs.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))[1]
agg = collections.defaultdict(float)
for s, e, n in q0:
    agg[n.split("(")[0][-60:]] += (e - s) / steps / 1e6
for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {v:.3f} ms/step  {n}")
