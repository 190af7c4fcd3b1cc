"""Tuning aid: the kernels of ONE captured training step in start order, from a rocprofv3 --kernel-trace CSV of bench.py
(queue, start us, duration us, kernel, grid in workgroups), plus launches per kernel name.
usage: python tests/probes/step_sequence.py <dir with *_kernel_trace.csv> > sequence.txt"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("at::native::", "")[:64]


idx = [i for i, r in enumerate(rows) if "adam_segments" in r["Kernel_Name"]]
gaps = [idx[i + 1] - idx[i] for i in range(len(idx) - 1)]
n = collections.Counter(gaps).most_common(1)[0][0]
i0 = [i for i in range(len(idx) - 1) if gaps[i] == n][3]
step = rows[idx[i0] + 1:idx[i0 + 1] + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"# {n} kernels per step; queues: {dict(collections.Counter(r['Queue_Id'] for r in step))}")
names = collections.Counter()
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    g = tuple(int(r[f"Grid_Size_{a}"]) // int(r[f"Workgroup_Size_{a}"]) for a in "XYZ")
    names[short(r["Kernel_Name"])] += 1
    print(f"{r['Queue_Id']:>2} {s:8.1f} {e - s:6.1f} {short(r['Kernel_Name']):64s} {g}")
print("# launches per kernel")
for k, c in names.most_common():
    print(f"# {c:4d}  {k}")
