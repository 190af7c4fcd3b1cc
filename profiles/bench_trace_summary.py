#!/usr/bin/env python3
"""`rocprofv3 --kernel-trace --stats -- python3 bench.py` (the bench command itself, graph mode) -> the table kept under
profiles/: per-kernel totals of the whole command, and the launches of the roofline kernel in bench.py's own timed
region (attention_roofline: after the last optimizer kernel), whose average must agree with roofline.launch_us.
usage: bench_trace_summary.py <trace dir> <stdout log of that run>"""
import collections, csv, glob, json, sys
d, log = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows)
line = [l for l in open(log) if l.startswith('{"metric"')][-1]
b = json.loads(line)
print(f"bench line of this run: {b['ms_per_step']:.3f} ms/step under the tracer (graph launches become host-bound: the nodes are "
      f"submitted one by one), roofline.launch_us {b['roofline']['launch_us']:.2f}\n")
last_adam = max(i for i, e in enumerate(ev) if "adam_" in e[2])
tail = collections.defaultdict(list)
for s, e, n, blk in ev[last_adam + 1:]:
    if "attn_fwd_kernel" in n or "gemm_kernel" in n:
        tail[(n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), blk)].append((e - s) / 1e3)
print("| roofline phase of bench.py (isolated launches after the last step) | workgroups | launches | avg us | min us |\n|---|---|---|---|---|")
for (n, blk), v in sorted(tail.items(), key=lambda kv: -sum(kv[1])):
    print(f"| {n} | {blk} | {len(v)} | {sum(v) / len(v):.2f} | {min(v):.2f} |")
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, blk in ev:
    k = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
    tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
allt = sum(v[1] for v in tot.values())
print(f"\n| kernel (whole command, {len(ev)} launches, {allt / 1e3:.1f} ms of kernel time) | launches | avg us | total ms | % |\n|---|---|---|---|---|")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"| {k} | {c} | {t / c:.1f} | {t / 1e3:.2f} | {100 * t / allt:.1f} |")
