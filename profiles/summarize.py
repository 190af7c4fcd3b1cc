#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into the per-kernel table kept under profiles/.
usage: summarize.py <dir with *_kernel_stats.csv / *_kernel_trace.csv> <steps in the run> [> profiles/rNN_xxx.md]"""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
stats = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in stats)
print(f"total kernel time {tot / 1e6:.2f} ms over {steps:g} steps = {tot / 1e6 / steps:.3f} ms/step\n")
print("| kernel | calls/step | avg us | ms/step | % |\n|---|---|---|---|---|")
for r in stats[:40]:
    print(f"| {r['Name'][:100]} | {float(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.1f} |")
tr = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
if tr:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        n = r["Kernel_Name"]
        if "gemm_kernel" in n or "gemm_glds_kernel" in n or "attn_" in n:
            if "gemm_glds_kernel" in n:
                short = "gemm_glds" + n.split("gemm_glds_kernel")[1].split("(")[0]
            elif "gemm_kernel" in n:
                short = "gemm" + n.split("gemm_kernel")[1].split("(")[0]
            else:
                short = "attn_" + n.split("attn_")[1].split("(")[0][:40]
            agg[(short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("\n| kernel | blocks x | y | z | calls/step | avg us | ms/step |\n|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:50]:
        print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {len(v) / steps:.1f} | {sum(v) / len(v):.1f} | {sum(v) / steps / 1e3:.3f} |")
