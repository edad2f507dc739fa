#!/usr/bin/env python3
"""Per kernel of a rocprofv3 --kernel-trace run: workgroups, threads, VGPRs (arch + accumulation), LDS, scratch, launches and us per step - what
decides which kernels of two streams can share a CU.    python tools/kernel_resources.py <dir with *_kernel_trace.csv> [steps] [min us per step]"""
import collections
import csv
import glob
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bot::", "")[-56:]
        key = (name, wg, int(r.get("VGPR_Count", 0) or 0), int(r.get("Accum_VGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0), int(r.get("Scratch_Size", 0) or 0))
        agg[key].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid // max(wg, 1)))
print(f"{'kernel':56s} {'threads':>7s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scratch':>7s} {'wgs':>7s} {'n/step':>6s} {'avg us':>8s} {'us/step':>8s}")
for (k, wg, v, a, l, s), rows in sorted(agg.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
    tot = sum(t for t, _ in rows)
    if tot / steps >= min_us:
        print(f"{k:56s} {wg:7d} {v:5d} {a:5d} {l:7d} {s:7d} {max(g for _, g in rows):7d} {len(rows) / steps:6.1f} {tot / len(rows):8.1f} {tot / steps:8.1f}")
