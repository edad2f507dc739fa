#!/usr/bin/env python3
"""Kernels of a rocprofv3 --kernel-trace run whose grid is smaller than the chip (fewer workgroups than CUs): such a launch is bound by what ITS
few CUs can pull (about 65 GB/s per CU from the L2), not by the chip - round 6 found a 12-workgroup second-stage kernel taking 146 us for 10 MB.
    python tools/small_grids.py <dir with *_kernel_trace.csv> [steps]"""
import collections
import csv
import glob
import sys

steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("bot::", "")[-60:], grid // max(wg, 1), wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':60s} {'workgroups':>10s} {'threads':>7s} {'launches/step':>13s} {'avg us':>8s} {'us/step':>8s}")
for (k, nwg, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if nwg < 256 and sum(v) / steps >= 4:
        print(f"{k:60s} {nwg:10d} {wg:7d} {len(v) / steps:13.1f} {sum(v) / len(v):8.1f} {sum(v) / steps:8.1f}")
