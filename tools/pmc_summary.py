#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter set) into one small per-kernel table.

    python tools/pmc_summary.py gpurun_out/pmc_* > profiles/rNN_pmc_microbench.csv
"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bot::" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "launches", "avg_per_launch"])
for (k, c), v in sorted(agg.items()):
    w.writerow([k, c, len(v), f"{sum(v) / len(v):.6g}"])
