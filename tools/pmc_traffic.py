#!/usr/bin/env python3
"""tools/pmc_bench.sh's last step: the per-launch L2<->fabric traffic of the kernel a bench line's roofline object names.

    python tools/pmc_traffic.py <workload> <pmc summary csv> <log of a bench run (its last line = the JSON)> <round tag>

Prints the entry for profiles/spmm_traffic.json[<workload>]: FETCH_SIZE (KB, doubled: gfx950 tallies 128-byte fabric requests as
64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KB) of the launches of that kernel, averaged per launch, and the L2 hit rate."""
import csv
import json
import sys

wl, summary, log, tag = sys.argv[1:5]
line = [l for l in open(log).read().splitlines() if l.startswith("{")][-1]
kernel = json.loads(line)["roofline"]["kernel"].split(" (")[0]
norm = lambda k: k.replace(" ", "")
rows = {r["counter"]: r for r in csv.DictReader(open(summary)) if norm(r["kernel"]) == norm(kernel)}
f, w = float(rows["FETCH_SIZE"]["avg_per_launch"]), float(rows["WRITE_SIZE"]["avg_per_launch"])
hit, miss = float(rows["TCC_HIT_sum"]["avg_per_launch"]), float(rows["TCC_MISS_sum"]["avg_per_launch"])
print(json.dumps({
    "kernel": kernel, "workload": json.loads(line)["config"]["workload"],
    "source": f"profiles/{tag}_pmc_bench_{wl}.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate "
              f"passes over `bench.py --workload {wl}`, tools/pmc_bench.sh)",
    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches": int(rows["FETCH_SIZE"]["launches"]),
    "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests as 64 B (MI355X_MICROARCH.md 'HBM'); doubled",
    "bytes_per_launch": int((2 * f + w) * 1024), "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4), "round": tag}, indent=1))
