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
bench = json.loads(line)
kernel = bench["roofline"]["kernel"].split(" (")[0]
# One roofline "launch" = one call of the launch function, which the L2-blocked SpMM runs as several kernel launches (rounds of
# one resident wave of workgroups): counters are summed over the kernel launches of a call.  Calls in the profiled run = calls per
# step (from the bench line: launches_timed / steps) x steps run (timed + warm-up + the 3 extra steps that time the halves GEMMs).
calls_per_step = bench["roofline"]["launches_timed"] / bench["steps"]
# (+ the 3 extra steps that time optimizer.step(), round 4: every un-captured run has them)
steps_run = bench["steps"] + bench["warmup"] + (3 if bench["roofline"].get("dense_projections") else 0) + (3 if bench.get("optimizer_ms") is not None else 0)
calls = calls_per_step * steps_run
norm = lambda k: k.replace(" ", "")
rows = {r["counter"]: r for r in csv.DictReader(open(summary)) if norm(r["kernel"]) == norm(kernel)}
per_call = lambda c: float(rows[c]["avg_per_launch"]) * int(rows[c]["launches"]) / calls
f, w = per_call("FETCH_SIZE"), per_call("WRITE_SIZE")
hit, miss = per_call("TCC_HIT_sum"), per_call("TCC_MISS_sum")
print(json.dumps({
    "kernel": kernel, "workload": json.loads(line)["config"]["workload"],
    "source": f"profiles/{tag}_pmc_bench_{wl}.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate "
              f"passes over `bench.py --workload {wl}`, tools/pmc_bench.sh)",
    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "kernel_launches_profiled": int(rows["FETCH_SIZE"]["launches"]), "calls_profiled": calls,
    "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests as 64 B (MI355X_MICROARCH.md 'HBM'); doubled",
    "bytes_per_launch": int((2 * f + w) * 1024), "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4), "round": tag}, indent=1))
