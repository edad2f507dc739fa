#!/usr/bin/env python3
"""Derived figures for profiles/r06_blocked_bound.txt from tools/r06_blocked_bound.sh's outputs (pmc.csv: per-kernel counter averages per launch,
kernel_stats.csv: rocprofv3 --kernel-trace --stats of the same command).   python tools/r06_blocked_bound.py gpurun_out/r06/bound_reddit <kernel substring> ..."""
import csv
import sys

d, wanted = sys.argv[1], sys.argv[2:]
pmc = {}
for r in csv.DictReader(open(f"{d}/pmc.csv")):
    pmc.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_per_launch"])
dur = {r["Name"].split("(")[0].replace("void ", ""): (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(f"{d}/kernel_stats.csv"))}
for k, c in pmc.items():
    if not any(w in k for w in wanted):
        continue
    ns, calls = dur[k]
    t = ns * 1e-9
    req = c["TCP_TCC_READ_REQ_sum"]
    clk = c["GRBM_GUI_ACTIVE"] / 8 / t                                  # the counter sums the 8 XCDs
    cu_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256
    print(f"{k}: {calls} launches, {ns / 1e3:.1f} us per launch, clock {clk / 1e9:.2f} GHz")
    print(f"  vector loads: {c['SQ_INSTS_VMEM_RD']:.4g} wave instructions; L1 accesses per instruction {c['TCP_TOTAL_CACHE_ACCESSES_sum'] / c['SQ_INSTS_VMEM_RD']:.1f}; "
          f"L1 -> L2 read requests per instruction {req / c['SQ_INSTS_VMEM_RD']:.1f} (128-byte requests)")
    print(f"  L2 -> L1 delivery: {req * 128 / 1e9:.1f} GB per launch = {req * 128 / t / 1e12:.1f} TB/s = {req * 128 / t / 256 / 1e9:.1f} GB/s per CU = "
          f"{req * 128 / cu_cycles:.1f} B per clock and CU  (MI355X_MICROARCH.md 'Indexed rows': 66-73 GB/s per CU = 16.8-18.8 TB/s for L2-resident rows, "
          f"29-31 / 7.4-7.9 from a 151 MB table through the fabric)")
    print(f"  L2: {c['TCC_REQ_sum']:.4g} requests, hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}; fabric reads {c['TCC_EA0_RDREQ_sum'] * 128 / 1e9:.1f} GB per launch = "
          f"{c['TCC_EA0_RDREQ_sum'] * 128 / t / 1e12:.2f} TB/s (128-byte requests: TCC_EA0_RDREQ_32B = {c['TCC_EA0_RDREQ_32B_sum']:.0f})")
    wc = c["SQ_WAVE_CYCLES"]
    print(f"  waves: {c['SQ_WAVES']:.0f}; of their cycles parked in s_waitcnt / barriers {c['SQ_WAIT_ANY'] / wc:.2f}, issue stalls {c['SQ_WAIT_INST_ANY'] / wc:.2f}, "
          f"issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.2f}; VALU per vector load {c['SQ_INSTS_VALU'] / c['SQ_INSTS_VMEM_RD']:.1f}")
    print(f"  LDS: {c['SQ_INSTS_LDS']:.4g} instructions, array active {c['SQ_LDS_IDX_ACTIVE'] / cu_cycles:.3f} of the CU cycles, bank-conflict cycles "
          f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f} of those, LDS issue stalls {c['SQ_WAIT_INST_LDS'] / wc:.4f} of the wave cycles")
