"""Per-step kernel time of two rocprofv3 kernel_stats.csv files side by side (calls / N steps):  python tools/stats_diff.py a.csv b.csv --steps 28"""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("a")
ap.add_argument("b")
ap.add_argument("--steps", type=float, default=1.0, help="divide totals by this many steps (warm-up + timed + profile legs)")
ap.add_argument("--min-us", type=float, default=5.0)
args = ap.parse_args()


def load(p):
    out = {}
    for r in csv.DictReader(open(p)):
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("bot::", "")[-70:]
        out[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return out


A, B = load(args.a), load(args.b)
rows = []
for k in sorted(set(A) | set(B)):
    ca, ta = A.get(k, (0, 0.0))
    cb, tb = B.get(k, (0, 0.0))
    rows.append((tb / args.steps - ta / args.steps, k, ca, ta / args.steps, cb, tb / args.steps))
print(f"{'kernel':70s} {'calls a':>7s} {'us/step a':>10s} {'calls b':>7s} {'us/step b':>10s} {'b - a':>9s}")
for d, k, ca, ta, cb, tb in sorted(rows):
    if max(ta, tb) >= args.min_us:
        print(f"{k:70s} {ca:7d} {ta:10.1f} {cb:7d} {tb:10.1f} {d:9.1f}")
print(f"{'TOTAL':70s} {'':7s} {sum(v[1] for v in A.values()) / args.steps:10.1f} {'':7s} {sum(v[1] for v in B.values()) / args.steps:10.1f}")
