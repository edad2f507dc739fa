"""Idle gaps of the main stream in ONE train step of a rocprofv3 --kernel-trace database (rocpd sqlite): every gap >= --min-us between the end of
a main-stream launch and the start of the next, with the kernels on both sides, and the totals.
    python tools/timeline_gaps.py results.db --step 8"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--step", type=int, default=8)
ap.add_argument("--min-us", type=float, default=4.0)
ap.add_argument("--delim", default="rmsprop")
a = ap.parse_args()
rows = list(sqlite3.connect(a.db).execute("select name,start,end,queue_id,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if a.delim in r[0]]
lo, hi = idx[a.step] + 1, idx[a.step + 1] + 1
main = rows[lo][4]
ms = [r for r in rows[lo:hi] if r[4] == main]
span = (ms[-1][2] - ms[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in ms) / 1e3
side = sum(r[2] - r[1] for r in rows[lo:hi] if r[4] != main) / 1e3
gaps = [((ms[i + 1][1] - ms[i][2]) / 1e3, ms[i][0], ms[i + 1][0], (ms[i][2] - ms[0][1]) / 1e3) for i in range(len(ms) - 1)]
tot = sum(g[0] for g in gaps if g[0] > 0)
print(f"# step {a.step}: main stream {len(ms)} launches, span {span:.1f} us, busy {busy:.1f} us, idle {tot:.1f} us; other streams busy {side:.1f} us")
short = lambda n: n.split("(")[0].replace("void ", "").replace("bot::", "").replace("(anonymous namespace)::", "")[-48:]
for g, p, n, t in sorted(gaps, reverse=True):
    if g >= a.min_us:
        print(f"{g:8.1f} us at {t:9.1f}   {short(p):48s} -> {short(n)}")
small = sum(g[0] for g in gaps if 0 < g[0] < a.min_us)
print(f"# gaps below {a.min_us} us: {small:.1f} us in {sum(1 for g in gaps if 0 < g[0] < a.min_us)} places")
