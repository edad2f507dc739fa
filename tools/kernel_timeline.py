"""Timeline of ONE train step from a rocprofv3 --kernel-trace database (rocpd sqlite): start, end, duration, stream of every launch that
lasts >= --min-us or runs on another stream than the step's first launch; steps are delimited by the optimizer kernel.
    python tools/kernel_timeline.py gpurun_out/r05/side_trace/side_results.db --step 5"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--step", type=int, default=5, help="index of the step (optimizer launch to optimizer launch)")
ap.add_argument("--min-us", type=float, default=20.0)
ap.add_argument("--delim", default="rmsprop")
a = ap.parse_args()
rows = list(sqlite3.connect(a.db).execute("select name,start,end,queue_id,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if a.delim in r[0]]
lo, hi = idx[a.step] + 1, idx[a.step + 1] + 1
t0 = rows[lo][1]
busy = sum(r[2] - r[1] for r in rows[lo:hi])
print(f"# step {a.step}: {hi - lo} launches, span {(rows[hi - 1][2] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
print("#  start_us    end_us   dur_us  stream  kernel")
for name, s, e, q, st in rows[lo:hi]:
    if (e - s) / 1e3 >= a.min_us or st != rows[lo][4]:
        print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  s{st}  {name.split('(')[0][-70:]}")
