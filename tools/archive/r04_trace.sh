#!/bin/bash
# kernel trace of the default bench command (rocprofv3 --kernel-trace --stats), summary into gpurun_out/r04
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
rm -rf /tmp/tr_r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_r04 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > $O/trace_bench.json 2> /tmp/tr_r04.err
tail -c 300 /tmp/tr_r04.err
f=$(find /tmp/tr_r04 -name "*kernel_stats.csv" | head -1)
cp "$f" $O/bench_arxiv_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 28.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
nat = [(r["Name"], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3) for r in rows if "at::native" in r["Name"] or r["Name"].startswith("void at::")]
print(f"kernel time per step {tot:.3f} ms over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches; at::native {sum(x[2] for x in nat):.1f} us over {sum(x[1] for x in nat):.1f} launches")
for r in rows[:26]:
    print(f"{int(r['Calls']) / steps:6.2f}/step {float(r['TotalDurationNs']) / steps / 1e3:8.1f} us  {r['Name'][:110]}")
print("--- at::native")
for name, c, t in sorted(nat, key=lambda x: -x[2])[:25]:
    print(f"{c:6.2f}/step {t:7.1f} us  {name[:130]}")
PY
