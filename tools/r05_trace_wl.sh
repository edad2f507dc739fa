#!/bin/bash
# kernel-trace stats of a few steps of one workload:  tools/r05_trace_wl.sh products
cd "$GRAFT_REPO_ROOT" || exit 1
W=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 2 --cpu-baseline off --gemm halves > /tmp/b_w.log 2>&1
find /tmp/prof_w -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$W.csv \;
tail -1 /tmp/b_w.log | cut -c1-300
