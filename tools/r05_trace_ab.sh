#!/bin/bash
# kernel-trace stats of the headline step under two settings of one environment switch:  tools/r05_trace_ab.sh BOT_DOUT_DIRECT [val_a val_b]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1
A=${2:-0}
B=${3:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/trace_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for k in $A $B; do
  export $V=$k
  rm -rf /tmp/prof_$i
  BOT_SIDE_STREAM=${SIDE:-1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$i -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b_$i.log 2>&1
  find /tmp/prof_$i -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$i.csv \;
  tail -1 /tmp/b_$i.log | cut -c1-120
  i=$((i+1))
done
