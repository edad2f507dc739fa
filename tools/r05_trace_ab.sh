#!/bin/bash
# kernel-trace stats of the headline step under two settings of one environment switch:  tools/r05_trace_ab.sh BOT_DOUT_DIRECT
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05/trace_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  export $V=$k
  rm -rf /tmp/prof_$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b_$k.log 2>&1
  find /tmp/prof_$k -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$k.csv \;
  tail -1 /tmp/b_$k.log | cut -c1-120
done
