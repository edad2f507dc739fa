#!/bin/bash
# kernel-trace stats of the headline step in the round-5 tree (ab_r05/) and in this tree, then the per-kernel diff (tools/stats_diff.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/trace_ab_r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in r05 r06; do
  D=$GRAFT_REPO_ROOT; [ $T = r05 ] && D=$GRAFT_REPO_ROOT/ab_r05
  rm -rf /tmp/prof_$T
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o t -- python3 $D/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b_$T.log 2>&1
  find /tmp/prof_$T -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$T.csv \;
  tail -1 /tmp/b_$T.log | cut -c1-120
done
python3 $GRAFT_REPO_ROOT/tools/stats_diff.py $OUT/stats_r05.csv $OUT/stats_r06.csv --steps 31 --min-us 3 > $OUT/diff.txt
cat $OUT/diff.txt
