#!/bin/bash
# Same-box alternating A/B of the headline step: this tree against the round-5 tree (ab_r05/: `git archive 74337a3` built in place; not
# committed).   tools/r06_ab_r05.sh <rounds> [extra bench args]
# Recreate ab_r05/ first (in the authoring container):  mkdir ab_r05 && git archive 74337a3 | tar -x -C ab_r05 && make -C ab_r05/bot_amd/csrc -j4 &&
#   make -C ab_r05/oracle/csrc && mkdir -p ab_r05/profiles && git show 74337a3:profiles/spmm_traffic.json > ab_r05/profiles/spmm_traffic.json
R=${1:-3}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
for i in $(seq 1 $R); do
  for T in r05 r06; do
    D=$GRAFT_REPO_ROOT; [ $T = r05 ] && D=$GRAFT_REPO_ROOT/ab_r05
    (cd $D && python3 bench.py --steps 30 --warmup 5 --cpu-baseline off "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$T round $i: %.3f ms/step  stock fp32 %.2f  spmm_rows %.4f ms  dense %.3f ms' % (l['ms_per_step'], (l.get('stock_fp32_gemm') or {}).get('ms_per_step', 0), l['roofline']['avg_launch_ms'], l['roofline']['dense_projections']['ms_per_step']))")
  done
done | tee $O/ab_r05.txt
