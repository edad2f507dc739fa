#!/bin/bash
# Alternating step A/B of one environment switch on one box:  gpurun -- '[WL=products STEPS=8 WARM=3] bash tools/r05_ab.sh BOT_BN_BWD_BYPRODUCT [a b [rounds]]'  -> gpurun_out/r05ab/<switch>[_<workload>].txt
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; A=${2:-0}; B=${3:-1}; R=${4:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05ab
mkdir -p $OUT; SUF=${WL:+_$WL}
echo "# $V A/B, one box, alternating; bench.py ${WL:+--workload $WL }--steps ${STEPS:-30} --warmup ${WARM:-5} --cpu-baseline off; ms per step" > $OUT/$V$SUF.txt
for r in $(seq $R); do
  for k in $A $B; do
    env $V=$k python bench.py ${WL:+--workload $WL} --steps ${STEPS:-30} --warmup ${WARM:-5} --cpu-baseline off > /tmp/b.log 2>&1
    python - "$V" "$k" >> $OUT/$V$SUF.txt <<'P'
import json, sys
d = json.loads(open('/tmp/b.log').read().strip().splitlines()[-1])
print(f"{sys.argv[1]}={sys.argv[2]} {d['ms_per_step']:.3f} ms/step")
P
  done
done
cat $OUT/$V$SUF.txt
