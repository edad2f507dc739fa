#!/bin/bash
# ABI 18 (the BatchNorm-backward reduce pass as a by-product of the next layer's `d h` product): its GPU tests, an alternating step A/B of
# BOT_BN_BWD_BYPRODUCT and the kernel stats of both settings.   gpurun -- 'bash tools/r05_bnb.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05bnb
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn_bwd or abi18 or native_library or dout_direct or side_stream" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
echo "# BOT_BN_BWD_BYPRODUCT A/B, one box, alternating; ms per step" > $OUT/step_ab.txt
for r in 1 2 3; do
  for k in 0 1; do
    BOT_BN_BWD_BYPRODUCT=$k python bench.py --steps 30 --warmup 5 --cpu-baseline off > /tmp/b.log 2>&1
    python - "$k" >> $OUT/step_ab.txt <<'P'
import json, sys
d = json.loads(open('/tmp/b.log').read().strip().splitlines()[-1])
print(f"bn_bwd_byproduct={sys.argv[1]} {d['ms_per_step']:.3f} ms/step")
P
  done
done
cat $OUT/step_ab.txt
bash tools/r05_trace_ab.sh BOT_BN_BWD_BYPRODUCT 0 1
cp gpurun_out/r05/trace_BOT_BN_BWD_BYPRODUCT/stats_0.csv $OUT/stats_off.csv
cp gpurun_out/r05/trace_BOT_BN_BWD_BYPRODUCT/stats_1.csv $OUT/stats_on.csv
python tools/stats_diff.py $OUT/stats_off.csv $OUT/stats_on.csv --steps 31 > $OUT/kernel_diff.txt 2>&1
head -40 $OUT/kernel_diff.txt
