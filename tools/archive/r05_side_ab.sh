#!/bin/bash
# bot_amd.side (weight-gradient products on a second stream): tests, then the headline step with / without, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "side_stream or captured_step_twelve or agg_first or stacks_golden or train_step_golden" 2>&1 | grep -a "passed\|failed\|Error" | tee gpurun_out/r05/side_tests.txt
for k in "0 524288" "1 524288" "1 0" "0 524288" "1 524288" "1 0"; do
  set -- $k
  BOT_SIDE_STREAM=$1 BOT_SIDE_MIN_OUT=$2 python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('side=$1 min_out=$2', round(l['ms_per_step'],3), 'ms/step', l['roofline']['dense_projections']['ms_per_step'])" | tee -a gpurun_out/r05/side_step_ab.txt
done
