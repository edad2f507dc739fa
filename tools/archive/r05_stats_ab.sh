#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stats_byproduct or grouped_halves or agg_first or stacks_golden or train_step_golden or full_size_config2 or captured_step_twelve or side_stream" 2>&1 | grep -a "passed\|failed\|Error\|error" | tee gpurun_out/r05/stats_tests.txt
for k in 0 1 0 1; do
  BOT_STATS_BYPRODUCT=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('stats_byproduct=$k', round(l['ms_per_step'],3), 'ms/step')" | tee -a gpurun_out/r05/stats_step_ab.txt
done
