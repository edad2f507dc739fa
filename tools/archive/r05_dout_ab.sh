#!/bin/bash
# ABI 17 (the hidden layer's gradient operand written by its producers): tests, then the headline step with / without, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dout_direct or side_stream or captured_step_twelve or stacks_golden or train_step_golden or full_size_config2" 2>&1 | grep -a "passed\|failed\|Error\|error" | tee gpurun_out/r05/dout_tests.txt
for k in 0 1 0 1; do
  BOT_DOUT_DIRECT=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('dout_direct=$k', round(l['ms_per_step'],3), 'ms/step', l['roofline']['dense_projections']['ms_per_step'], l.get('parity'))" | tee -a gpurun_out/r05/dout_step_ab.txt
done
