#!/bin/bash
# BatchNorm backward of the aggregate-first layer writing the halves operand itself under a bounded scale (BOT_L0_DIRECT=1) vs fp32 dx +
# split pass: tests, then the headline step both ways, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
[ -n "$AB_ONLY" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "grouped_halves or agg_first or full_size_config2 or captured or stacks_golden" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | grep "bn_bwd_bound\|passed\|failed\|Error\|assert" | tail -12
for k in 0 1 0 1; do
  BOT_L0_DIRECT=$k timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('BOT_L0_DIRECT=$k', round(l['ms_per_step'],3), 'ms/step')" | tee -a gpurun_out/r04/direct_step_ab.txt
done
