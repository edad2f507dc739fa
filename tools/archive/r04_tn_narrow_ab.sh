#!/bin/bash
# narrow TN results (the output layer's weight gradient) on the grouped halves kernel vs the library formulation: tests, headline step both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
[ -n "$AB_ONLY" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "gemm_halves or stacks_golden or full_size_config2 or midsize" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | grep "gemm.tn\|passed\|failed\|Error\|assert" | tail -12
for k in 0 1 0 1; do
  BOT_GEMM_TN_NARROW=$k timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('BOT_GEMM_TN_NARROW=$k', round(l['ms_per_step'],3), 'ms/step; dense', d['ms_per_step'], 'ms over', d['launches_per_step'], 'launches')" | tee -a gpurun_out/r04/tn_narrow_step_ab.txt
done
