#!/bin/bash
# halves3 NT kernel vs hipBLASLt: kernel-level comparison, then the headline step with each (same box, same process settings)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python tools/exp_halves3.py --ablate 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/halves3_kernel.txt
for k in lib halves3 lib halves3; do
  BOT_GEMM_NT=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$k', round(l['ms_per_step'],3), 'ms/step', l['roofline']['dense_projections'])" | tee -a gpurun_out/r04/halves3_step_ab.txt
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm_halves or stacks_golden or train_step_golden or full_size_config2 or captured_step_twelve" 2>&1 | tail -5
