#!/bin/bash
# the 128-byte-line NT kernel (+ fragment-major right operands): GEMM tests, then the headline step with the three forms, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python tools/exp_nt64.py 2>&1 | grep -v amdgpu.ids | tail -14 | tee gpurun_out/r05/nt64_kernel.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm_halves or grouped_halves or abi17 or dout_direct or stacks_golden or train_step_golden or full_size_config2 or agg_first or merged_linear" 2>&1 | grep -a "passed\|failed\|Error\|error" | tee gpurun_out/r05/nt64_tests.txt
for k in "128x64 0" "256x32 0" "256x32 1" "128x64 0" "256x32 0" "256x32 1"; do
  set -- $k
  BOT_NT_KERNEL=$1 BOT_RIGHT_FRAG=$2 python bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('nt=$1 frag=$2', round(l['ms_per_step'],3), 'ms/step', l['roofline']['dense_projections']['ms_per_step'], l['roofline']['dense_projections']['frac'])" | tee -a gpurun_out/r05/nt64_step_ab.txt
done
