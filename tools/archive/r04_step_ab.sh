#!/bin/bash
# fused step glue (csrc/step.hip + bot_amd.optim.RMSprop) vs the tensor-op step: tests, then the headline step both ways on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "step_glue or train_step_golden or captured or no_multi_workgroup or halves3" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
for k in 0 1 0 1; do
  BOT_FUSED_STEP=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('BOT_FUSED_STEP=$k', round(l['ms_per_step'],3), 'ms/step; optimizer', l['optimizer_ms'])" | tee -a gpurun_out/r04/step_glue_ab.txt
done
