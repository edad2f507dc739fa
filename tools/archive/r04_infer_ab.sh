#!/bin/bash
# eval-mode forward of the aggregate-first layer on the grouped halves kernels (BOT_L0_INFER=1) vs projection + fused sweep: tests, then
# tools/f3_timing.py both ways, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
[ -n "$AB_ONLY" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "agg_first or evaluate or stacks_golden or train_step_golden or infer or grouped_halves" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -6
for k in 0 1 0 1; do
  BOT_L0_INFER=$k timeout 600 python tools/f3_timing.py 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('BOT_L0_INFER=$k evaluate', l['evaluate_n_label_iters=0']['inference_layers_ms'], 'ms; forward only', l['forward_only_inference_ms'], 'ms; train step', l['train_step_ms'], 'ms; evaluate with 1 label iteration', l['evaluate_n_label_iters=1']['inference_layers_ms'])" | tee -a gpurun_out/r04/infer_ab.txt
done
