#!/bin/bash
# TN halves kernel vs the library formulation in the headline step, same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
for k in lib halves3 lib halves3; do
  BOT_GEMM_TN=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('BOT_GEMM_TN=$k', round(l['ms_per_step'],3), 'ms/step; dense', d['ms_per_step'], 'ms over', d['launches_per_step'], 'launches')" | tee -a gpurun_out/r04/tn_step_ab.txt
done
