#!/bin/bash
# left halves operands without the duplicate h1 piece (order 2, ABI 15) vs with it: tests, then the headline step both ways, same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm_halves or bn_halves or bn_act or stacks_golden or full_size_config2 or captured or step_glue or f4_kernels or train_step" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15
for k in 1 0 1 0; do
  BOT_HALVES_DUP=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('BOT_HALVES_DUP=$k', round(l['ms_per_step'],3), 'ms/step; dense', d['ms_per_step'], 'ms over', d['launches_per_step'], 'launches; parity', l.get('parity', {}).get('ok') if isinstance(l.get('parity'), dict) else l.get('parity'))" | tee -a gpurun_out/r04/nodup_step_ab.txt
done
