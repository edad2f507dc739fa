cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "grouped_halves or agg_first or full_size_config2" 2>&1 | grep -v "^\[Gloo\]\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -4
for k in 0 1 0 1; do
  BOT_L0_HALVES=$k timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('BOT_L0_HALVES=$k', round(l['ms_per_step'],3), 'ms/step; dense', d['ms_per_step'], 'ms over', d['launches_per_step'], 'launches')"
done
