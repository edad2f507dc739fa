cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nt64 or gemm_halves or merged_linear or products_golden or proteins_golden or midsize_edge_gat or stacks_golden or dout_direct or absmax or train_step_golden or config1" 2>&1 | grep -a "passed\|failed\|Error\|error" | tee gpurun_out/r05/nodup_tests.txt
for k in 256 64 256 64; do
  BOT_NODUP_MIN_PIECE=$k python bench.py --workload products --steps 5 --warmup 2 --cpu-baseline off --gemm halves 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('products nodup_min_piece=$k', round(l['ms_per_step'],2), 'ms/step dense', d['ms_per_step'], d['frac'], d['combined_roofline'])" | tee -a gpurun_out/r05/nodup_products_ab.txt
done
for k in 256 64; do
  BOT_NODUP_MIN_PIECE=$k python bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); d=l['roofline']['dense_projections']; print('arxiv nodup_min_piece=$k', round(l['ms_per_step'],3), 'ms/step dense', d['ms_per_step'], d['frac'], d['combined_roofline'])" | tee -a gpurun_out/r05/nodup_products_ab.txt
done
for W in reddit proteins; do bash tools/r05_trace_wl.sh $W > /dev/null 2>&1; python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r05/stats_$W.csv')))
print('$W', 'hipBLASLt / rocBLAS kernels in the trace:', [r['Name'][:40] for r in rows if 'Cijk' in r['Name']][:5])
PY
done
