# halves-only hidden states (BOT_SKIP_Y): GPU checks, then A/B of the headline step on one box
cd /root/repo
mkdir -p gpurun_out/r03s
python -m pytest tests -m gpu -x -q -k "halves_only or absmax or captured or stacks_golden or train_step or agg_first or full_size_config2 or bench_line or evaluate or f4_kernels" 2>&1 | tail -4
for i in 1 2; do
  for S in 1 0; do
    BOT_SKIP_Y=$S python bench.py --steps 30 --warmup 5 --cpu-baseline off 2>/dev/null > gpurun_out/r03s/bench_skip${S}_$i.json
    python -c "import json; d=json.loads(open('gpurun_out/r03s/bench_skip${S}_$i.json').read().strip().splitlines()[-1]); print('BOT_SKIP_Y=$S', d['ms_per_step'])"
  done
done
