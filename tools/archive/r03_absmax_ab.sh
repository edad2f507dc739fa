cd /root/repo
mkdir -p gpurun_out/r03a2
python -m pytest tests -m gpu -x -q -k "absmax or halves or captured or stacks_golden or train_step or agg_first or random_shapes or ops_against or midsize" 2>&1 | tail -5
python bench.py --steps 30 --warmup 5 --cpu-baseline off > gpurun_out/r03a2/bench_by.json 2> gpurun_out/r03a2/bench_by.err; tail -c 250 gpurun_out/r03a2/bench_by.json; echo
BOT_ABSMAX_BYPRODUCT=0 python bench.py --steps 30 --warmup 5 --cpu-baseline off > gpurun_out/r03a2/bench_noby.json 2> gpurun_out/r03a2/bench_noby.err; tail -c 250 gpurun_out/r03a2/bench_noby.json; echo
python bench.py --steps 30 --warmup 5 --cpu-baseline off > gpurun_out/r03a2/bench_by2.json 2> /dev/null; tail -c 250 gpurun_out/r03a2/bench_by2.json
