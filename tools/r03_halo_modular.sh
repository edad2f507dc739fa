# The modular layers' overlapped halo exchange (bot_amd/halo.py) on the real kernels: the one-process GPU checks, then every
# workload's partitioned step on ONE rank over RCCL (no halo rows, but the whole overlapped code path, eager and captured)
# next to the unpartitioned step -> gpurun_out/r03h2/
set -x
cd /root/repo
O=gpurun_out/r03h2; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "halo or block_graphs or f4_community" > $O/gpu_tests_halo.log 2>&1; echo rc=$? >> $O/gpu_tests_halo.log; tail -4 $O/gpu_tests_halo.log
for W in reddit proteins products; do
  for MODE in plain part; do
    EXTRA=""; [ $MODE = part ] && EXTRA="--force-partitioned"
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload $W --scale 0.25 --steps 5 --warmup 2 --cpu-baseline off $EXTRA > $O/bench_${W}_$MODE.json 2> $O/bench_${W}_$MODE.err
    echo rc=$? ; tail -c 400 $O/bench_${W}_$MODE.json | cut -c1-400
  done
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --norm-adj symm --steps 10 --warmup 3 --cpu-baseline off --force-partitioned > $O/bench_arxiv_symm_part.json 2> $O/bench_arxiv_symm_part.err; echo rc=$?; tail -c 300 $O/bench_arxiv_symm_part.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload reddit --scale 0.25 --steps 5 --warmup 2 --cpu-baseline off --force-partitioned --capture on > $O/bench_reddit_part_capture.json 2> $O/bench_reddit_part_capture.err; echo rc=$?; tail -c 300 $O/bench_reddit_part_capture.json
