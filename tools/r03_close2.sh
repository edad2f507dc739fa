# after the halo.py work: the whole GPU suite again, the kernel fuzzer (now with the flat-lane SpMM), the headline line -> gpurun_out/r03y/
set -x
cd /root/repo
O=gpurun_out/r03y; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo rc=$? >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 900 python tools/fuzz_kernels.py 150 3 > $O/fuzz_kernels.txt 2>&1; echo rc=$? >> $O/fuzz_kernels.txt; tail -3 $O/fuzz_kernels.txt
python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2> $O/bench_arxiv.err; tail -c 200 $O/bench_arxiv.json
