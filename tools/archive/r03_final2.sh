# Round 3, closing run at the final code (bot_amd/halo.py, by-product maxima, halves-only hidden states; ABI 12): GPU suite, one bench line per BASELINE config,
# smoke, kernel trace of the headline command, PMC traffic of the headline SpMM -> gpurun_out/r03n/
set -x
cd /root/repo
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo rc=$? >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2> $O/bench_arxiv.err; tail -c 200 $O/bench_arxiv.json
for W in cora reddit proteins products; do
  timeout 1500 python bench.py --workload $W --steps 10 --warmup 3 > $O/bench_$W.json 2> $O/bench_$W.err
  tail -c 200 $O/bench_$W.json
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/bench_arxiv_kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
cd /root/repo
bash tools/pmc_bench.sh arxiv r03 2>&1 | tail -12
cp gpurun_out/pmc/r03_pmc_bench_arxiv.csv gpurun_out/pmc/r03_traffic_arxiv.json $O/ 2>/dev/null
ls -la $O
