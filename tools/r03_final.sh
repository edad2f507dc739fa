# Round 3 closing run: the GPU test-suite, one bench line per BASELINE config (with cpu_baseline / parity / PMC traffic from
# profiles/spmm_traffic.json), the stock-GEMM line, and the rocprofv3 kernel-trace summary of the headline command -> gpurun_out/r03z/
set -x
cd /root/repo
O=gpurun_out/r03z; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1; echo rc=$? >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2> $O/bench_arxiv.err; tail -c 300 $O/bench_arxiv.json
for W in cora reddit proteins products; do
  timeout 1500 python bench.py --workload $W --steps 10 --warmup 3 > $O/bench_$W.json 2> $O/bench_$W.err
  tail -c 300 $O/bench_$W.json
done
python tools/train_halves_vs_f32.py 60 > $O/train_halves_vs_f32.txt 2>&1; tail -4 $O/train_halves_vs_f32.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/bench_arxiv_kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
ls -la /root/repo/$O
