# Round 3: one bench line per BASELINE config (cpu_baseline + parity on the bounded sample, roofline from live HIP events), the rocprofv3
# kernel-trace summary of the headline command, and the PMC traffic of each config's dominant SpMM -> gpurun_out/r03c/
set -x
cd /root/repo
O=gpurun_out/r03c; mkdir -p $O
for W in cora reddit proteins products; do
  timeout 1200 python bench.py --workload $W --steps 10 --warmup 3 > $O/bench_$W.json 2> $O/bench_$W.err
  tail -c 400 $O/bench_$W.json
done
python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2> $O/bench_arxiv.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/bench_arxiv_kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
cd /root/repo
for W in arxiv reddit proteins products; do
  bash tools/pmc_bench.sh $W r03 2>&1 | tail -15
done
cp gpurun_out/pmc/* $O/ 2>/dev/null
ls -la $O
