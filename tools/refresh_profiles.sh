# Regenerates what profiles/ holds for a round: bench line, rocprofv3 kernel stats of the same command, the --pmc passes over
# tools/microbench.py (FETCH_SIZE / WRITE_SIZE / TCC hits in separate passes), microbench and config 3-5 checks -> gpurun_out/final/.
set -x
cd /root/repo
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off > /tmp/b.log 2>&1
cp /tmp/prof_b/*kernel_stats.csv /root/repo/gpurun_out/final/bench_kernel_stats.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o r -- python3 /root/repo/tools/microbench.py --only spmm,spmm_dot,sddmm --iters 3 > /tmp/pmc_$n.log 2>&1
done
python3 /root/repo/tools/pmc_summary.py /tmp/pmc_* > /root/repo/gpurun_out/final/pmc_microbench.csv
wc -l /root/repo/gpurun_out/final/pmc_microbench.csv
python3 /root/repo/tools/microbench.py --iters 10 > /root/repo/gpurun_out/final/microbench.jsonl 2>/dev/null
python3 /root/repo/tools/scale_check.py reddit proteins products 2>/dev/null | grep workload > /root/repo/gpurun_out/final/scale_check.jsonl
