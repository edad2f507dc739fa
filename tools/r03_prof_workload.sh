cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o r -- python3 /root/repo/bench.py --workload ${WL:-products} --steps 5 --warmup 2 --cpu-baseline off --gemm halves > /tmp/p.log 2>&1
mkdir -p /root/repo/gpurun_out/r03p
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r03p/kernel_stats_${WL:-products}.csv \;
tail -1 /tmp/p.log | cut -c1-160
