cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
mkdir -p /root/repo/gpurun_out/r03p
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r03p/kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
