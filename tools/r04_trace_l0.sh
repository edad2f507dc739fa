#!/bin/bash
# kernel stats of the headline step (rocprofv3 --kernel-trace --stats), top kernels by total time
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_l0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l0 -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b_l0.log 2>&1
mkdir -p /root/repo/gpurun_out/r04
find /tmp/prof_l0 -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r04/l0_kernel_stats.csv \;
tail -1 /tmp/b_l0.log | cut -c1-200
