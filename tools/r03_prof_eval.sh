cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_e
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o r -- python3 /root/repo/tools/exp_eval_profile.py 20 > /tmp/e.log 2>&1
mkdir -p /root/repo/gpurun_out/r03e
find /tmp/prof_e -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r03e/eval_kernel_stats.csv \;
tail -2 /tmp/e.log
