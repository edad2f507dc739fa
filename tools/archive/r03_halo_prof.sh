# kernel traces of the one-rank partitioned S-products step (scale 0.25) with the one-exchange form and with the overlapped form
set -x
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r03h3; mkdir -p $O
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
for OV in 0 1; do
  export BOT_HALO_OVERLAP=$OV
  rm -rf /tmp/prof_h$OV
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h$OV -o r -- python3 /root/repo/bench.py --gpus 1 --workload ${WL:-products} --scale 0.25 --steps 5 --warmup 2 --cpu-baseline off --force-partitioned --gemm halves > $O/b_$OV.log 2>&1
  find /tmp/prof_h$OV -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_ov$OV.csv \;
  tail -1 $O/b_$OV.log | cut -c1-160
done
