# Bench line (+ the stock-fp32-GEMM line beside it) and the rocprofv3 kernel stats of the same command -> gpurun_out/final/.
# The trace run sets BOT_GEMM_TUNE=0 so that the candidate-timing launches of the first step are not in the statistics, and names
# --gemm halves so that the extra stock-fp32 loop of the default run is not in them either.
set -x
cd /root/repo
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 300 gpurun_out/final/bench.json
python bench.py --steps 20 --warmup 5 --gemm f32 --cpu-baseline off > gpurun_out/final/bench_gemm_f32.json 2>> gpurun_out/final/bench.err
cd /tmp; export TMPDIR=/tmp
BOT_GEMM_TUNE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
cp /tmp/prof_b/*kernel_stats.csv /root/repo/gpurun_out/final/bench_kernel_stats.csv || find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/final/bench_kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
