#!/bin/bash
# Round 4 closing run: one bench line per BASELINE config, kernel trace of the headline command, PMC traffic of each dominant SpMM,
# MFMA-busy of the dense kernels, the GPU suite.  Everything lands in gpurun_out/r04f/ (copied to profiles/ by hand).
cd /root/repo
O=gpurun_out/r04j; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2> $O/bench_arxiv.err
tail -c 300 $O/bench_arxiv.json; echo
for W in cora reddit proteins products; do
  timeout 1500 python bench.py --workload $W --steps 10 --warmup 3 > $O/bench_$W.json 2> $O/bench_$W.err
  tail -c 300 $O/bench_$W.json; echo
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/bench_arxiv_kernel_stats.csv \;
tail -1 /tmp/b.log | cut -c1-200
cd /root/repo
for W in arxiv reddit proteins products; do
  bash tools/pmc_bench.sh $W r04 2>&1 | tail -12
done
cp gpurun_out/pmc/r04_* $O/ 2>/dev/null
bash tools/pmc_halves3.sh > /dev/null 2>&1; cp gpurun_out/r04/pmc_halves3.csv $O/ 2>/dev/null
bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/r04m/r04_pmc_mfma.csv gpurun_out/r04m/r04_pmc_mfma_summary.txt $O/ 2>/dev/null
ls -la $O
python tools/exp_halves3.py --ablate 2>&1 | grep -v amdgpu.ids > $O/halves3_kernel.txt
bash tools/pmc_halves3_tn.sh > $O/pmc_halves3_tn.csv 2>/dev/null
python tools/exp_halves3.py --layouts 2>&1 | grep -v amdgpu.ids > $O/halves3_layouts.txt
rm -f gpurun_out/r04/l0_step_ab.txt; L0_AB_ONLY=1 bash tools/r04_l0_ab.sh > /dev/null 2>&1; cp gpurun_out/r04/l0_step_ab.txt $O/ 2>/dev/null
rm -f gpurun_out/r04/direct_step_ab.txt; AB_ONLY=1 bash tools/r04_direct_ab.sh > /dev/null 2>&1; cp gpurun_out/r04/direct_step_ab.txt $O/ 2>/dev/null
rm -f gpurun_out/r04/tn_narrow_step_ab.txt; AB_ONLY=1 bash tools/r04_tn_narrow_ab.sh > /dev/null 2>&1; cp gpurun_out/r04/tn_narrow_step_ab.txt $O/ 2>/dev/null
python -m pytest tests -x -q -m gpu --durations=8 -s 2>&1 | grep -v "^\[Gloo\]\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_tests.log
tail -25 $O/gpu_tests.log
# config 4 at its FULL size once per closing run (the suite's default is half size: 300 s of oracle time otherwise)
BOT_CONFIG4_TEST_SCALE=1 python -m pytest tests/test_zz_full_size_gpu.py -x -q -m gpu -s -k config4 2>&1 | grep -v "^\[Gloo\]\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_test_config4_full.log
tail -3 $O/gpu_test_config4_full.log
