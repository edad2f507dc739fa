#!/bin/bash
# Round 6 closing runs.  Everything lands in gpurun_out/r06f/ and is copied into profiles/ by hand.
#   bash tools/r06_final.sh headline   bench line of the default command, kernel-trace stats of the same command, PMC traffic of the dominant SpMM
#                                      (three separate --pmc passes), MFMA-busy of the dense kernels (three more passes)
#   bash tools/r06_final.sh configs    one bench line per other BASELINE config + the PMC traffic of each one's dominant SpMM
#   bash tools/r06_final.sh suite      the GPU suite with durations
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
case "$1" in
headline)
  python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_pre.json 2> $O/bench_arxiv.err
  tail -c 300 $O/bench_arxiv_pre.json; echo
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/prof_b
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r -- python3 /root/repo/bench.py --steps 20 --warmup 5 --cpu-baseline off --gemm halves > /tmp/b.log 2>&1
  find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $O/bench_arxiv_kernel_stats.csv \;
  tail -1 /tmp/b.log | cut -c1-200
  cd /root/repo
  bash tools/pmc_bench.sh arxiv r06 2>&1 | tail -14
  cp gpurun_out/pmc/r06_* $O/ 2>/dev/null
  bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/r04m/r04_pmc_mfma.csv $O/r06_pmc_mfma.csv; cp gpurun_out/r04m/r04_pmc_mfma_summary.txt $O/r06_pmc_mfma_summary.txt
  cat $O/r06_pmc_mfma_summary.txt
  ;;
headline2)   # the bench line again, now replaying the traffic figures the PMC passes of `headline` produced (profiles/spmm_traffic.json updated in between)
  python bench.py --steps 20 --warmup 5 > $O/bench_arxiv.json 2>> $O/bench_arxiv.err
  tail -c 400 $O/bench_arxiv.json; echo
  python bench.py --steps 100 --warmup 5 --cpu-baseline off > $O/bench_arxiv_100steps.json 2>/dev/null
  tail -c 200 $O/bench_arxiv_100steps.json; echo
  ;;
configs)
  for W in cora reddit proteins products; do
    timeout 1500 python bench.py --workload $W --steps 10 --warmup 3 > $O/bench_$W.json 2> $O/bench_$W.err
    tail -c 300 $O/bench_$W.json; echo
  done
  for W in reddit proteins products; do
    bash tools/pmc_bench.sh $W r06 2>&1 | tail -12
  done
  cp gpurun_out/pmc/r06_* $O/ 2>/dev/null
  ;;
suite)
  python -m pytest tests -x -q -m gpu --durations=8 -s 2>&1 | grep -av "^\[Gloo\]\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gpu_tests.log
  tail -25 $O/gpu_tests.log
  ;;
esac
ls -la $O | tail -30
