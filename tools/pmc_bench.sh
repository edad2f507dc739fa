# PMC traffic of the dominant SpMM of a bench workload: three rocprofv3 --pmc passes (one counter set per pass, NO tracing flags —
# gpurun refuses --pmc combined with trace domains) over the SAME command the bench line comes from, folded per kernel by
# tools/pmc_summary.py, then written into profiles/spmm_traffic.json[<workload>] by tools/pmc_traffic.py.
#   bash tools/pmc_bench.sh arxiv|reddit|proteins|products|cora [round tag]
W=${1:-arxiv}; TAG=${2:-r03}
cd /tmp; export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/pmc
rm -rf /tmp/pmc_b_*
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  k=$(echo $c | tr ' ' '_')
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_b_$k -o r -- python3 /root/repo/bench.py --workload $W --steps 3 --warmup 2 --cpu-baseline off --gemm halves > /tmp/pmc_b_$k.log 2>&1
  tail -c 200 /tmp/pmc_b_$k.log
done
python3 /root/repo/tools/pmc_summary.py /tmp/pmc_b_* > /root/repo/gpurun_out/pmc/${TAG}_pmc_bench_$W.csv
python3 /root/repo/tools/pmc_traffic.py $W /root/repo/gpurun_out/pmc/${TAG}_pmc_bench_$W.csv /tmp/pmc_b_FETCH_SIZE.log $TAG > /root/repo/gpurun_out/pmc/${TAG}_traffic_$W.json
cat /root/repo/gpurun_out/pmc/${TAG}_traffic_$W.json
