# rocprofv3 --pmc passes (one counter set per pass, no tracing flags) over tools/f4_locality.py pmc for the SpMM on four
# (graph, numbering) pairs; summaries -> gpurun_out/f4/pmc_<graph>_<numbering>.csv
cd /tmp; export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/f4
for gn in "arxiv none" "arxiv degree" "arxiv-comm none" "arxiv-comm community"; do
  set -- $gn
  rm -rf /tmp/pmc_f4_*
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    k=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_f4_$k -o r -- python3 /root/repo/tools/f4_locality.py pmc $1 $2 > /tmp/pmc_f4_$k.log 2>&1
  done
  python3 /root/repo/tools/pmc_summary.py /tmp/pmc_f4_* > /root/repo/gpurun_out/f4/pmc_$1_$2.csv
done
