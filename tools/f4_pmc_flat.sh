# rocprofv3 --pmc passes over the flat 16-byte-lane SpMM on the reordered community graph (rows on a 752-float pitch), next to the
# 8-byte head-segment kernel on the same operands -> gpurun_out/f4/pmc_flat_<layout>.csv
cd /tmp; export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/f4
export F4_PITCH=752
for flat in 1 0; do
  export BOT_SPMM_FLAT=$flat
  rm -rf /tmp/pmc_f4_*
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    k=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_f4_$k -o r -- python3 /root/repo/tools/f4_locality.py pmc arxiv-comm community > /tmp/pmc_f4_$k.log 2>&1
  done
  python3 /root/repo/tools/pmc_summary.py /tmp/pmc_f4_* > /root/repo/gpurun_out/f4/pmc_flat_$flat.csv
  cat /root/repo/gpurun_out/f4/pmc_flat_$flat.csv
done
