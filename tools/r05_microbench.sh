#!/bin/bash
# SURVEY 8d kernel micro-bench at the current ABI (tools/microbench.py on S-arxiv: F = 3 x 250, 128, 256, 40, the 168-wide broadcast forms, the
# halves-writing sweep) + the L2 <-> fabric traffic of every kernel from three separate rocprofv3 --pmc passes over the same command
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $OUT
python tools/microbench.py --iters 10 2>/dev/null > $OUT/microbench_arxiv.jsonl
tail -3 $OUT/microbench_arxiv.jsonl | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_m_*
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  k=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_m_$k -o r -- python3 $GRAFT_REPO_ROOT/tools/microbench.py --iters 3 > /tmp/pmc_m_$k.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_m_* > $OUT/pmc_microbench.csv
wc -l $OUT/pmc_microbench.csv
