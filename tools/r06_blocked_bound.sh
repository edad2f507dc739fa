#!/bin/bash
# Round 6: what bounds the dominant sweeps of configs 3 and 4 INSIDE the XCD.  Separate rocprofv3 --pmc passes (one counter
# set per pass, no tracing flags) over the bench command of one workload, folded per kernel by tools/pmc_summary.py; the kernel-trace
# stats of the same command give the durations.    bash tools/r06_blocked_bound.sh reddit|proteins
W=${1:-reddit}
O=$GRAFT_REPO_ROOT/gpurun_out/r06/bound_$W; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python3 /root/repo/bench.py --workload $W --steps 3 --warmup 2 --cpu-baseline off --gemm halves"
rm -rf /tmp/bb_*
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP_TCC_READ_REQ_sum|TCP_TCC_READ_REQ_LATENCY_sum|TCC_REQ_sum|TCC_READ_sum|TCC_HIT_sum|TCC_MISS_sum|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TCC_READ_REQ_LATENCY_sum|TCP_PENDING_STALL_CYCLES_sum|TCP_TA_TCP_STATE_READ_sum|TA_BUSY_avr|TA_TA_BUSY_sum|TCP_GATE_EN1_sum|TCP_TCR_TCP_STALL_CYCLES_sum|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_ACTIVE_INST_LDS|SQ_WAVES|SQ_INSTS_VALU|SQ_INSTS_VMEM_RD|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|SQ_BUSY_CYCLES|SQ_INST_CYCLES_VMEM_RD|SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_VMEM|SQ_LEVEL_WAVES|GRBM_GUI_ACTIVE)\b" | sort -u > $O/available_counters.txt
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LEVEL_WAVES" "GRBM_GUI_ACTIVE"; do
  k=$(echo $c | tr ' ' '_' | cut -c1-60)
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/bb_$k -o r -- $CMD > /tmp/bb_$k.log 2>&1
  echo "$c: rc $? $(tail -c 120 /tmp/bb_$k.log | tr '\n' ' ')"
done
python3 /root/repo/tools/pmc_summary.py /tmp/bb_* > $O/pmc.csv
rm -rf /tmp/bb_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bb_trace -o r -- $CMD > /tmp/bb_trace.log 2>&1
find /tmp/bb_trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -8 $O/kernel_stats.csv | cut -c1-200
grep -E "spmm_blocked|spmm_dot_rows|spmm_rows" $O/pmc.csv
