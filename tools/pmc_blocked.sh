# rocprofv3 --pmc passes (one counter set per pass) over tools/exp_blocked.py: L2 hit rate, LDS conflicts, wait cycles of the
# L2-blocked SpMM on S-reddit; folded by tools/pmc_summary.py.  Run on the GPU box.
cd /tmp; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb_$n -o r -- python3 /root/repo/tools/exp_blocked.py > /tmp/pmcb_$n.log 2>&1
  tail -2 /tmp/pmcb_$n.log | head -1
done
python3 /root/repo/tools/pmc_summary.py /tmp/pmcb_* | grep -E "kernel,|spmm_blocked" 
