cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r04; mkdir -p $O
rm -rf /tmp/pmc_tn_*
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_tn_$n -o r -- python3 /root/repo/tools/exp_halves3.py --pmc-tn > /tmp/pmc_tn_$n.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_tn_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_halves3_tn" in k:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"gemm_halves3_tn_kernel,{c},{len(v)},{sum(v)/len(v):.6g}")
PY
