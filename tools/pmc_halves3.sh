# PMC passes over the hand-written halves GEMM (csrc/halves3.hip) and the library kernel on the config-2 forward shape.
# Separate rocprofv3 --pmc passes, no tracing flags (program directly after --).
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r04; mkdir -p $O
rm -rf /tmp/pmc_h3_*
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_h3_$n -o r -- python3 /root/repo/tools/exp_halves3.py --pmc > /tmp/pmc_h3_$n.log 2>&1
  tail -c 200 /tmp/pmc_h3_$n.log
done
python3 - <<'PY' > $O/pmc_halves3.csv
import collections, csv, glob
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_h3_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_halves3" in k or "Cijk_" in k:
            agg[(k.split("(")[0].replace("void ", "")[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,avg_per_launch")
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.6g}")
PY
cat $O/pmc_halves3.csv
