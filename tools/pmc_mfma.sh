# MFMA-busy cycles of the matrix-core kernels of the headline step (bot:: kernels and the hipBLASLt ones): separate rocprofv3 --pmc passes,
# no tracing flags.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of active cycles.
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r03m; mkdir -p $O
rm -rf /tmp/pmc_m_*
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES"; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_m_$c -o r -- python3 /root/repo/bench.py --steps 3 --warmup 2 --cpu-baseline off --gemm halves > /tmp/pmc_m_$c.log 2>&1
  tail -c 120 /tmp/pmc_m_$c.log
done
python3 - <<'PY' > $O/r03_pmc_mfma.csv
import collections, csv, glob, os
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_m_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("skinny_gemm", "tn_gemm", "Cijk_", "edge_mlp", "gemm_halves3")):
            agg[(k.split("(")[0].replace("void ", "")[:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,avg_per_launch")
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.6g}")
PY
cat $O/r03_pmc_mfma.csv | head -60
