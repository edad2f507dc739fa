# MFMA-busy cycles of the matrix-core kernels of the headline step (bot:: kernels and the hipBLASLt ones): separate rocprofv3 --pmc passes,
# no tracing flags.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of active cycles.
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r04m; mkdir -p $O
rm -rf /tmp/pmc_m_*
for c in "SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES"; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_m_$c -o r -- python3 /root/repo/bench.py --steps 3 --warmup 2 --cpu-baseline off --gemm halves > /tmp/pmc_m_$c.log 2>&1
  tail -c 120 /tmp/pmc_m_$c.log
done
python3 - <<'PY' > $O/r04_pmc_mfma.csv
import collections, csv, glob, os
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_m_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("skinny_gemm", "tn_gemm", "Cijk_", "edge_mlp", "gemm_halves3")):
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(", ", ",")[:90]
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,avg_per_launch")
for (k, c), v in sorted(agg.items()):
    print(f"{k},{c},{len(v)},{sum(v)/len(v):.6g}")
PY
cat $O/r04_pmc_mfma.csv | head -80
python3 - <<'PY' > $O/r04_pmc_mfma_summary.txt
import csv, collections
rows = collections.defaultdict(dict)
for line in open("/root/repo/gpurun_out/r04m/r04_pmc_mfma.csv").read().splitlines()[1:]:
    k, c, n, v = line.rsplit(",", 3)            # (kernel names hold commas)
    rows[k][c] = (int(n), float(v))
print("# MFMA-busy share of wall cycles = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per launch (tools/pmc_mfma.sh: separate --pmc passes over bench.py --steps 3 --warmup 2 --gemm halves)")
out = []
for k, d in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"][1] > 0:
        wall = d["GRBM_GUI_ACTIVE"][1] / 8
        out.append((d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (1024 * wall), wall, d["GRBM_GUI_ACTIVE"][0], k))
for share, wall, n, k in sorted(out, reverse=True):
    print(f"{share:5.3f}  {wall:12.0f} cycles  x {n:3d}  {k}")
PY
cat $O/r04_pmc_mfma_summary.txt
