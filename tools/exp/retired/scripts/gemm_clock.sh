# RETIRED with the experiment it drove: the DSMI_EXP_* switch it sets existed only in the experiment builds whose kernels are kept
# under tools/exp/retired/*.hip.inc; kept as the record of how the numbers in profiles/r03_gemm_bounds.txt / r03_duo_slot_stamps.txt were taken.
# Shader clock and MFMA-busy share of the GEMM in each timing experiment: GRBM_GUI_ACTIVE / (end - start), MFMA busy / active
export TMPDIR=/tmp
for V in base nodma nolds nomfma m256n256st3; do
  if [ $V = base ]; then unset DSMI_EXP_GEMM; else export DSMI_EXP_GEMM=$V; fi
  rm -rf gpurun_out/gclk_$V
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/gclk_$V -- python3 tools/exp/kernel_times_1inflight.py > gpurun_out/gclk_$V.log 2>&1 || echo "pass $V failed"
  python3 - $V <<'PY'
import csv, glob, collections, sys
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/gclk_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gemm" if "gemm_f16x3_kernel<false" in n else "gemm_l0" if "gemm_f16x3_kernel<true" in n else "conv2" if "conv_f16x3" in n else "persist" if "rnn_persist" in n else None
        if k:
            v[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            v[k]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(v):
    m = {c: sum(x) / len(x) for c, x in v[k].items()}
    g = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%-14s %-8s %7.1f us  %8.0f cycles/XCD  %.2f GHz  mfma busy %.3f" % (sys.argv[1], k, m["ns"] / 1e3, g, g / m["ns"], m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / g if g else 0))
PY
done
