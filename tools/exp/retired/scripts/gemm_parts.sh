# RETIRED with the experiment it drove: the DSMI_EXP_* switch it sets existed only in the experiment builds whose kernels are kept
# under tools/exp/retired/*.hip.inc; kept as the record of how the numbers in profiles/r03_gemm_bounds.txt / r03_duo_slot_stamps.txt were taken.
# What bounds the split-fp16 GEMM: the kernel alone on the chip with parts of its loop removed (results are wrong, timing only)
for V in base noepi; do
  if [ $V = base ]; then unset DSMI_EXP_GEMM; else export DSMI_EXP_GEMM=$V; fi
  echo "== $V"; timeout 200 python tools/exp/kernel_times_1inflight.py 2>&1 | grep -E "^gemm"
done
