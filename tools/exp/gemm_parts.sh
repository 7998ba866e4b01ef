# What bounds the split-fp16 GEMM: the kernel alone on the chip with parts of its loop removed (results are wrong, timing only)
for V in base noepi; do
  if [ $V = base ]; then unset DSMI_EXP_GEMM; else export DSMI_EXP_GEMM=$V; fi
  echo "== $V"; timeout 200 python tools/exp/kernel_times_1inflight.py 2>&1 | grep -E "^gemm"
done
