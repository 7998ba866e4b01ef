# RETIRED with the experiment it drove: the DSMI_EXP_* switch it sets existed only in the experiment builds whose kernels are kept
# under tools/exp/retired/*.hip.inc; kept as the record of how the numbers in profiles/r03_gemm_bounds.txt / r03_duo_slot_stamps.txt were taken.
# A/B of the split-fp16 GEMM's experimental shapes (DSMI_EXP_GEMM): alone on the chip, parity, in the pipeline
mkdir -p gpurun_out
for V in ${GEMM_AB_SHAPES:-base m256n256 m256n256st3}; do
  if [ $V = base ]; then unset DSMI_EXP_GEMM; else export DSMI_EXP_GEMM=$V; fi
  echo "== $V"
  timeout 200 python tools/exp/kernel_times_1inflight.py 2>&1 | grep -E "^gemm"
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -n 1
  for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-side-paths 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], 'gemm', k['gemm']['avg_us'], 'l0', k['gemm_l0']['avg_us'], 'conv2', k['conv2']['avg_us'], 'persist', k['rnn_layer_persistent']['avg_us'])"; done
done
