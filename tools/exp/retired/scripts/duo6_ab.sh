# RETIRED with the experiment it drove: the DSMI_EXP_* switch it sets existed only in the experiment builds whose kernels are kept
# under tools/exp/retired/*.hip.inc; kept as the record of how the numbers in profiles/r03_gemm_bounds.txt / r03_duo_slot_stamps.txt were taken.
# A/B of the six-slot paired-tile recurrent kernel (DSMI_EXP_DUO6) against the four-slot one
mkdir -p gpurun_out
for V in 0 1 0 1; do
  if [ $V = 0 ]; then unset DSMI_EXP_DUO6; else export DSMI_EXP_DUO6=1; fi
  echo "== duo6=$V"
  timeout 200 python bench.py --no-side-paths 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], d['parity_checked'], d.get('max_err'), d.get('transcripts_identical'), 'gemm', k['gemm']['avg_us'], 'l0', k['gemm_l0']['avg_us'], 'conv2', k['conv2']['avg_us'], 'persist', k['rnn_layer_persistent']['avg_us'])"
done
export DSMI_EXP_DUO6=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timeout.py tests/test_gpu_workloads.py -q -x 2>&1 | tail -n 3
