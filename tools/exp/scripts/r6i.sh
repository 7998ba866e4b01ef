#!/bin/bash
# round 6, call i: what the per-dispatch stamps cost the driver's 20-step command (experiments build: the ring launches' sampling stride)
export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
export DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so
run() {   # label, env...
  L=$1; shift
  for R in 1 2 3; do
    env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-paths $EXTRA > $O/b.json 2>/dev/null
    python3 - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("$L run $R: ms_per_step", d["ms_per_step"], "ring launch us", r.get("avg_launch_us"), "kinds", {k: v["avg_us"] for k, v in (d.get("kernels") or {}).items()})
PY
  done
}
EXTRA=--no-kernel-sampling run "no stamps" A=1
EXTRA= run "every ring launch + every 5th dense (the default)" A=1
run "every 2nd ring launch + every 5th dense" DSMI_DEBUG_SAMPLE_RING_EVERY=2
run "every 4th ring launch + every 5th dense" DSMI_DEBUG_SAMPLE_RING_EVERY=4
run "every ring launch, no dense" DSMI_DEBUG_SAMPLE_EVERY=1000000
run "every 4th ring launch + every 20th dense" DSMI_DEBUG_SAMPLE_RING_EVERY=4 DSMI_DEBUG_SAMPLE_EVERY=20
unset DSMI_LIBRARY
echo "--- short calls with the last round dealt evenly"
python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | tee $O/short_calls_tail.txt
echo "--- the same without (pipeline_balance_tail off)"
DSMI_TEST_NO_TAIL=1 python3 tools/exp/short_calls.py 1 2 4 8 20 2>&1 | grep "batches per call"
timeout 900 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_workloads.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
echo "--- config 4: forwards kept out of step (progress words beside the turn lock)"
python3 tools/run_configs.py 4 2>&1 | grep "^config"
python3 tools/exp/config_stream.py 4 2 48 2>&1 | grep "^config"
python3 tools/run_configs.py 4 2>&1 | grep "^config"
rocprofv3 --kernel-trace --output-format csv -d $O/t4 -- python3 tools/run_configs.py 4 > $O/cfg4.log 2>&1
python3 tools/exp/overlap_report.py $(ls $O/t4/*/*kernel_trace.csv | head -1) 70 > $O/cfg4_overlap.txt 2>&1; rm -rf $O/t4; head -24 $O/cfg4_overlap.txt
timeout 900 python -m pytest tests/test_gpu_timeout.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4
