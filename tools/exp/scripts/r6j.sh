#!/bin/bash
# round 6, call j: the failing short-call test; config 4 in / out of step, five streams each
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
python3 tools/exp/debug_short_forms.py 2>&1 | grep -v "amdgpu.ids\|Using device\|updated" | tee $O/debug_short.txt
echo "--- config 4, run_configs' stream, forwards kept apart / free"
for R in 1 2 3 4; do
  python3 tools/run_configs.py 4 2>&1 | grep "^config" | sed 's/.*in flight/apart:/'
  DSMI_PERSIST_STEP=free python3 tools/run_configs.py 4 2>&1 | grep "^config" | sed 's/.*in flight/free: /'
done
