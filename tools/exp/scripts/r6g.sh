#!/bin/bash
# round 6, call g: uploads issued from the staging thread; bench with one warm-up call; the whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
echo "--- uploads per call (now on the staging thread)"; python3 tools/exp/second_call_stall.py 16 16 4 2>&1 | grep "^call\|device /" | tee $O/stall_after.txt
echo "--- short calls"; python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | tee $O/short_calls_after.txt
echo "--- bench, the driver's command, twice (one warm-up call); then with two warm-up calls"
for V in "--warmup-calls 1" "--warmup-calls 1" "--warmup-calls 2"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-paths $V > $O/b.json 2>/dev/null
  python3 - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$V: ms_per_step", d["ms_per_step"], "warmup_done", d["warmup_done"], "ring launch us", d["roofline"]["avg_launch_us"])
PY
done
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -8
