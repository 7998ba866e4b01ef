#!/bin/bash
# round 6, call f: copy-engine warm-up from threads; clips as int16; where bench's second-call time goes
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
echo "--- second call after the threaded copy-engine warm-up (float64 uploads: DSMI_NO_PACK)"; python3 tools/exp/second_call_stall.py 16 16 4 2>&1 | grep "^call\|device /" | tee $O/stall_after.txt
echo "--- short calls"; python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | tee $O/short_calls_after.txt
echo "--- bench, the driver's command: one warm-up call / two / one without kernel sampling"
for V in "--warmup-calls 1" "--warmup-calls 2" "--warmup-calls 1 --no-kernel-sampling"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-paths $V > $O/b.json 2>/dev/null
  python3 - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$V: ms_per_step", d["ms_per_step"], "warmup_done", d["warmup_done"])
PY
done
( time python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real
python3 tools/exp/show_bench_line.py < $O/bench_driver.json
python3 - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "steady", d["steady_state"]["ms_per_step"], "warmup_done", d["warmup_done"], "parity", d.get("parity_checked"), d.get("transcripts_identical"), d.get("max_err"))
print({k: (v.get("ms_per_batch"), v.get("forwards_in_flight")) for k, v in d["other_configs"].items() if isinstance(v, dict)})
PY
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
