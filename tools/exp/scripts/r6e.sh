#!/bin/bash
# round 6, call e: the copy-engine warm-up; short calls with forms by live occupancy; config 4 with at most two forwards between their
# first and last whole-device launch; the bench line with other_configs; the tests the changes touch
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
echo "--- second call after the copy-engine warm-up"; python3 tools/exp/second_call_stall.py 16 16 4 2>&1 | grep "^call\|device /" | tee $O/stall_after.txt
echo "--- short calls"; python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | tee $O/short_calls_after.txt
echo "--- config 4, 48 batches"
for L in 2 3 4; do python3 tools/exp/config_stream.py 4 $L 48 2>&1 | grep "^config"; done | tee $O/config4.txt
echo "--- bench, the driver's command"
( time python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | grep real
python3 tools/exp/show_bench_line.py < $O/bench_driver.json
python3 - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "steady", d["steady_state"]["ms_per_step"], "warmup_done", d["warmup_done"], "parity", d.get("parity_checked"), d.get("transcripts_identical"))
print(json.dumps(d.get("other_configs"), indent=1)[:3000])
PY
timeout 1500 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_timeout.py tests/test_gpu_ring.py tests/test_gpu_session.py -m gpu -x -q 2>&1 | tail -5
