#!/bin/bash
# round 6, call h: the new GPU tests, then every profiles/ artefact of the round from this tree
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_recognizer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
bash tools/final_profiles.sh r06 $1 > $O/final.log 2>&1
tail -5 $O/final.log
python3 - <<PY
import json
for f in ("gpurun_out/r06_bench.json", "gpurun_out/r06_bench_driver_command.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], "steady", d["steady_state"]["ms_per_step"], "frac", d["roofline"]["frac"], "parity", d.get("parity_checked"), d.get("transcripts_identical"))
    print("   other:", {k: v.get("ms_per_batch") for k, v in (d.get("other_configs") or {}).items() if isinstance(v, dict)})
PY
cat gpurun_out/r06_run_configs.txt
