#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r06_gpu_tests.txt
bash tools/final_profiles.sh r06 $1 > gpurun_out/r6x_final.log 2>&1
python3 - <<PY
import json
for f in ("gpurun_out/r06_bench.json", "gpurun_out/r06_bench_driver_command.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], "steady", d["steady_state"]["ms_per_step"], "frac", d["roofline"]["frac"], "parity", d.get("parity_checked"), d.get("transcripts_identical"))
    print("   other:", {k: v.get("ms_per_batch") for k, v in (d.get("other_configs") or {}).items() if isinstance(v, dict)})
PY
cat gpurun_out/r06_run_configs.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
