#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
echo "--- uploads by kernel: per-call times (second_call_stall.py logs torch copy_ only: none expected)"; python3 tools/exp/second_call_stall.py 16 16 4 2>&1 | grep "^call" | cut -c1-120
echo "--- short calls"; python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | cut -c1-110
echo "--- bench, driver command x3, one warm-up call"
for R in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-paths 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'warmup_done', d['warmup_done'])"; done
timeout 900 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_workloads.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5
