#!/bin/bash
# last check of the rebuilt libraries: smoke, the ring tests, the recognizer's pipeline tests, the default bench line
export TMPDIR=/tmp
cd /root/repo
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_ring.py tests/test_gpu_recognizer.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver command: ms_per_step', d['ms_per_step'], d['value'], 'steady', d['steady_state']['ms_per_step'], 'parity', d.get('parity_checked'), d.get('transcripts_identical'))"
