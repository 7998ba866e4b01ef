#!/bin/bash
# the GPU suite again, twice, on whatever box this is (flakiness), and the driver's command three times
export TMPDIR=/tmp
cd /root/repo
for r in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4; done
for r in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver command: ms_per_step', d['ms_per_step'], d['value'], 'steady', d['steady_state']['ms_per_step'], 'parity', d.get('parity_checked'), d.get('transcripts_identical'))"; done
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
