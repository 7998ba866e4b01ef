#!/bin/bash
# the final tree: the whole GPU suite, smoke, the driver's command
export TMPDIR=/tmp
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_driver_command.json 2>/dev/null; echo "bench rc=$? stdout lines: $(wc -l < gpurun_out/final_driver_command.json)"
python3 -c "import json; d=json.loads(open('gpurun_out/final_driver_command.json').read().strip().splitlines()[-1]); print('driver command: ms_per_step', d['ms_per_step'], d['value'], 'steady', d['steady_state']['ms_per_step'], 'parity', d.get('parity_checked'), d.get('transcripts_identical'), 'other', {k: v.get('ms_per_batch') for k, v in (d.get('other_configs') or {}).items() if isinstance(v, dict)})"
