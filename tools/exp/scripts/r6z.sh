#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
b() { python3 bench.py --steps 20 --no-cpu-baseline --no-side-paths "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', 'ms_per_step', d['ms_per_step'], 'warmup_done', d['warmup_done'])"; }
b --warmup 5; b --warmup 5; b --warmup 5; b --warmup 5 --no-kernel-sampling; b --warmup 5 --warmup-calls 2
