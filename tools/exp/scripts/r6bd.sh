#!/bin/bash
# the GEMM as a persistent workgroup of two groups of four waves one phase apart (ping-pong) against two independent workgroups per CU
export TMPDIR=/tmp
cd /root/repo
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for r in 1 2; do
echo "--- alone, 64 clips, ping-pong"; timeout 300 python3 tools/exp/kernel_times_1inflight.py 64 2>/dev/null | grep -i "conv2\|gemm"
echo "--- alone, 64 clips, two workgroups"; DSMI_TEST_GEMM_PINGPONG=0 timeout 300 python3 tools/exp/kernel_times_1inflight.py 64 2>/dev/null | grep -i "conv2\|gemm"
done
b() { timeout 600 python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-side-paths --no-other-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
echo "--- steady ping-pong"; b
echo "--- steady two workgroups"; DSMI_TEST_GEMM_PINGPONG=0 b
done
