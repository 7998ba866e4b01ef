#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for d in 0 25 50 100 150; do
echo "--- persistent ping-pong, workgroups' beginnings spread over $d % of a pair's time"; DSMI_TEST_GEMM_STAGGER=$d timeout 300 python3 tools/exp/kernel_times_1inflight.py 64 2>/dev/null | grep -i "gemm"
done
