#!/bin/bash
# conv2 with a tap's x fragments read during the tap before (XPIPE) against the plain form: alone, parity, steady state
export TMPDIR=/tmp
cd /root/repo
E=danspeech_amd/lib/libdsmi_exp.so
for r in 1 2; do
echo "--- alone, 64 clips, XPIPE"; DSMI_LIBRARY=$E python3 tools/exp/kernel_times_1inflight.py 64 2>/dev/null | grep -i "conv\|gemm"
echo "--- alone, 64 clips, plain"; DSMI_LIBRARY=$E DSMI_DEBUG_CONV_XPIPE=0 python3 tools/exp/kernel_times_1inflight.py 64 2>/dev/null | grep -i "conv\|gemm"
done
b() { python3 bench.py --steps 48 --warmup 16 --no-cpu-baseline --no-side-paths --no-other-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
echo "--- steady XPIPE"; DSMI_LIBRARY=$E b
echo "--- steady plain"; DSMI_LIBRARY=$E DSMI_DEBUG_CONV_XPIPE=0 b
done
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
