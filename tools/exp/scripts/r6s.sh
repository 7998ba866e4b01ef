#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
echo "=== fixed kernel, four-wave form forced, 4 handles"
for H in 64 96 128 160 192; do DSMI_RNN_KERNEL=ring4 python3 tools/exp/ring4_race.py $H 4 40 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-130 | tail -2; done
for K in lstm rnn; do DBG_KIND=$K DSMI_RNN_KERNEL=ring4 python3 tools/exp/ring4_race.py 128 4 20 ragged 2>&1 | grep "rounds with"; done
DSMI_RNN_KERNEL=ring4 python3 tools/exp/ring4_race.py 800 4 20 ragged 2>&1 | grep "rounds with"
echo "=== ring layer alone: us per step (was 6.4)"; python3 tools/exp/ring_layer_time.py 800 64 --only-auto 2>&1 | grep "^H"
timeout 1500 python -m pytest tests/test_gpu_ring.py tests/test_gpu_timeout.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6
