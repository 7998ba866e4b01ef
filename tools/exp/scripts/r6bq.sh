#!/bin/bash
# ring4 small-shape defect: is it the poll's spin path?  An experiments build that takes that path once in every phase (tools/exp/ring4_force_spin.patch),
# ALONE on the chip: one handle against its own first run, and the ring tests' small shapes against the oracle.
export TMPDIR=/tmp
cd /root/repo
L=danspeech_amd/lib/libdsmi_spin.so
echo "--- one handle, forced spin path, H = 128 / 64 / 192 / 256"
for H in 128 64 192 256; do DSMI_LIBRARY=$L DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py $H 1 200 2>/dev/null | tail -1; done
echo "--- two handles, forced spin path, H = 128 (the library as it is: 4 of 24 with two handles)"
DSMI_LIBRARY=$L DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 2 200 2>/dev/null | tail -1
echo "--- tests/test_gpu_ring.py, form ring4, with that library"
DSMI_LIBRARY=$L timeout 900 python3 -m pytest tests/test_gpu_ring.py -q -k "ring4" 2>&1 | grep -E "passed|failed" | tail -2
echo "--- one handle, the experiments library as it is, H = 128"
DSMI_LIBRARY=danspeech_amd/lib/libdsmi_exp.so DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 1 200 2>/dev/null | tail -1
