#!/bin/bash
export TMPDIR=/tmp
export DSMI_RNN_KERNEL=ring4
cd /root/repo
echo "=== as it is"; python3 tools/exp/ring4_race.py 128 4 12 ragged 2>&1 | grep "rounds with\|^H"
echo "=== every ring4 workgroup a CU of its own (LDS floor 84 KB)"; DSMI_TEST_RING4_LDS=86016 python3 tools/exp/ring4_race.py 128 4 12 ragged 2>&1 | grep "rounds with\|^H\|^round" | head -8
echo "=== H 256 / 320 / 400 as they are, 40 rounds"; for H in 256 320 400; do python3 tools/exp/ring4_race.py $H 4 40 ragged 2>&1 | grep "rounds with\|^round" | head -4; done
echo "=== H 800, 20 rounds"; python3 tools/exp/ring4_race.py 800 4 20 ragged 2>&1 | grep "rounds with\|^round" | head -4
