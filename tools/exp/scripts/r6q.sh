#!/bin/bash
export TMPDIR=/tmp
export DSMI_RNN_KERNEL=ring4
cd /root/repo
echo "=== product library"; python3 tools/exp/ring4_race.py 128 4 12 ragged 2>&1 | grep "rounds with"
echo "=== signal as a RELEASE at agent scope"; DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_rel.so python3 tools/exp/ring4_race.py 128 4 24 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -6
for H in 64 192; do DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_rel.so python3 tools/exp/ring4_race.py $H 4 24 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -3; done
