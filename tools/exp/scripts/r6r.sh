#!/bin/bash
export TMPDIR=/tmp
export DSMI_RNN_KERNEL=ring4
cd /root/repo
for V in 10 11; do
  echo "=== variant $V"
  DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_v$V.so python3 tools/exp/ring4_race.py 128 4 16 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-120 | tail -3
done
