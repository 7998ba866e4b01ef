#!/bin/bash
# the four-wave ring kernel beside other ring windows, shapes the engine gives it (H >= 224): many rounds (tools/exp/ring4_race.py)
export TMPDIR=/tmp
cd /root/repo
for H in 224 256 400 800; do
  DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py $H 4 3000 2>/dev/null | grep -v "^round  *[0-9]" | tail -2
  DSMI_RNN_KERNEL=ring4 DBG_LEN=48000 timeout 600 python3 tools/exp/ring4_race.py $H 4 600 2>/dev/null | grep -v "^round  *[0-9]" | tail -1
done
echo "--- and the shape that fails, as a check of the hunt itself"
DSMI_RNN_KERNEL=ring4 timeout 300 python3 tools/exp/ring4_race.py 128 4 200 2>/dev/null | tail -1
