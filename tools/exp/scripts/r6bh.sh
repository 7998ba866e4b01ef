#!/bin/bash
# ring4 small-shape defect: which of three never-taken branches (experiments build, DSMI_NAP_ONLY=k at compile time) makes it go away
export TMPDIR=/tmp
cd /root/repo
for k in 0 1 2 3; do
  echo "--- library with dead branch $k only (0: none; 1: in front of the signal; 2: in front of the phase's requests and MFMAs; 3: behind the body)"
  DSMI_LIBRARY=danspeech_amd/lib/libdsmi_nap$k.so DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 4 300 2>/dev/null | tail -1
done
echo "--- production library"; DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 4 300 2>/dev/null | tail -1
