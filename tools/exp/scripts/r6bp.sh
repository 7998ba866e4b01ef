#!/bin/bash
# ring4 small-shape defect: idle cycles behind / in front of every barrier, or a second wait behind it (experiments builds, -DDSMI_BARRIER_PAD=k)
export TMPDIR=/tmp
cd /root/repo
for k in 1 2 3; do
  echo "--- pad $k (1: 3 x s_nop 15 behind s_barrier; 2: the same in front of it; 3: s_waitcnt vmcnt(0) lgkmcnt(0) behind it)"
  DSMI_LIBRARY=danspeech_amd/lib/libdsmi_pad$k.so DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 4 300 2>/dev/null | tail -1
done
echo "--- the experiments library as it is"; DSMI_LIBRARY=danspeech_amd/lib/libdsmi_exp.so DSMI_RNN_KERNEL=ring4 timeout 600 python3 tools/exp/ring4_race.py 128 4 300 2>/dev/null | tail -1
