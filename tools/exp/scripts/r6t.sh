#!/bin/bash
# round 6, call t: at most K forwards in a dense kernel at a time (experiments build): does keeping the lanes out of step pay?
export TMPDIR=/tmp
cd /root/repo
export DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so
for R in 1 2; do
for K in 0 1 2 3; do
  echo "=== tokens $K (run $R)"; DSMI_DEBUG_DENSE_TOKENS=$K python3 tools/exp/short_calls.py 20 96 2>&1 | grep "batches per call" | cut -c1-110
done
done
