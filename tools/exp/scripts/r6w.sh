#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
export DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so
for R in 1 2; do for S in 0 1 2; do echo "=== scope $S (run $R)"; DSMI_DEBUG_DENSE_SCOPE=$S python3 tools/exp/short_calls.py 20 96 2>&1 | grep "batches per call" | cut -c1-100; done; done
