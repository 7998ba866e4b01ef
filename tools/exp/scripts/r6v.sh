#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for L in 3 4 5; do echo "=== $L forwards in flight, dense token on"; DBG_LANES=$L python3 tools/exp/short_calls.py 20 96 2>&1 | grep "batches per call" | cut -c1-100; done
