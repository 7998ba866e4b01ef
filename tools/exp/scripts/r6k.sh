#!/bin/bash
# round 6, call k: which of this round's changes corrupts the shortest clips of 64-clip forwards in the pipeline
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
run() { echo "=== $1"; shift; env "$@" python3 tools/exp/debug_short_forms.py 2>&1 | grep "clips differ\|recomputed"; }
run "as it is" A=1
# (needed a debug switch in audio/parsers.py that is gone: uploads issued at enqueue time, main thread)
# run "uploads issued at enqueue time (main thread)" DSMI_TEST_LATE_UPLOAD=1
run "float64 uploads" NO_PACK=1
# run "both" NO_PACK=1 DSMI_TEST_LATE_UPLOAD=1
run "experiments build, ring directions by blockIdx" DSMI_LIBRARY=$PWD/danspeech_amd/lib/libdsmi_exp.so DSMI_DEBUG_RING_XCD=0
run "eight-wave ring form everywhere" DSMI_RNN_KERNEL=ring8
run "paired-tile kernels instead of ring" DSMI_RNN_KERNEL=duo
