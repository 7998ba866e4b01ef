#!/bin/bash
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_recognizer.py tests/test_gpu_ring.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
run() { echo "=== $1"; shift; env "$@" python3 tools/exp/debug_short_forms.py 2>&1 | grep "clips differ" | sed 's/tail plan False, //' | tr '\n' ';'; echo; }
for H in 64 128 192 224 256; do run "GRU H=$H (auto form)" DBG_H=$H; done
for H in 224 256 320 400; do run "GRU H=$H four-wave form, twenty calls" DBG_H=$H DBG_MANY=1; done
