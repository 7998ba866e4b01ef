#!/bin/bash
export TMPDIR=/tmp
run() { echo "=== $1"; shift; env "$@" python3 tools/exp/debug_short_forms.py 2>&1 | grep "clips differ" | sed 's/tail plan False, //'; }
for H in 64 96 128 160 192 256 320 400 512 800; do run "GRU H=$H" DBG_H=$H; done
run "LSTM H=128" DBG_H=128 DBG_KIND=lstm
run "RNN H=128" DBG_H=128 DBG_KIND=rnn
