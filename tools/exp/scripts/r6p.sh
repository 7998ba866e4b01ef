#!/bin/bash
export TMPDIR=/tmp
export DSMI_RNN_KERNEL=ring4
cd /root/repo
for B in 16 32 48 64; do echo "=== B=$B, 4 handles"; DBG_B=$B python3 tools/exp/ring4_race.py 128 4 10 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -5; done
echo "=== one handle + a neighbour stream of fp32 GEMMs (torch)"; DBG_BURN=6 python3 tools/exp/ring4_race.py 128 1 16 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -5
echo "=== two handles + neighbour"; DBG_BURN=6 python3 tools/exp/ring4_race.py 128 2 16 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -5
echo "=== RNN / LSTM H=128, 4 handles, 20 rounds"; DBG_KIND=rnn python3 tools/exp/ring4_race.py 128 4 20 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -4; DBG_KIND=lstm python3 tools/exp/ring4_race.py 128 4 20 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -4
echo "=== GRU H=192 / 224, 4 handles"; for H in 192 224; do python3 tools/exp/ring4_race.py $H 4 20 ragged 2>&1 | grep "rounds with\|^round" | cut -c1-150 | head -3; done
