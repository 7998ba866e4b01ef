#!/bin/bash
export TMPDIR=/tmp
export DSMI_RNN_KERNEL=ring4
cd /root/repo
python3 tools/exp/ring4_race.py 128 4 12 ragged 2>&1 | grep -v amdgpu.ids
python3 tools/exp/ring4_race.py 128 4 12 equal 2>&1 | grep -v amdgpu.ids
python3 tools/exp/ring4_race.py 128 2 12 ragged 2>&1 | grep -v amdgpu.ids
python3 tools/exp/ring4_race.py 128 1 12 ragged 2>&1 | grep -v amdgpu.ids
DBG_LAYERS=1 python3 tools/exp/ring4_race.py 128 4 12 ragged 2>&1 | grep -v amdgpu.ids
