#!/bin/bash
# round 6, call d: the second call's blocking uploads; whose rounding flips transcript 256; the short-call curve
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
echo "--- calls of 16 batches"; python3 tools/exp/second_call_stall.py 16 16 4 2>&1 | grep -v "amdgpu.ids\|Using device\|updated"
echo "--- call 1 of 32 batches (every staging slot used twice in it)"; python3 tools/exp/second_call_stall.py 32 16 3 2>&1 | grep "^call\|device /"
echo "--- the runtime's own log of the blocking copies"
AMD_LOG_LEVEL=4 python3 tools/exp/second_call_stall.py 16 16 3 > $O/stall_level4.out 2> /tmp/amdlog.txt
grep "^call" $O/stall_level4.out
ls -la /tmp/amdlog.txt | awk '{print "log bytes", $5}'
python3 tools/exp/amdlog_slow_copies.py /tmp/amdlog.txt 80 > $O/slow_copies.txt 2>&1; head -c 20000 $O/slow_copies.txt | head -150
echo "--- whose rounding"
python3 tools/whose_rounding.py > $O/whose_rounding.txt 2>&1; tail -60 $O/whose_rounding.txt
echo "--- short calls"
python3 tools/exp/short_calls.py 2>&1 | grep "batches per call\|recognize_batch" | tee $O/short_calls.txt
