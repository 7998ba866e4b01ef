#!/bin/bash
# round 6, call b: config 4 with the device-side turn lock at 2 / 3 / 4 forwards in flight (trace at 3), the event chain as control; the tests the lock touches
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
python3 tools/exp/config_stream.py 4 2 12 2>&1 | grep "^config"
rocprofv3 --kernel-trace --output-format csv -d $O/t4_3 -- python3 tools/exp/config_stream.py 4 3 12 > $O/cfg4_l3.log 2>&1
F=$(ls $O/t4_3/*/*kernel_trace.csv | head -1)
python3 tools/exp/overlap_report.py $F 150 > $O/cfg4_l3_overlap.txt 2>&1
rm -rf $O/t4_3
grep "^config" $O/cfg4_l3.log
python3 tools/exp/config_stream.py 4 4 12 2>&1 | grep "^config"
DSMI_PERSIST_TURNS=events python3 tools/exp/config_stream.py 4 2 12 2>&1 | grep "^config"
timeout 1200 python -m pytest tests/test_gpu_timeout.py tests/test_gpu_parity.py tests/test_gpu_workloads.py -m gpu -x -q 2>&1 | tail -5
