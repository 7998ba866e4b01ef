#!/bin/bash
# round 6, call a: config 4 at 2 / 3 / 4 forwards in flight with a kernel trace each; config 5 split into 64-clip forwards; the driver's bench command
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
for L in 2 4; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t4_$L -- python3 tools/exp/config_stream.py 4 $L 12 > $O/cfg4_l$L.log 2>&1
  F=$(ls $O/t4_$L/*/*kernel_trace.csv | head -1)
  python3 tools/exp/overlap_report.py $F 120 > $O/cfg4_l${L}_overlap.txt 2>&1
  rm -rf $O/t4_$L
  grep "^config" $O/cfg4_l$L.log
done
python3 tools/exp/config_stream.py 4 3 12 2>&1 | grep "^config"
python3 tools/exp/config_stream.py 5 4 8 2>&1 | grep "^config"
python3 tools/exp/config_stream.py 5 2 8 128 2>&1 | grep "^config"
python3 tools/exp/config_stream.py 5 4 8 128 2>&1 | grep "^config"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
python3 tools/exp/show_bench_line.py < $O/bench_driver.json
