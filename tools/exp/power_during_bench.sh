#!/bin/bash
# Socket power and shader clock while bench.py's timed region runs (run on the GPU box from the repo root): is the pipeline power-bound?
python3 bench.py --steps 1920 --warmup 32 --no-cpu-baseline --no-side-paths > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
PID=$!
while kill -0 $PID 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "socket\|sclk\|mclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.4
done | grep -v "(9[0-9]Mhz)" | tail -40
tail -c 400 gpurun_out/power_bench.json | head -c 120; echo
