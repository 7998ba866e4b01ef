#!/bin/bash
# round 6, call c: config 4 steady state (48 batches) at 2 / 3 / 4 forwards in flight with the turn lock, events as control; ring XCD tickets A/B
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
for L in 2 3 4; do python3 tools/exp/config_stream.py 4 $L 48 2>&1 | grep "^config"; done
DSMI_PERSIST_TURNS=events python3 tools/exp/config_stream.py 4 2 48 2>&1 | grep "^config"
echo "--- ring kernel alone, directions by XCD half (tickets) / by blockIdx"
python3 tools/exp/ring_layer_time.py 800 64 --only-auto 2>&1 | grep "^H"
DSMI_RING_XCD=0 python3 tools/exp/ring_layer_time.py 800 64 --only-auto 2>&1 | grep "^H"
python3 tools/exp/ring_layer_time.py 800 64 --only-auto 2>&1 | grep "^H"
DSMI_RING_XCD=0 python3 tools/exp/ring_layer_time.py 800 64 --only-auto 2>&1 | grep "^H"
echo "--- FETCH_SIZE of the ring launches (one pass each)"
for X in 1 0; do
  DSMI_RING_XCD=$X rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$X -- python3 tools/exp/ring_layer_time.py 800 64 --only-auto > $O/fetch_$X.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$O/fetch_$X/*/*counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "ring4" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("DSMI_RING_XCD=$X: %d ring4 launches, fetched %.3f GB per launch (FETCH_SIZE KiB x 2 x 1024, the guide's gfx950 correction)" % (len(v), sum(v) / max(len(v), 1) * 2048 / 1e9))
PY
  rm -rf $O/fetch_$X
done
echo "--- bench, steady state, tickets on / off"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_xcd1.json 2>/dev/null; python3 tools/exp/show_bench_line.py < $O/bench_xcd1.json
DSMI_RING_XCD=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_xcd0.json 2>/dev/null; python3 tools/exp/show_bench_line.py < $O/bench_xcd0.json
python3 - <<PY
import json
for x in (1, 0):
    d = json.loads(open("$O/bench_xcd%d.json" % x).read().strip().splitlines()[-1])
    print("XCD tickets %d: driver-command %.3f ms/step, steady %.3f ms/step, ring launch %.1f us, energy %s" % (x, d["ms_per_step"], d["steady_state"]["ms_per_step"], d["roofline"]["avg_launch_us"], d["energy"]))
PY
timeout 900 python -m pytest tests/test_gpu_ring.py -m gpu -x -q 2>&1 | tail -3
