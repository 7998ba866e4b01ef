#!/usr/bin/env python3
"""The four-wave ring kernel with 4 / 6 / 8 tiles per window (DSMI_RING_TILES), one BiGRU layer alone on the chip:
us per launch, per step and per 16-clip tile-step.  ring_tiles_time.py [H]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
here = os.path.dirname(os.path.abspath(__file__))
H = sys.argv[1] if len(sys.argv) > 1 else "800"
for tiles, B in ((4, 64), (6, 96), (8, 128)):
    env = exp_env(DSMI_RING_TILES=tiles)
    env.pop("DSMI_RNN_KERNEL", None)
    out = subprocess.run([sys.executable, os.path.join(here, "ring_layer_time.py"), H, str(B), "--only-auto"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if "kernel auto  inflight 2" in l]
    if not line:
        print("tiles %d B %d: %s" % (tiles, B, (out.stdout + out.stderr)[-300:]))
        continue
    us_step = float(line[0].split("=")[1].split("us per step")[0])
    print("tiles %d B %3d: %s | %.2f us per tile-step" % (tiles, B, line[0].split(":", 1)[1].strip(), us_step / tiles), flush=True)
