#!/usr/bin/env python3
"""Ring kernel with parts removed (DSMI_DEBUG_RING_SKIP; results are garbage, timing only): what a 64-clip step is made of."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
only = [int(a) for a in sys.argv[1:]]
for skip, what in ((0, "complete"), (1, "no state DMA"), (2, "no MFMAs"), (3, "no DMA, no MFMAs"), (4, "no polls"), (5, "no DMA, no polls"),
                   (8, "no x-projection requests"), (16, "no output / publish stores"), (25, "no DMA, no x-projection requests, no stores: no memory request but the polls"), (29, "no memory request at all"), (27, "no memory request but the polls, no MFMAs"), (31, "barriers + cell only"), (32, "no wave priorities"), (63, "barriers + cell only, no priorities")):
    if only and skip not in only:
        continue
    env = dict(os.environ, DSMI_DEBUG_RING_SKIP=str(skip))
    out = subprocess.run([sys.executable, os.path.join(here, "ring_layer_time.py"), "800", "64"], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "kernel auto  inflight 2" in l]
    print("skip %2d (%s): %s" % (skip, what, line[0].split(":", 1)[1].strip() if line else out[-300:]), flush=True)
