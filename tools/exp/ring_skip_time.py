#!/usr/bin/env python3
"""The four-wave ring kernel with parts removed (DSMI_DEBUG_RING_SKIP: compile-time variants of cfgA's shape; results are garbage,
timing only): what a 64-clip step is made of.  ring_skip_time.py [mask ...]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
here = os.path.dirname(os.path.abspath(__file__))
only = [int(a) for a in sys.argv[1:]]
for skip, what in ((0, "complete"), (1, "no state DMA"), (2, "no MFMAs"), (4, "no polls"), (8, "no x-projection requests"), (16, "no output / publish stores"),
                   (32, "no cell"), (9, "no state DMA, no x-projection requests"), (13, "no DMA, no x-projection, no polls"),
                   (29, "no memory request at all"), (31, "barriers + cell only"), (61, "MFMAs + partial tiles + barriers only"), (63, "barriers + partial tiles"),
                   (0x600, "complete, poll's first read at k-block 6"), (0x700, "... at k-block 7"), (0x900, "... at k-block 9"), (0xA00, "... at k-block 10"), (0xB00, "... at k-block 11")):
    if only and skip not in only:
        continue
    env = exp_env(DSMI_DEBUG_RING_SKIP=skip)
    env.pop("DSMI_RNN_KERNEL", None)
    out = subprocess.run([sys.executable, os.path.join(here, "ring_layer_time.py"), "800", "64", "--only-auto"], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "kernel auto  inflight 2" in l]
    print("skip %4d (%s): %s" % (skip, what, line[0].split(":", 1)[1].strip() if line else out[-300:]), flush=True)
