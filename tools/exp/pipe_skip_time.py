#!/usr/bin/env python3
"""The tile-walking recurrent kernel (H > 896) with parts removed (DSMI_DEBUG_PIPE_SKIP, timing only, results garbage):
   pipe_skip_time.py [H] [B] -> us per step of one layer alone on the chip for each mask.  One child process per mask (the
   library reads the variable once)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
HERE = os.path.dirname(os.path.abspath(__file__))
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import numpy as np, torch
    from danspeech_amd import _native, synthetic as syn
    T = 1001
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
    m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", H, 2, seed=0))
    m.set_profiling(2)
    x = torch.from_numpy(syn.make_features(B, T, seed=1)).cuda()
    lens = np.full(B, T, dtype=np.int32)
    for _ in range(4):
        m.forward(x, lens)
    v = m.kernel_stats()["rnn_layer_persistent"]
    print("%.2f" % (v["avg_us"] * v["launches"] / 8.0 / ((T + 1) // 2)))
    sys.exit(0)
NAMES = {0: "complete", 1: "no state loads", 2: "no MFMAs", 3: "no loads, no MFMAs", 4: "polls taken as answered", 8: "no stores", 16: "no cell",
         5: "no loads, no polls", 24: "no cell, no stores", 31: "barriers + partial tiles only"}
for mask in (0, 1, 2, 3, 4, 5, 8, 16, 24, 31):
    env = exp_env(DSMI_DEBUG_PIPE_SKIP=mask)
    out = subprocess.run([sys.executable, __file__, str(H), str(B), "child"], env=env, capture_output=True, text=True)
    print("skip %2d (%s): %s us per step" % (mask, NAMES[mask], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
