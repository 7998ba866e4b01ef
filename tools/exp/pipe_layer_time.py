#!/usr/bin/env python3
"""Time of one recurrent layer on the multi-tile kernels (library timers): H, B from the command line (default 1200, 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = 1001
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", H, 2, seed=0))
m.set_profiling(2)
x = torch.from_numpy(syn.make_features(B, T, seed=1)).cuda()
lens = np.full(B, T, dtype=np.int32)
for _ in range(3):
    m.forward(x, lens)
ks = m.kernel_stats()
for k, v in ks.items():
    if "rnn" in k or "gemm" in k:
        print("H %d B %d  %-22s %8.1f us per launch (%d launches)  -> %.2f us per step" % (H, B, k, v["avg_us"], v["launches"], v["avg_us"] / ((T + 1) // 2)))
