#!/usr/bin/env python3
"""Diagnostics: where the persistent recurrent layer kernel spends its time per step."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danspeech_amd import _native, synthetic as syn
import ctypes as C

H = int(sys.argv[1]) if len(sys.argv) > 1 else 800
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", H, 2, seed=0)
m = _native.NativeModel(cfg, sd)
L = _native.lib()
T = 501
buf = np.zeros((256, 8, 8), dtype=np.uint64)
n = L.dsmi_debug_persist_stamps(m._h, 1, 32, T, buf.ctypes.data_as(C.c_void_p), buf.size)
assert n > 0, n
buf = buf[:n]
print("%d workgroups stamped" % n)
us = buf.astype(np.float64) * 10.0 / 1000.0 / T      # per-step average, microseconds
names = ["loop head", "wait", "h load + mfma", "lds + barrier", "cell", "publish"]
tot = us[:, :, :6].sum(axis=2)
print("per-step total: median %.2f us (min %.2f max %.2f)" % (np.median(tot), tot.min(), tot.max()))
for k, n in enumerate(names):
    c = us[:, :, k]
    print("  %-14s wave0 median %.2f | all waves median %.2f  min %.2f  max %.2f" % (n, np.median(c[:, 0]), np.median(c), c.min(), c.max()))
