#!/usr/bin/env python3
"""Diagnostics: where the paired-tile persistent kernel (rnn_persist_duo.hip) spends its slots."""
import os, sys
import numpy as np
os.environ["DSMI_STAMP_DUO"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danspeech_amd import _native, synthetic as syn
import ctypes as C
H, T = 800, 501
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", H, 2, seed=0))
buf = np.zeros((256, 8, 8), dtype=np.uint64)
n = _native.lib().dsmi_debug_persist_stamps(m._h, 1, 32, T, buf.ctypes.data_as(C.c_void_p), buf.size)
assert n > 0, n
us = buf[:n].astype(np.float64) * 10.0 / 1000.0 / T
names = ["slot0 load+mfma", "slot1 cell", "slot2 drain+signal", "slot3 poll"]
for half in (0, 1):
    for wv in (0, 1):
        c = us[:, 4 * half + wv, :]
        print("half %d wave %d: " % (half, wv) + " | ".join("%s %.2f (+barrier %.2f)" % (names[k], np.median(c[:, k]), np.median(c[:, 4 + k])) for k in range(4)),
              "| slot0 shader clock %.0f MHz" % np.median(buf[:n][:, 4 * half + wv, 7].astype(np.float64) / np.maximum(buf[:n][:, 4 * half + wv, 0].astype(np.float64), 1) * 100.0))
