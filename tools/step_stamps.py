#!/usr/bin/env python3
"""Diagnostics: where one recurrent-step launch spends its time (per-wave s_memrealtime stamps)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danspeech_amd import _native, synthetic as syn
import ctypes as C

cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=2, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", 800, 2, seed=0)
m = _native.NativeModel(cfg, sd)
L = _native.lib()
nwg = 80
for step in (0, 5, 100):
    buf = np.zeros((2 * nwg, 8, 8), dtype=np.uint64)
    rc = L.dsmi_debug_step_stamps(m._h, 1, 32, 501, step, buf.ctypes.data_as(C.c_void_p), buf.size)
    assert rc == 0, rc
    t = buf.astype(np.int64)
    t0 = t[:, :, 0].min()
    rel = (t - t0) * 10.0 / 1000.0   # us
    names = ["entry", "loads issued", "chunk0 done", "mfma done", "after sync", "end"]
    print("step", step)
    for k, n in enumerate(names):
        col = rel[:, :, k][t[:, :, k] > 0]
        if col.size:
            print("  %-14s min %.2f  median %.2f  max %.2f us" % (n, col.min(), np.median(col), col.max()))
