#!/usr/bin/env python3
"""Diagnostics: where the ring kernel spends its phases.  ring_stamps.py [B]
The four-wave form (rnn_persist_ring4.hip; default): per wave and item (one phase): phase work (requests, MFMAs with the cell
between them, partial tiles), the wait for the wave's requests (publish stores drained, state DMA landed, x-projection arrived),
wave 0's poll spin, the barrier.  DSMI_RNN_KERNEL=ring8: the eight-wave form (two slots per item: M work, M-end waits, C work, the
barrier behind each, B wave 0's poll spin).  A diagnostics build whose stamps live in LDS."""
import os, sys
import numpy as np
os.environ["DSMI_STAMP_RING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danspeech_amd import _native, synthetic as syn
import ctypes as C
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, T = 800, 501
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
m = _native.NativeModel(cfg, syn.make_state_dict(2, "gru", H, 2, seed=0))
buf = np.zeros((256, 8, 16), dtype=np.uint64)
n = _native.lib().dsmi_debug_persist_stamps(m._h, 1, B, T, buf.ctypes.data_as(C.c_void_p), buf.size)
assert n > 0, n
items = T * 4
if os.environ.get("DSMI_RNN_KERNEL") == "ring8":
    raw = buf[:n].astype(np.float64)
    names = ["M work", "M-end waits", "C work", "barrier after M", "barrier after C", "poll spin"]
    print("B %d: %d workgroups, %d items per half; us per item (median over workgroups)" % (B, n, items))
    for half in (0, 1):
        for wv in (0, 1, 3):
            c = raw[:, 4 * half + wv, :]
            us = c[:, :6] * 10.0 / 1000.0 / items
            tot = np.median(us[:, 0] + us[:, 1] + us[:, 2] + us[:, 3] + us[:, 4])
            print("half %s wave %d: " % ("AB"[half], wv) + " | ".join("%s %.3f" % (names[k], np.median(us[:, k])) for k in range(6)) +
                  " | sum %.3f -> %.2f us per step" % (tot, tot * 4))
else:
    # the kernel writes [workgroup][4 waves][8] contiguously: the first 32 words of every 128-word row... of the flat buffer
    raw = buf.reshape(-1)[: n * 32].reshape(n, 4, 8).astype(np.float64)
    names = ["phase work", "wait for requests", "poll spin", "barrier"]
    tiles = max(4, -(-B // 16)) if B > 64 else 4
    print("B %d: %d workgroups, %d phases; us per phase (median over workgroups), wave = (group, K half)" % (B, n, int(np.median(raw[:, 0, 7]))))
    print("wave 0: polls whose first read was too early: %.0f of %.0f phases (median workgroup), re-reads per such poll %.2f"
          % (np.median(raw[:, 0, 5]), np.median(raw[:, 0, 7]), np.median(raw[:, 0, 6] / np.maximum(raw[:, 0, 5], 1))))
    for wv in range(4):
        us = raw[:, wv, :4] * 10.0 / 1000.0 / np.maximum(raw[:, wv, 7:8], 1)
        tot = np.median(us.sum(axis=1))
        mhz = np.median(raw[:, wv, 4] / np.maximum(raw[:, wv, 0], 1)) * 100.0
        print("wave %d (%d, %d): " % (wv, wv & 1, wv >> 1) + " | ".join("%s %.3f" % (names[k], np.median(us[:, k])) for k in range(4)) +
              " | sum %.3f -> %.2f us per step; shader clock during the phase work %.0f MHz" % (tot, tot * tiles, mhz))
