#!/usr/bin/env python3
"""One recurrent layer alone on the chip, the four-wave ring kernel against the eight-wave form (ring8) and the older ones (library timers, cfgA's width):
   ring_layer_time.py [H] [B ...]   -> us per launch and per step for each B with DSMI_RNN_KERNEL unset and =duo, inflight 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
H = int(sys.argv[1]) if len(sys.argv) > 1 else 800
only_auto = "--only-auto" in sys.argv
Bs = [int(a) for a in sys.argv[2:] if not a.startswith("--")] or [32, 64, 128]
T = 1001
To = (T + 1) // 2
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=H, rnn_layers=2, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", H, 2, seed=0)
for B in Bs:
    x = torch.from_numpy(syn.make_features(B, T, seed=1)).cuda()
    lens = np.full(B, T, dtype=np.int32)
    for kern, infl in ((("", 2),) if only_auto else (("", 2), ("ring8", 2), ("duo", 2), ("", 1))):
        if kern:
            os.environ["DSMI_RNN_KERNEL"] = kern
        else:
            os.environ.pop("DSMI_RNN_KERNEL", None)
        m = _native.NativeModel(cfg, sd)
        m.set_inflight(infl)
        m.set_profiling(2)
        for _ in range(4):
            m.forward(x, lens)
        v = m.kernel_stats()["rnn_layer_persistent"]
        per_layer = v["avg_us"] * v["launches"] / 8.0
        print("H %d B %3d kernel %-5s inflight %d: %8.1f us per launch, %d launches per layer, %8.1f us per layer = %.2f us per step, recomputed %d" %
              (H, B, kern or "auto", infl, v["avg_us"], v["launches"] // 8, per_layer, per_layer / To, m.recompute_count()), flush=True)
        m.close()
