#!/usr/bin/env python3
"""The split-fp16 GEMM's forms alone on the chip (one 64-clip forward of cfgA at a time: per-kernel dispatch times): the 128 x 128 tile
with one and with two LDS stages (DSMI_DEBUG_GEMM_WIDE=0, DSMI_DEBUG_GEMM_STAGES), the 128 x 256 tile (default); the whole forward's
probabilities compared between the forms.  gemm_stages_time.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(%r)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
m = _native.NativeModel(cfg, sd)
x = torch.from_numpy(syn.make_features(64, 1001)).cuda()
lens = np.full(64, 1001, dtype=np.int32)
for _ in range(3):
    p, _ = m.forward(x, lens)
np.save(sys.argv[1], p.cpu().numpy())
m.set_profiling(2); m.reset_kernel_stats()
for _ in range(8):
    m.forward(x, lens)
torch.cuda.synchronize()
ks = m.kernel_stats()
print(" | ".join("%%s %%.0f us = %%.0f TF-equiv" %% (k, ks[k]["avg_us"], ks[k]["flops_per_launch"] / (ks[k]["avg_us"] * 1e-6) / 1e12) for k in ("gemm_l0", "gemm", "conv2", "rnn_layer_persistent")))
''' % os.path.join(here, "x.py")
outs = []
forms = (("128 x 128, one stage", {"DSMI_DEBUG_GEMM_WIDE": "0"}), ("128 x 128, two stages", {"DSMI_DEBUG_GEMM_WIDE": "0", "DSMI_DEBUG_GEMM_STAGES": "2"}), ("128 x 256, panel of 3 pairs (default)", {}), ("128 x 256, panel of 4 pairs", {"DSMI_DEBUG_GEMM_PN": "8"}))
for k, (name, extra) in enumerate(forms):
    env = exp_env(**extra)
    f = "/tmp/gemm_form_%d.npy" % k
    r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True)
    print("%-40s: %s" % (name, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
    outs.append(f)
import numpy as np
a = np.load(outs[0])
for k in range(1, len(forms)):
    print("max |probs(%s) - probs(%s)| = %.3g" % (forms[0][0], forms[k][0], float(np.abs(a - np.load(outs[k])).max())))
