#!/usr/bin/env python3
"""The STFT kernel alone on the chip: 64 clips of 10 s float64 PCM resident in HBM, both forms (DSMI_DEBUG_STFT=direct: the float64
direct DFT on the vector pipe; default: v_mfma_f64_16x16x4_f64), microseconds per launch by the library's dispatch timers, and the
largest difference between the two forms' features.   stft_time.py [clips] [t_stride: row pitch of the feature tensor in frames, default 1001]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TS = int(sys.argv[2]) if len(sys.argv) > 2 else None
child = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from danspeech_amd import _native as native, synthetic as syn
fe = native.NativeFrontend()
clips = np.stack([syn.make_clip(i, 160000) for i in range(%d)])
pcm = torch.from_numpy(clips.reshape(-1)).cuda()
n = np.full(len(clips), 160000, dtype=np.int64)
TS = %r
for _ in range(3): feat, fr = fe.features(pcm, n, t_stride=TS)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20): feat, fr = fe.features(pcm, n, t_stride=TS)
ev[1].record(); torch.cuda.synchronize()
np.save(sys.argv[1], feat.cpu().numpy())
print("%%.0f us per dsmi_features call (STFT + statistics + normalise, 20 calls back to back)" %% (ev[0].elapsed_time(ev[1]) * 1000 / 20))
''' % (root, B, TS)
outs = []
for name, extra in (("direct float64 DFT (vector pipe)", {"DSMI_DEBUG_STFT": "direct"}), ("v_mfma_f64_16x16x4_f64", {})):
    f = "/tmp/stft_form_%d.npy" % len(outs)
    r = subprocess.run([sys.executable, "-c", child, f], env=exp_env(**extra), capture_output=True, text=True)
    print("%-36s: %s" % (name, (r.stdout.strip() or r.stderr.strip()[-400:])))
    outs.append(f)
import numpy as np
a, b = np.load(outs[0]), np.load(outs[1])
print("largest difference between the two forms' normalised features: %.3g" % np.abs(a - b).max())
