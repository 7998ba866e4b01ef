#!/usr/bin/env python3
"""What the matrix-pipe STFT kernel's time is made of: dsmi_features for 64 clips of 10 s (float64 PCM in HBM) with parts of the kernel
removed (DSMI_DEBUG_STFT_SKIP: 1 no hypotf / log1pf, 2 no MFMAs, 4 no sample / window loads; results are garbage), us per call
(STFT + clip statistics + normalise; the last two are ~40 us)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from explib import exp_env
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from danspeech_amd import _native as native, synthetic as syn
fe = native.NativeFrontend()
clips = np.stack([syn.make_clip(i, 160000) for i in range(64)])
pcm = torch.from_numpy(clips.reshape(-1)).cuda()
n = np.full(len(clips), 160000, dtype=np.int64)
for _ in range(3): feat, fr = fe.features(pcm, n)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20): feat, fr = fe.features(pcm, n)
ev[1].record(); torch.cuda.synchronize()
print("%%.0f" %% (ev[0].elapsed_time(ev[1]) * 1000 / 20))
''' % root
for skip, what in ((0, "the kernel"), (1, "no hypotf / log1pf"), (2, "MFMAs -> f64 vector ops (slower: says nothing)"), (4, "no sample / window loads"), (3, "no epilogue, MFMAs -> vector ops"),
                   (6, "no loads, MFMAs -> vector ops"), (7, "stores + vector ops only")):
    r = subprocess.run([sys.executable, "-c", child], env=exp_env(DSMI_DEBUG_STFT_SKIP=skip), capture_output=True, text=True)
    print("%-48s %s us per dsmi_features call" % (what, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
