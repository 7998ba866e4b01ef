#!/usr/bin/env python3
"""Per-clip error of one ring-kernel case against the oracle: ring_case_debug.py kind H B inflight bidirectional [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import _native, synthetic as syn
from oracle import torch_port as tp
kind, H, B, infl, bi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1"
T = int(sys.argv[6]) if len(sys.argv) > 6 else 181
cfg = dict(conv_layers=2, rnn_type=kind, rnn_hidden_size=H, rnn_layers=2, bidirectional=bi, context=20)
sd = syn.make_state_dict(2, kind, H, 2, bidirectional=bi, context=20, seed=91, **syn.TALKATIVE)
lens = np.sort(np.random.default_rng(92).integers(T // 2, T + 1, size=B))[::-1].astype(np.int32)
lens[0] = T
x = syn.make_features(B, T, seed=92)
for b, L in enumerate(lens):
    x[b, :, :, L:] = 0
ref, ol = tp.forward(sd, cfg, x, lens)
for rep in range(3):
    m = _native.NativeModel(cfg, sd)
    m.set_inflight(infl)
    p, _ = m.forward(torch.from_numpy(x).cuda(), lens)
    pn = p.cpu().numpy()
    errs = np.array([np.abs(pn[b, :ol[b]] - ref[b, :ol[b]]).max() for b in range(B)])
    bad = np.nonzero(errs > 1e-4)[0]
    print("%s H %d B %d inflight %d bi %d rep %d: max err %.3g, clips over 1e-4: %s, recomputed %d" % (kind, H, B, infl, bi, rep, errs.max(), list(bad), m.recompute_count()))
    for b in bad[:3]:
        e = np.abs(pn[b, :ol[b]] - ref[b, :ol[b]]).max(axis=1)
        fr = np.nonzero(e > 1e-4)[0]
        print("   clip %d (len %d): frames over 1e-4: %d, first %s last %s" % (b, ol[b], len(fr), fr[:5], fr[-5:]))
    m.close()
