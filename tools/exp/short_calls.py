#!/usr/bin/env python3
"""Throughput of ONE recognize_batches call against the number of batches it is handed (cfgA, 32 x 10 s float64 host clips per batch):
1, 2, 4, 8, 20, 96 batches -> ms per batch and audio-s/s, median of five calls after two warm-up calls of 16.  The reference's
recognize() is one synchronous call (danspeech/Recognizer.py:82-95): the left end of this curve is what such a caller gets.
    short_calls.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 20, 96]
B, N = 32, 160000
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
model = DeepSpeech("cfgA", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd)
rec = Recognizer(model=model)
if os.environ.get("DBG_LANES"):
    rec.danspeech_recognizer.pipeline_lanes = int(os.environ["DBG_LANES"])
    rec.danspeech_recognizer._lanes_that_pay = lambda most, clips: most
if os.environ.get("DSMI_TEST_NO_TAIL"):
    rec.danspeech_recognizer.pipeline_balance_tail = False
host = [syn.make_clip(i, N) for i in range(B)]
for _ in range(2):
    for res in rec.recognize_batches([host] * 16):
        pass
one = rec.recognize_batch(host)
for n in ns:
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for res in rec.recognize_batches([host] * n):
            assert res == one
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print("%3d batches per call: %7.2f ms per call, %6.2f ms per batch = %6.0f audio-s/s  (min %.2f, max %.2f ms per batch)"
          % (n, t * 1e3, t / n * 1e3, B * 10.0 * n / t, min(ts) / n * 1e3, max(ts) / n * 1e3), flush=True)
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    rec.recognize_batch(host)
    ts.append(time.perf_counter() - t0)
print("recognize_batch (one synchronous call of 32 clips): %.2f ms = %.0f audio-s/s" % (sorted(ts)[2] * 1e3, B * 10.0 / sorted(ts)[2]))
