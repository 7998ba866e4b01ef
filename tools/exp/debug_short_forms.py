#!/usr/bin/env python3
"""Debug: tests/test_gpu_recognizer.py::test_short_calls_take_other_kernel_forms_and_say_the_same, n = 6 from a list: which clips differ from
the single call, how, and whether it repeats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
H = int(os.environ.get("DBG_H", "128")); KIND = os.environ.get("DBG_KIND", "gru")
sd = syn.make_state_dict(2, KIND, H, 3, seed=12, fc_gain=8.0)
model = DeepSpeech("small", rnn_type=KIND, rnn_hidden_size=H, rnn_layers=3, conv_layers=2).load_state_dict(sd)
rec = Recognizer(model=model)
eng = rec.danspeech_recognizer
if os.environ.get("NO_PACK"):
    from danspeech_amd.audio.parsers import SpectrogramAudioParser
    SpectrogramAudioParser.pack_int16 = False
clips = [syn.make_clip(i, 9000 + 400 * (i % 5)) for i in range(32)]
want = rec.recognize_batch(clips)
for tail in (False,):
    eng.pipeline_balance_tail = tail
    for n in ((6, 8) * (10 if os.environ.get("DBG_MANY") else 3)):
        got = list(rec.recognize_batches([clips] * n))
        bad = [(k, i) for k in range(n) for i in range(32) if got[k][i] != want[i]]
        print("tail plan %s, %d batches: %d clips differ %s" % (tail, n, len(bad), bad[:6]), flush=True)
        for k, i in bad[:3]:
            print("    batch %d clip %d (len %d): got %r\n%swant %r" % (k, i, len(clips[i]), got[k][i], " " * 33, want[i]))
handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
print("recomputed:", [h.recompute_count() for h in handles])
