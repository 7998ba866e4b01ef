#!/usr/bin/env python3
"""The end of a short recognize_batches call: the last `lanes` batches of a LIST of batches as one forward each (pipeline_balance_tail)
against merged to the end; ms for 20 (or argv[1]) batches of 32 x 10 s float64 host clips, cfgA, alternating, after two warm-up calls
(a process's second call pays a one-off blocking upload per lane: pipeline_fill_log.py).
NOT ADOPTED (profiles/r05_fill_drain.txt): the switch exists in commit 0d2b105 only; on a later tree both rows measure the merged form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N = 32, 160000
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
model = DeepSpeech("cfgA", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd)
rec = Recognizer(model=model)
host = [syn.make_clip(i, N) for i in range(B)]
eng = rec.danspeech_recognizer
for _ in range(2):
    for res in rec.recognize_batches(host for _ in range(16)):
        want = res
torch.cuda.synchronize()
for rep in range(4):
    for balance in (False, True):
        eng.pipeline_balance_tail = balance
        t0 = time.perf_counter()
        outs = []
        for res in rec.recognize_batches([host] * steps):
            assert res == want
            outs.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
        print("last batches %-22s: %.1f ms for %d batches = %.2f ms per batch; results out at %s" %
              ("one forward each" if balance else "merged to the end", total, steps, total / steps, " ".join("%.0f" % t for t in outs[::2][-7:])), flush=True)
