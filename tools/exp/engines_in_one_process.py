#!/usr/bin/env python3
"""Several engines, one after the other, in ONE process: does the n-th run as fast as the first?  (Before the lane streams were shared by
the engines of a process the later ones landed on hardware queues that were already taken.)"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
sd = syn.make_state_dict(2, "gru", 800, 5, bidirectional=True, seed=0, **syn.TALKATIVE)
clips = [syn.make_clip(i, 160000) for i in range(32)]
for k in range(4):
    model = DeepSpeech("cfgA", rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, conv_layers=2).load_state_dict(sd)
    with contextlib.redirect_stdout(io.StringIO()):
        rec = Recognizer(model=model)
    for _ in rec.recognize_batches([clips] * 16):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in rec.recognize_batches([clips] * 64):
        n += 1
    torch.cuda.synchronize()
    print("engine %d of the process: %.2f ms per batch" % (k + 1, (time.perf_counter() - t0) / n * 1e3), flush=True)
    # (the engine stays alive: its streams, handles and workspaces with it)
    globals()["keep%d" % k] = rec
