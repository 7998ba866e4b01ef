#!/usr/bin/env python3
"""Config 3 (cfgA + 3-gram, beam 64, 32 x 10 s) as a stream of batches at different pipeline widths."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import synthetic as syn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from run_configs import build
rec = build(800, 5, 3, 64)
clips = [syn.make_clip(i, 160000) for i in range(32)]
want = rec.recognize_batch(clips)
eng = rec.danspeech_recognizer
for lanes, merge in ((2, 32), (2, 64), (3, 32), (3, 64), (4, 32), (4, 64)):
    for _ in eng.transcribe_batches([clips] * 16, lanes=lanes, merge_clips=merge):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for res in eng.transcribe_batches([clips] * 40, lanes=lanes, merge_clips=merge):
        n += 1
        assert res == want
    torch.cuda.synchronize()
    print("config 3, lanes %d merge %d: %.2f ms per batch" % (lanes, merge, (time.perf_counter() - t0) / n * 1e3), flush=True)
