#!/usr/bin/env python3
"""One BASELINE config as a stream of batches through the pipeline with a FORCED number of forwards in flight (run_configs.py
lets the engine choose): ms per batch.  The short form a kernel trace is taken of.
    python tools/exp/config_stream.py <config 2..5> <lanes> [batches] [merge_clips]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from danspeech_amd import synthetic as syn
import run_configs as rc

cfg, lanes = int(sys.argv[1]), int(sys.argv[2])
shape = {2: (800, 5, None, None, 32, 10.0), 3: (800, 5, 3, 64, 32, 10.0), 4: (1200, 7, 5, 128, 64, 10.0), 5: (800, 5, 3, 64, 128, 30.0)}[cfg]
H, L, order, beam, B, seconds = shape
nb = int(sys.argv[3]) if len(sys.argv) > 3 else max(8, 512 // B)
merge = int(sys.argv[4]) if len(sys.argv) > 4 else None
rec = rc.build(H, L, order, beam)
eng = rec.danspeech_recognizer
clips = [syn.make_clip(i, int(seconds * 16000)) for i in range(B)]
for _ in range(2):
    for _r in eng.transcribe_batches([clips] * max(4, 256 // B), lanes=lanes, merge_clips=merge):
        pass
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for res in eng.transcribe_batches([clips] * nb, lanes=lanes, merge_clips=merge):
    n += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("config %d, %d forwards in flight%s, %d batches of %d x %.0f s: %.2f ms per batch = %.0f audio-s/s; recomputed %d"
      % (cfg, lanes, "" if merge is None else ", merge_clips %d" % merge, n, B, seconds, dt * 1e3, B * seconds / dt,
         sum(h._native.recompute_count() for h in [eng.model] + [r[0] for r in eng._replicas])), flush=True)
