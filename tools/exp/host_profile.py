#!/usr/bin/env python3
"""Where the host thread of the batch pipeline spends its time (cProfile, cumulative), host arrays and device-resident clips."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd.audio.parsers import DeviceClips
B, N = 32, 160000
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
rec = Recognizer(model=DeepSpeech("cfgA", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd))
host = [syn.make_clip(i, N) for i in range(B)]
clips = DeviceClips(torch.from_numpy(np.stack(host)).cuda().view(-1), np.full(B, N, dtype=np.int64))
eng = rec.danspeech_recognizer
for kind, src in (("host", host), ("device", clips)):
    for _ in eng.transcribe_batches([src] * 16):
        pass
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in eng.transcribe_batches([src] * 64):
        pass
    pr.disable()
    torch.cuda.synchronize()
    print("%s: %.2f ms per batch" % (kind, (time.perf_counter() - t0) / 64 * 1e3))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:40]))
