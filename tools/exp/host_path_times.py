#!/usr/bin/env python3
"""Where the host spends a step of Recognizer.recognize_batches on float64 host arrays (cfgA, 32 x 10 s): wall time inside the
pipeline's pieces, per step, beside the same loop on clips resident in HBM."""
import collections, contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd.audio.parsers import DeviceClips, SpectrogramAudioParser
import importlib
eng_mod = importlib.import_module("danspeech_amd.DanSpeechRecognizer")

acc = collections.defaultdict(float)
def wrap(cls, name, key):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t
    setattr(cls, name, g)

B, N, STEPS = 32, 160000, 40
sd = syn.make_state_dict(2, "gru", 800, 5, bidirectional=True, seed=0, **syn.TALKATIVE)
model = DeepSpeech("cfgA", rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, conv_layers=2).load_state_dict(sd)
with contextlib.redirect_stdout(io.StringIO()):
    rec = Recognizer(model=model)
clips = [syn.make_clip(i, N) for i in range(B)]
dev = DeviceClips(torch.from_numpy(np.concatenate(clips)).cuda(), np.full(B, N, dtype=np.int64))
for what, batches in (("host arrays", lambda n: [clips] * n), ("resident", lambda n: [dev] * n)):
    for _ in rec.recognize_batches(batches(4)):
        pass
    torch.cuda.synchronize()
    if what == "host arrays" and not acc:
        wrap(SpectrogramAudioParser, "stage", "stage (pinned copy + upload enqueue)")
        wrap(SpectrogramAudioParser, "parse_batch", "parse_batch (incl. stage when not staged early)")
        wrap(eng_mod._BatchJob, "collect_forward", "collect_forward (waiting for the GPU)")
        wrap(eng_mod.DanSpeechRecognizer, "_enqueue_batch", "_enqueue_batch (total)")
        wrap(eng_mod.DanSpeechRecognizer, "_finish_batch", "_finish_batch (total, incl. its collect)")
    acc.clear()
    t0 = time.perf_counter()
    for _ in rec.recognize_batches(batches(STEPS)):
        pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS * 1e3
    print("%s: %.2f ms per step" % (what, dt))
    for k, v in sorted(acc.items()):
        print("    %-52s %.2f ms per step" % (k, v / STEPS * 1e3))
