#!/usr/bin/env python3
"""The batch pipeline of Recognizer.recognize_batches at different widths: forwards in flight (lanes) x clips per forward (merge),
cfgA, 32 x 10 s clips per batch resident in HBM, greedy.   pipeline_lanes.py [steps]   (DSMI_RNN_KERNEL=duo: the paired-tile kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd.audio.parsers import DeviceClips
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B, N = 32, 160000
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
model = DeepSpeech("cfgA", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd)
rec = Recognizer(model=model)
host = [syn.make_clip(i, N) for i in range(B)]
pcm = torch.from_numpy(np.stack(host)).cuda()
clips = DeviceClips(pcm.view(-1), np.full(B, N, dtype=np.int64))
want = None
combos = [(4, 64), (5, 64)] if not os.environ.get("DSMI_RNN_KERNEL") else [(2, 32), (2, 64), (4, 32)]
if os.environ.get("PIPE_COMBOS"):          # e.g. PIPE_COMBOS="4x64,2x128,3x128,4x128"
    combos = [tuple(int(v) for v in c.split("x")) for c in os.environ["PIPE_COMBOS"].split(",")]
for lanes, merge in combos:
    eng = rec.danspeech_recognizer
    host16 = [h.astype(np.int16) for h in host]
    host32 = [h.astype(np.float32) for h in host]
    order = (("host", lambda: (host for _ in range(steps))), ("device", lambda: (clips for _ in range(steps))))
    for kind, src in order:
        for res in eng.transcribe_batches((src() if False else (clips if kind == "device" else {"host": host, "host16": host16, "host32": host32}[kind]) for _ in range(16)), lanes=lanes, merge_clips=merge):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for res in eng.transcribe_batches(src(), lanes=lanes, merge_clips=merge):
            n += 1
            if want is None:
                want = res
            assert res == want, "transcripts differ"
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert n == steps
        print("lanes %d merge %3d %-6s: %.3f ms per 32-clip batch = %.1f k audio-s/s" % (lanes, merge, kind, dt / steps * 1e3, B * 10 * steps / dt / 1e3), flush=True)
