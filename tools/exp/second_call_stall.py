#!/usr/bin/env python3
"""The one-off blocking upload of a process's SECOND recognize_batches call (profiles/r05_fill_drain.txt section 2): per call, every
host-to-device copy_ of a forward's clips that held the calling thread longer than 0.2 ms, with its lane.
    second_call_stall.py [batches of call 1 = 16] [batches of the later calls = 16] [calls = 4]
STALL_TOUCH=1: after call 1 every lane's staging slots are uploaded once more on the lane's stream (the candidate fix)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from danspeech_amd import Recognizer, synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
later = int(sys.argv[2]) if len(sys.argv) > 2 else 16
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B, N = 32, 160000
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
model = DeepSpeech("cfgA", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd)
rec = Recognizer(model=model)
host = [syn.make_clip(i, N) for i in range(B)]
eng = rec.danspeech_recognizer
log = []
_copy = torch.Tensor.copy_
def timed_copy(self, *a, **k):
    s = time.perf_counter()
    r = _copy(self, *a, **k)
    e = time.perf_counter()
    if self.is_cuda and self.numel() * self.element_size() > (1 << 20):
        log.append((torch.cuda.current_stream().cuda_stream, (e - s) * 1e3, self.data_ptr(), a[0].data_ptr()))
    return r
torch.Tensor.copy_ = timed_copy
for c in range(calls):
    del log[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = first if c == 0 else later
    for res in rec.recognize_batches([host] * n):
        pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    streams = {}
    for s, ms, dp, hp in log:
        streams.setdefault(s, len(streams))
    print("call %d: %d batches, %.1f ms (%.2f per batch); uploads (lane: ms, * = blocked): %s"
          % (c + 1, n, dt, dt / n, " ".join("%d:%.2f%s" % (streams[s], ms, "*" if ms > 1.0 else "") for s, ms, dp, hp in log)), flush=True)
    if c == 0:
        print("   device / pinned buffers seen in call 1: %d / %d" % (len({dp for _, _, dp, _ in log}), len({hp for _, _, _, hp in log})))
    if c == 0 and os.environ.get("STALL_TOUCH"):
        eng._touch_lanes() if hasattr(eng, "_touch_lanes") else None
