#!/usr/bin/env python3
"""Host-side timeline of a short recognize_batches call (the driver's 20 steps): when each forward's staging starts / ends (helper
thread), when it is enqueued (caller's thread) and when its results come out, ms from the call's start; cfgA, 32 x 10 s float64 host
clips per batch.    pipeline_fill_log.py [steps = 20]"""
import os, sys, time, threading
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
log = []
t0 = [0.0]
def wrap(name):
    inner = getattr(eng, name)
    def f(*a, **k):
        s = time.perf_counter()
        r = inner(*a, **k)
        log.append((name, threading.current_thread().name[:12], (s - t0[0]) * 1e3, (time.perf_counter() - t0[0]) * 1e3))
        return r
    setattr(eng, name, f)
for name in ("_stage_batch", "_enqueue_batch", "_finish_batch"):
    wrap(name)
# ... and the pieces of an enqueue, at class level (every lane's objects)
from danspeech_amd import _native
from danspeech_amd.audio import parsers as P
from danspeech_amd.deepspeech import model as M, decoder as D
def wrap_cls(cls, name, tag):
    inner = getattr(cls, name)
    def f(self, *a, **k):
        s = time.perf_counter()
        r = inner(self, *a, **k)
        log.append(("  " + tag, threading.current_thread().name[:12], (s - t0[0]) * 1e3, (time.perf_counter() - t0[0]) * 1e3))
        return r
    setattr(cls, name, f)
wrap_cls(_native.NativeFrontend, "features", "features")
wrap_cls(M.DeepSpeech, "enqueue", "model.enqueue")
wrap_cls(D.GreedyDecoder, "decode_enqueue", "decode_enqueue")
_empty = torch.empty
def timed_empty(*a, **k):
    s = time.perf_counter()
    r = _empty(*a, **k)
    e = time.perf_counter()
    if e - s > 2e-4:
        log.append(("    torch.empty %s" % (tuple(a[0]) if a and not isinstance(a[0], int) else a,), threading.current_thread().name[:12], (s - t0[0]) * 1e3, (e - t0[0]) * 1e3))
    return r
torch.empty = timed_empty
_copy = torch.Tensor.copy_
def timed_copy(self, *a, **k):
    s = time.perf_counter()
    r = _copy(self, *a, **k)
    e = time.perf_counter()
    if e - s > 2e-4:
        log.append(("    copy_ %d bytes" % (self.numel() * self.element_size()), threading.current_thread().name[:12], (s - t0[0]) * 1e3, (e - t0[0]) * 1e3))
    return r
torch.Tensor.copy_ = timed_copy
# PIPE_WARM="8,8": the warm-up calls before every timed call (lists of that many batches; default one generator of 16);
# PIPE_PROF=1: per-dispatch timestamps on (set_profiling(2)) as bench.py has them in its timed call;  PIPE_LIST=1: the timed call gets a list
warm = [int(v) for v in os.environ.get("PIPE_WARM", "").split(",") if v]
for rep in range(2):
    if warm:
        for n in warm:
            for res in rec.recognize_batches([host] * n):
                pass
    else:
        for res in rec.recognize_batches(host for _ in range(16)):
            pass
    torch.cuda.synchronize()
    if os.environ.get("PIPE_PROF"):
        for h in [eng.model._native] + [r[0]._native for r in eng._replicas]:
            h.set_profiling(2); h.reset_kernel_stats()
    del log[:]
    t0[0] = time.perf_counter()
    outs = []
    for res in rec.recognize_batches([host] * steps if os.environ.get("PIPE_LIST") else (host for _ in range(steps))):
        outs.append((time.perf_counter() - t0[0]) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0[0]) * 1e3
    print("run %d: %.1f ms for %d batches = %.2f ms per batch" % (rep, total, steps, total / steps))
    for name, th, s, e in sorted(log, key=lambda r: r[2]):
        print("   %-44s %-12s %7.2f .. %7.2f  (%5.2f)" % (name, th, s, e, e - s))
    print("   results out at: " + " ".join("%.1f" % t for t in outs))
