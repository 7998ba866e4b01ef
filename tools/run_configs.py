#!/usr/bin/env python3
"""Timing/sanity of the BASELINE.json configs through the public surface (synthetic weights/LM).
    python tools/run_configs.py [2 3 4 5]
"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from danspeech_amd import synthetic as syn
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd import Recognizer


def build(H, L, lm_order=None, beam=None, n_words=5000, conv=2):
    sd = syn.make_state_dict(conv, "gru", H, L, seed=0, **syn.TALKATIVE)
    m = DeepSpeech("cfg", rnn_hidden_size=H, rnn_layers=L, conv_layers=conv).load_state_dict(sd)
    rec = Recognizer(model=m)
    if lm_order:
        path = os.path.join(tempfile.gettempdir(), "syn%d.arpa" % lm_order)
        if not os.path.exists(path):
            syn.make_arpa(path, order=lm_order, n_words=n_words, seed=11, ngrams_per_order=20000)
        rec.update_decoder(lm=path, beam_width=beam)
    return rec


def run(name, rec, B, seconds, reps=3):
    """One batch at a time (recognize_batch: the latency of one call) and a stream of batches (recognize_batches: two in
    flight, the throughput of the surface); float64 host arrays in, strings out."""
    clips = [syn.make_clip(i, int(seconds * 16000)) for i in range(B)]
    for _ in range(5):                       # steady state: both pinned staging slots exist and torch's caching allocator
        rec.recognize_batch(clips)           # has settled (its blocks are recycled across streams; the first four calls grow it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = rec.recognize_batch(clips)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # a stream of batches with the forwards in flight the engine picks for this model and batch size (four where a forward is one
    # ring window, see DanSpeechRecognizer._lanes_that_pay; batches of more than 64 clips are cut into 64-clip forwards)
    nb = max(4 * reps + 2, 2048 // B)
    eng = rec.danspeech_recognizer
    for _ in range(2):
        for _r in rec.recognize_batches([clips] * max(6, 512 // B)):   # (the pipeline's own buffers: replica handles, decoder slots, allocator)
            pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for res in rec.recognize_batches([clips] * nb):
        n += 1
        assert res == out
    torch.cuda.synchronize()
    stream = (time.perf_counter() - t0) / n
    print("%-58s one call %7.1f ms = %7.0f audio-s/s | stream of %d batches, %d forwards of <= %d clips in flight %7.2f ms per batch = %7.0f audio-s/s"
          % (name, dt * 1e3, B * seconds / dt, nb, 1 + len(eng._replicas), min(B, eng.pipeline_merge_clips), stream * 1e3, B * seconds / stream), flush=True)
    return out


which = ([int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6]) if __name__ == "__main__" else []
if len(which) > 1:
    # One process per configuration: the ROCm runtime deals every stream a process creates onto GPU_MAX_HW_QUEUES hardware queues in
    # turn, so the second engine's lane and decode streams land on queues the first engine's (idle) streams already hold, two of its
    # own on one queue as often as not -- and two streams on one queue run one after the other (config 3 read 9.3 ms per batch as the
    # second engine of a process, 6.8-7.9 alone).
    import subprocess
    for cfg_no in which:
        subprocess.run([sys.executable, os.path.abspath(__file__), str(cfg_no)])
    which = []
if 2 in which:
    run("config 2: cfgA greedy B=32 x 10 s", build(800, 5), 32, 10.0)
if 3 in which:
    run("config 3: cfgA + 3-gram beam=64 B=32 x 10 s", build(800, 5, 3, 64), 32, 10.0)
if 4 in which:
    run("config 4: 7 x BiGRU1200 + 5-gram beam=128 B=64 x 10 s", build(1200, 7, 5, 128), 64, 10.0, reps=2)
if 5 in which:
    run("config 5 (one GPU's share): cfgA + 3-gram beam=64 B=128 x 30 s", build(800, 5, 3, 64), 128, 30.0, reps=1)
if 6 in which:     # SURVEY 8(d) side row: the docstring shape of DanSpeechPrimary (3 conv, 9 x BiGRU 1200), config 2's workload
    run("side row: 3 conv + 9 x BiGRU1200 greedy B=32 x 10 s", build(1200, 9, conv=3), 32, 10.0, reps=2)
