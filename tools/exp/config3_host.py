"""Where the host's time goes per batch in config 3 (stream of batches): wraps the engine's steps with timers."""
import os, sys, time, tempfile, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from danspeech_amd import synthetic as syn, _native
from danspeech_amd.deepspeech.model import DeepSpeech
from danspeech_amd.deepspeech import decoder as dec_mod
from danspeech_amd import Recognizer
from danspeech_amd.DanSpeechRecognizer import DanSpeechRecognizer as Eng

acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; return r
    return w
_native.NativeDecoder.beam_collect = timed("native.beam_collect (C: wait, copies, rescoring)", _native.NativeDecoder.beam_collect)
_native.NativeDecoder.beam_enqueue = timed("native.beam_enqueue", _native.NativeDecoder.beam_enqueue)
dec_mod.BeamCTCDecoder.decode_collect = timed("decode_collect (incl. native.beam_collect)", dec_mod.BeamCTCDecoder.decode_collect)
Eng._enqueue_batch = timed("_enqueue_batch (staging, upload, enqueue)", Eng._enqueue_batch)
Eng._finish_batch = timed("_finish_batch (incl. decode_collect)", Eng._finish_batch)
DeepSpeech.collect = timed("model.collect (wait for the forward)", DeepSpeech.collect)

sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
m = DeepSpeech("cfg", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd)
rec = Recognizer(model=m)
path = os.path.join(tempfile.gettempdir(), "syn3.arpa")
syn.make_arpa(path, order=3, n_words=5000, seed=11, ngrams_per_order=20000)
rec.update_decoder(lm=path, beam_width=64)
clips = [syn.make_clip(i, 160000) for i in range(32)]
for _ in rec.recognize_batches([clips] * 8):
    pass
for show_all in (False, True):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for _ in rec.recognize_batches([clips] * 20, show_all=show_all):
        n += 1
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("show_all=%s: %.2f ms per batch" % (show_all, dt * 1e3))
    for k, v in sorted(acc.items()):
        print("   %-55s %6.2f ms per batch" % (k, v / n * 1e3))
