#!/usr/bin/env python3
"""The REFERENCE itself timed beside its port, on the same cores (build container only: /root/reference is imported, nothing of
it travels).  SURVEY 8(d)(i) / BASELINE.md 3:

 (a) the reference's only public path: ``Recognizer.recognize()`` one clip at a time over 32 x 10 s clips
     (/root/reference/danspeech/Recognizer.py:82-95 -> DanSpeechRecognizer.py:218-231 -> parsers.py:50-72 -> model.py:496-515 ->
     decoder.py:183-198), with ``librosa.stft`` supplied by the numpy framing + rFFT stand-in of tools/gen_golden_surface.py
     (librosa is absent; stated as such);
 (b) the reference's modules batched: its ``SpectrogramAudioParser`` per clip, then ``model(x, lens)`` + ``GreedyDecoder.decode`` at B = 32;
 (c) oracle/torch_port.py (what bench.py's cpu_baseline times on the GPU box's host) on the same clips and threads.

Seeded cfgA weights (danspeech_amd.synthetic, TALKATIVE recipe: what bench.py runs), torch threads = all cores of this container;
median of 3 after a warm-up.  Writes profiles/r04_cpu_reference.txt.

    python tools/time_reference_cpu.py [clips]
"""
import contextlib
import io
import os
import statistics
import sys
import time
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy.signal  # noqa: E402
import scipy.signal.windows as _W  # noqa: E402
for _w in ("hamming", "hann", "blackman", "bartlett"):
    setattr(scipy.signal, _w, getattr(_W, _w))
import torch  # noqa: E402


def _stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect"):
    y = np.asarray(y, dtype=np.float64)
    w = window(win_length) if callable(window) else None
    if center:
        y = np.pad(y, n_fft // 2, mode=pad_mode)
    T = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    return np.fft.rfft(y[idx] * w[:, None], axis=0).astype(np.complex64)


_lib = types.ModuleType("librosa")
_lib.stft, _lib.magphase = _stft, (lambda D: (np.abs(D), None))
sys.modules["librosa"] = _lib
for _n in ("Levenshtein", "wget"):
    sys.modules[_n] = types.ModuleType(_n)

from danspeech import Recognizer  # noqa: E402  (the reference)
from danspeech.deepspeech.model import DeepSpeech  # noqa: E402
from danspeech.deepspeech.decoder import GreedyDecoder  # noqa: E402
from danspeech.audio.parsers import SpectrogramAudioParser  # noqa: E402

from danspeech_amd import synthetic as syn  # noqa: E402
from oracle import torch_port as tp, decoder as od  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 160000
threads = os.cpu_count() or 8
torch.set_num_threads(threads)
cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
labels = syn.DANSPEECH_LABELS
audio_conf = {"sampling_rate": 16000, "window_size": 0.02, "window_stride": 0.01, "window": "hamming", "normalize": True}
model = DeepSpeech(model_name="cfgA", conv_layers=2, rnn_type=torch.nn.GRU, rnn_hidden_size=800, rnn_layers=5, labels=labels,
                   audio_conf=audio_conf, bidirectional=True)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
model.eval()
with contextlib.redirect_stdout(io.StringIO()):
    rec = Recognizer(model=model)
clips = [syn.make_clip(i, N) for i in range(B)]
audio_s = B * N / 16000.0


def median3(fn):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def run_a():
    return [rec.recognize(c) for c in clips]


parser = SpectrogramAudioParser(audio_conf)
greedy = GreedyDecoder(labels=labels, blank_index=labels.index("_"))


def run_b():
    with torch.no_grad():
        feats = [parser.parse_audio(c) for c in clips]
        x = torch.stack(feats).unsqueeze(1)
        lens = torch.full((B,), x.shape[-1], dtype=torch.int)
        probs, out_lens = model(x, lens)
        strings, _ = greedy.decode(probs, out_lens)
    return [s[0] for s in strings]


def run_c():
    x, fr = tp.spectrogram_batch(clips)
    probs, out_lens = tp.forward(sd, cfg, x, fr)
    strings, _ = od.greedy_decode(probs, out_lens, labels, 0)
    return [s[0] for s in strings]


lines = ["# The reference itself beside its port, same container, same cores (tools/time_reference_cpu.py; commit %s)" %
         os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip(),
         "# cfgA (2 conv + 5 x BiGRU 800), seeded TALKATIVE weights, %d x 10 s clips, torch %s, %d threads, median of 3 after a warm-up" % (B, torch.__version__, threads),
         "# STFT of (a) and (b): numpy framing + rFFT stand-in for librosa.stft (absent here)"]
ta, sa = median3(run_a)
lines.append("(a) reference Recognizer.recognize() loop, one clip at a time : %7.2f s per %d clips = %6.1f audio-s/s" % (ta, B, audio_s / ta))
tb, sb = median3(run_b)
lines.append("(b) reference modules batched (parser x %d, model(x, lens), GreedyDecoder): %7.2f s = %6.1f audio-s/s" % (B, tb, audio_s / tb))
tc, sc = median3(run_c)
lines.append("(c) oracle/torch_port.py (bench.py's cpu_baseline) on the same threads  : %7.2f s = %6.1f audio-s/s" % (tc, audio_s / tc))
lines.append("port / reference-batched = %.3f ; port / reference-loop = %.3f ; transcripts (a) == (b): %s, (b) == (c): %s" %
             (tb / tc, ta / tc, sa == sb, sb == sc))
out = "\n".join(lines)
print(out)
with open(os.path.join(ROOT, "profiles", "r04_cpu_reference.txt"), "w") as f:
    f.write(out + "\n")
