#!/usr/bin/env python3
"""Pins of the reference's OWN lines around its absent third parties (build container only; data-only fixtures).

The reference delegates the STFT to librosa and the beam search to ctcdecode; neither is installable here, so their
arithmetic stays unpinned.  Everything the reference itself does around those calls can be pinned by running it with
stand-ins that carry no reference logic:

* G10  ``SpectrogramAudioParser.parse_audio`` and ``InferenceSpectrogramAudioParser.parse_audio``
       (/root/reference/danspeech/audio/parsers.py:50-72,102-164) with ``librosa.stft`` / ``librosa.magphase`` provided by the
       documented numpy restatement: pins log1p, the float32 conversion, torch's unbiased std, the streaming parser's hop
       carry-over, its drifting statistics and its short-last-part rule.  THE STFT ITSELF STAYS UNPINNED.
* G11  ``Recognizer(model, lm, ...)``, ``update_model``, ``update_decoder(...)`` and ``recognize(...)`` sequences
       (/root/reference/danspeech/Recognizer.py:39-130, DanSpeechRecognizer.py:14-95,218-231, deepspeech/decoder.py:91-144)
       with ``ctcdecode.CTCBeamDecoder`` replaced by a recorder (tests/_fake_ctc.py): the constructor-argument trace, what
       ``decode`` is handed, the prints, the warnings and what ``recognize`` returns.
* G12  a ``.pth`` package that the reference's ``DeepSpeech.load_model`` (model.py:599-624) loads.

    python tools/gen_golden_surface.py            # rewrites tests/golden/g10_parsers.npz, g11_surface.json/.npz, g12_package.pth
"""
import contextlib
import io
import json
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy.signal  # noqa: E402
import scipy.signal.windows as _W  # noqa: E402
for _w in ("hamming", "hann", "blackman", "bartlett"):
    setattr(scipy.signal, _w, getattr(_W, _w))
import torch  # noqa: E402

from _fake_ctc import fake_beams  # noqa: E402


# ---- librosa stand-in: framing + rFFT only (what oracle/features.py documents; no reference logic) ---------------------
def _stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect"):
    y = np.asarray(y, dtype=np.float64)
    w = window(win_length) if callable(window) else None          # a callable window is evaluated as window(win_length): symmetric
    if center:
        y = np.pad(y, n_fft // 2, mode=pad_mode)
    T = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    return np.fft.rfft(y[idx] * w[:, None], axis=0).astype(np.complex64)


def _magphase(D):
    mag = np.abs(D)
    return mag, None


_lib = types.ModuleType("librosa")
_lib.stft, _lib.magphase = _stft, _magphase
sys.modules["librosa"] = _lib
for _n in ("Levenshtein", "wget"):
    sys.modules[_n] = types.ModuleType(_n)

# ---- ctcdecode stand-in: records how it is built and called ----------------------------------------------------------------
TRACE = []


class _RecordingCTCBeamDecoder(object):
    def __init__(self, *args, **kwargs):
        self.args = args
        names = ("labels", "model_path", "alpha", "beta", "cutoff_top_n", "cutoff_prob", "beam_width", "num_processes", "blank_id")
        TRACE.append({"event": "ctor", "positional": len(args), "kwargs": sorted(kwargs),
                      "args": {n: (list(a) if n == "labels" and not isinstance(a, str) else a) for n, a in zip(names, args)}})

    def decode(self, probs, sizes=None):
        TRACE.append({"event": "decode", "probs_shape": list(probs.shape), "probs_dtype": str(probs.dtype), "probs_device": str(probs.device),
                      "sizes": None if sizes is None else [int(v) for v in sizes]})
        tok, steps, lens, scores = fake_beams(probs.detach().numpy(), None if sizes is None else sizes.numpy(), self.args[6], self.args[8])
        return torch.from_numpy(tok), torch.from_numpy(scores), torch.from_numpy(steps), torch.from_numpy(lens)


_ctc = types.ModuleType("ctcdecode")
_ctc.CTCBeamDecoder = _RecordingCTCBeamDecoder
sys.modules["ctcdecode"] = _ctc

from danspeech import Recognizer  # noqa: E402
from danspeech.audio.parsers import SpectrogramAudioParser, InferenceSpectrogramAudioParser  # noqa: E402
from danspeech.deepspeech.model import DeepSpeech, supported_rnns  # noqa: E402
from danspeech.audio import load_audio  # noqa: E402

from danspeech_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(4)


def g10():
    out = {}
    wav = load_audio(os.path.join(OUT, "u0013002.wav"))
    clips = {"wav": wav, "c0": syn.make_clip(0, 16000), "c1": syn.make_clip(1, 4321), "short": syn.make_clip(2, 700)}
    for name, y in clips.items():
        for normalize in (True, False):
            p = SpectrogramAudioParser({"normalize": normalize})
            s = p.parse_audio(y)
            assert s.dtype == torch.float32
            out["spect_%s_%d" % (name, normalize)] = s.numpy()
    # the streaming parser: one utterance in ragged parts, then a second utterance on the same object
    plan = [("c0", [2400, 1000, 1777, 160, 3333, 319]), ("c1", [1500, 1500, 1321])]
    p = InferenceSpectrogramAudioParser()
    for ui, (name, parts) in enumerate(plan):
        y, pos = clips[name], 0
        for k, n in enumerate(parts):
            last = k == len(parts) - 1
            s = p.parse_audio(y[pos:pos + n], is_last=last)
            pos += n
            key = "stream_u%d_p%d" % (ui, k)
            out[key] = np.zeros((0, 0), dtype=np.float32) if isinstance(s, list) else s.numpy()
            out[key + "_state"] = np.array([p.input_mean, p.input_std, p.alpha], dtype=np.float64)
        p.reset()
    out["stream_plan"] = np.array(json.dumps(plan))
    path = os.path.join(OUT, "g10_parsers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


def _ref_model(name, labels, sd, cfg):
    m = DeepSpeech(name, rnn_type=supported_rnns[cfg["rnn_type"]], labels=labels, rnn_hidden_size=cfg["rnn_hidden_size"],
                   rnn_layers=cfg["rnn_layers"], bidirectional=True, context=20, conv_layers=cfg["conv_layers"])
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.eval()
    return m


def g11():
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=32, rnn_layers=2, bidirectional=True, context=20)
    labels = syn.DANSPEECH_LABELS
    sd = syn.make_state_dict(2, "gru", 32, 2, seed=77, **syn.TALKATIVE)
    other = "_'abcdefghijklmnopqrstuvwxyzæøåé "              # a second alphabet of the same size (the head's width is fixed)
    clips = [syn.make_clip(10, 24000), syn.make_clip(11, 17000)]
    script = [
        ["new", {"model": "m1"}],
        ["recognize", 0, False], ["recognize", 0, True],
        ["update_decoder", {"lm": "/some/dsl_3gram.klm"}],
        ["recognize", 1, False], ["recognize", 1, True],
        ["update_decoder", {"alpha": 1.2, "beta": 0.15, "beam_width": 10}],
        ["update_decoder", {}], ["update_decoder", {"alpha": 0, "beta": None}],
        ["recognize", 0, True],
        ["update_decoder", {"lm": "/some/dsl_3gram.klm"}],
        ["update_model", "m2"],                                # same shapes, another alphabet
        ["recognize", 0, False],
        ["update_decoder", {"lm": "greedy"}],
        ["recognize", 1, False],
        ["new", {"model": "m1", "lm": "/other/lm.klm", "alpha": 0.9, "beta": 0.4, "beam_width": 5}],
        ["recognize", 1, True],
    ]
    models = {"m1": _ref_model("golden-m1", labels, sd, cfg), "m2": _ref_model("golden-m2", other, sd, cfg)}
    events, probs_of = [], {}
    rec = None
    for op in script:
        TRACE.clear()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            result = None
            if op[0] == "new":
                kw = dict(op[1])
                rec = Recognizer(model=models[kw.pop("model")], **kw)
            elif op[0] == "recognize":
                result = rec.recognize(clips[op[1]], show_all=op[2])
            elif op[0] == "update_decoder":
                rec.update_decoder(**op[1])
            elif op[0] == "update_model":
                rec.update_model(models[op[1]])
        eng = rec.danspeech_recognizer
        events.append({"op": op, "stdout": buf.getvalue(), "warnings": [type(x.message).__name__ + ": " + str(x.message) for x in w],
                       "result": result, "result_type": type(result).__name__, "trace": [dict(t) for t in TRACE],
                       "state": {"lm": eng.lm, "alpha": eng.alpha, "beta": eng.beta, "beam_width": eng.beam_width,
                                 "decoder": type(eng.decoder).__name__, "decoder_labels": "".join(eng.decoder.labels),
                                 "labels": "".join(eng.labels)}})
    # the model's outputs for the two clips (what the decoders were handed), for the replay's stand-in model
    with torch.no_grad():
        for k, y in enumerate(clips):
            s = SpectrogramAudioParser(models["m1"].audio_conf).parse_audio(y)
            p, n = models["m1"](s.view(1, 1, s.size(0), s.size(1)), torch.IntTensor([s.size(1)]))
            probs_of["probs%d" % k] = p.numpy()
            probs_of["sizes%d" % k] = n.numpy().astype(np.int32)
    with open(os.path.join(OUT, "g11_surface.json"), "w", encoding="utf-8") as f:
        json.dump({"cfg": cfg, "seed": 77, "labels": labels, "other_labels": other, "clip_ids": [[10, 24000], [11, 17000]],
                   "events": events}, f, ensure_ascii=False, indent=1)
    np.savez_compressed(os.path.join(OUT, "g11_surface.npz"), **probs_of)
    print("wrote g11_surface.json/.npz:", len(events), "events")


def g12():
    """A package in the layout load_model reads (model.py:607-619): tiny -- 1 conv layer at sampling_rate 100 gives a 32-wide
    first recurrent layer."""
    audio_conf = dict(sampling_rate=100, window_size=0.02, window_stride=0.01, window="hamming")
    sd = syn.make_state_dict(1, "lstm", 8, 2, seed=12, sample_rate=100)
    package = {"model_name": "g12", "rnn_hidden_size": 8, "rnn_layers": 2, "labels": syn.DANSPEECH_LABELS, "audio_conf": audio_conf,
               "rnn_type": "lstm", "bidirectional": True, "conv_layers": 1, "context": 20, "streaming_model": False,
               "state_dict": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}}
    path = os.path.join(OUT, "g12_package.pth")
    torch.save(package, path)
    m = DeepSpeech.load_model(path)                              # the reference's own loader accepts it
    assert m.model_name == "g12" and len(m.rnns) == 2 and m.labels == syn.DANSPEECH_LABELS
    for k, v in m.state_dict().items():
        if k in sd:
            assert np.array_equal(v.numpy(), sd[k]), k
    m.eval()                                                     # as DanSpeechRecognizer.update_model does (DanSpeechRecognizer.py:50)
    x = torch.from_numpy(syn.make_features(2, 30, n_freq=2, seed=3))
    with torch.no_grad():
        p, n = m(x, torch.IntTensor([30, 30]))
    np.savez_compressed(os.path.join(OUT, "g12_package_forward.npz"), x=x.numpy(), probs=p.numpy(), out_lens=n.numpy().astype(np.int32))
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    g10()
    g11()
    g12()
