"""The reference's own lines around its absent third parties, replayed on the CPU (fixtures from tools/gen_golden_surface.py,
which ran the reference with stand-ins for librosa and ctcdecode that carry none of its logic):

* G10: the oracle's parsers reproduce what the reference's ``SpectrogramAudioParser`` / ``InferenceSpectrogramAudioParser``
  make of the same STFT (log1p, float32, unbiased std; hop carry-over, drifting statistics, the short-last-part rule);
* G11: ``Recognizer`` construction, ``update_model``, ``update_decoder`` and ``recognize`` -- prints, warnings, return
  shapes, and the arguments the beam decoder is built with -- against the same stand-in decoder;
* G12: a ``.pth`` package the reference's ``load_model`` accepts loads here with the same contents.
"""
import contextlib
import io
import json
import os
import warnings

import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from _fake_ctc import fake_beams

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


# ---------------------------------------------------------------------------------------------------------------- G10
def test_g10_spectrogram_parser_post_stft_arithmetic(golden):
    from danspeech_amd.audio import load_audio
    from oracle import features as of
    g = golden("g10_parsers")
    clips = {"wav": load_audio(os.path.join(GOLD, "u0013002.wav")), "c0": syn.make_clip(0, 16000), "c1": syn.make_clip(1, 4321),
             "short": syn.make_clip(2, 700)}
    for name, y in clips.items():
        for normalize in (True, False):
            want = g["spect_%s_%d" % (name, normalize)]
            got = of.spectrogram(y, normalize=normalize)
            assert got.shape == want.shape and got.dtype == np.float32
            # same STFT values on both sides: what is compared is log1p / float32 / mean / unbiased std (torch's in-place
            # float32 arithmetic against numpy's: a few ulp)
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)


def test_g10_streaming_parser_carry_over_and_statistics(golden):
    from oracle.streaming import StreamingParser
    g = golden("g10_parsers")
    plan = json.loads(str(g["stream_plan"]))
    clips = {"c0": syn.make_clip(0, 16000), "c1": syn.make_clip(1, 4321)}
    p = StreamingParser()
    for ui, (name, parts) in enumerate(plan):
        y, pos = clips[name], 0
        for k, n in enumerate(parts):
            last = k == len(parts) - 1
            s = p.parse_audio(y[pos:pos + n], is_last=last)
            pos += n
            want = g["stream_u%d_p%d" % (ui, k)]
            state = g["stream_u%d_p%d_state" % (ui, k)]
            if want.size == 0:
                assert len(s) == 0                      # a closing part shorter than one window: nothing, and a reset
            else:
                assert s.shape == want.shape
                np.testing.assert_allclose(np.asarray(s), want, rtol=0, atol=3e-6)
            np.testing.assert_allclose([p.input_mean, p.input_std, p.alpha], state, rtol=1e-6, atol=1e-7)
        p.reset()


# ---------------------------------------------------------------------------------------------------------------- G11
class _ReplayModel(object):
    """Stands in for DeepSpeech on the CPU: hands out the probabilities the REFERENCE model produced for the clip."""
    audio_conf = {"sampling_rate": 16000, "window_size": 0.02, "window_stride": 0.01, "window": "hamming", "normalize": True}
    device = "cuda:0"

    def __init__(self, name, labels, outputs):
        self.model_name, self.labels, self.outputs, self.current = name, labels, outputs, None

    def to(self, device):
        return self

    def eval(self):
        return self

    def enqueue(self, feats, frames):
        import torch
        p, n = self.outputs[self.current]
        return torch.from_numpy(p), torch.from_numpy(n)

    def collect(self):
        return False


class _RecordingNativeDecoder(object):
    """Stands in for _native.NativeDecoder (the GPU handle): records how it is configured, answers with the stand-in search."""
    log = []

    def __init__(self, labels, blank_index=0, device=0):
        self.labels, self.blank, self.lm = labels, blank_index, None
        self._pending = None

    def set_lm(self, path, alpha, beta):
        self.lm = (path, alpha, beta)

    def greedy(self, probs, sizes):
        tok, steps, lens, _ = fake_beams(probs.numpy(), sizes, 1, self.blank)
        return [(tok[b, 0, :lens[b, 0]], steps[b, 0, :lens[b, 0]]) for b in range(tok.shape[0])]

    def beam_enqueue(self, probs, sizes=None, beam_width=64, cutoff_top_n=40, cutoff_prob=1.0):
        _RecordingNativeDecoder.log.append({"labels": "".join(self.labels), "model_path": self.lm[0], "alpha": self.lm[1], "beta": self.lm[2],
                                            "cutoff_top_n": cutoff_top_n, "cutoff_prob": cutoff_prob, "beam_width": beam_width,
                                            "blank_id": self.blank, "sizes": None if sizes is None else [int(v) for v in sizes]})
        self._pending = fake_beams(probs.numpy(), sizes, beam_width, self.blank)

    def beam_collect(self):
        out, self._pending = self._pending, None
        return out

    def close(self):
        pass


@pytest.fixture()
def cpu_engine(monkeypatch):
    """The engine's host logic without a GPU: stand-ins for the native decoder handle, the parser and torch's streams."""
    import torch
    from danspeech_amd import _native
    from danspeech_amd.audio import parsers
    from danspeech_amd.DanSpeechRecognizer import DanSpeechRecognizer as Eng
    monkeypatch.setattr(_native, "NativeDecoder", _RecordingNativeDecoder)
    monkeypatch.setattr(parsers.SpectrogramAudioParser, "parse_batch",
                        lambda self, recs: (torch.zeros(len(recs), 1, 161, 4), np.full(len(recs), 4, dtype=np.int32)))
    monkeypatch.setattr(Eng, "_side_stream", lambda self, name: None)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    _RecordingNativeDecoder.log = []


def test_g11_recognizer_plumbing_replays_the_reference(cpu_engine):
    from danspeech_amd import Recognizer
    gold = json.load(open(os.path.join(GOLD, "g11_surface.json"), encoding="utf-8"))
    z = np.load(os.path.join(GOLD, "g11_surface.npz"))
    outputs = {k: (z["probs%d" % k], z["sizes%d" % k]) for k in (0, 1)}
    models = {"m1": _ReplayModel("golden-m1", gold["labels"], outputs), "m2": _ReplayModel("golden-m2", gold["other_labels"], outputs)}
    rec, deviations = None, 0
    for ev in gold["events"]:
        op = ev["op"]
        _RecordingNativeDecoder.log = []
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            result = None
            if op[0] == "new":
                kw = dict(op[1])
                rec = Recognizer(model=models[kw.pop("model")], **kw)
            elif op[0] == "recognize":
                rec.danspeech_recognizer.model.current = op[1]
                result = rec.recognize(syn.make_clip(*gold["clip_ids"][op[1]]), show_all=op[2])
            elif op[0] == "update_decoder":
                rec.update_decoder(**op[1])
            elif op[0] == "update_model":
                rec.update_model(models[op[1]])
        eng = rec.danspeech_recognizer
        # prints: the reference says "Using device: cpu" on its CPU path; this build has only the GPU
        assert buf.getvalue() == ev["stdout"].replace("Using device: cpu", "Using device: cuda"), op
        assert [type(x.message).__name__ + ": " + str(x.message) for x in w] == ev["warnings"], op
        st = ev["state"]
        assert (eng.lm, eng.alpha, eng.beta, eng.beam_width, type(eng.decoder).__name__) == \
            (st["lm"], st["alpha"], st["beta"], st["beam_width"], st["decoder"]), op
        assert "".join(eng.labels) == st["labels"], op
        stale = st["decoder_labels"] != st["labels"]
        if stale:
            # THE documented deviation: after update_model with a new alphabet the reference keeps decoding with
            # the old decoder's labels (DanSpeechRecognizer.py:48-56 sets self.labels before comparing); here the decoder follows
            assert "".join(eng.decoder.labels) == st["labels"]
            deviations += 1
        else:
            assert "".join(eng.decoder.labels) == st["decoder_labels"], op
        # what recognize() returns: a string, or the list of all beams
        assert type(result).__name__ == ev["result_type"], op
        if not stale:
            assert result == ev["result"], op
        # the beam decoder's construction as ctcdecode would have seen it (decoder.py:99-100, positional) and its call
        ctor = [t for t in ev["trace"] if t["event"] == "ctor"]
        calls = [t for t in ev["trace"] if t["event"] == "decode"]
        assert len(_RecordingNativeDecoder.log) == len(calls), op
        if ctor:
            assert ctor[-1]["positional"] == 9 and not ctor[-1]["kwargs"]
            d, a = eng.decoder, ctor[-1]["args"]
            assert (d.lm_path, d.alpha, d.beta, d.cutoff_top_n, d.cutoff_prob, d.beam_width, d.num_processes, d.blank_index) == \
                (a["model_path"], a["alpha"], a["beta"], a["cutoff_top_n"], a["cutoff_prob"], a["beam_width"], a["num_processes"], a["blank_id"])
            assert "".join(d.labels) == "".join(a["labels"])
        for mine, theirs in zip(_RecordingNativeDecoder.log, calls):
            assert mine["sizes"] == theirs["sizes"]
            if not stale:
                d = eng.decoder
                assert (mine["model_path"], mine["alpha"], mine["beta"], mine["cutoff_top_n"], mine["cutoff_prob"], mine["beam_width"], mine["blank_id"]) == \
                    (d.lm_path, d.alpha, d.beta, d.cutoff_top_n, d.cutoff_prob, d.beam_width, d.blank_index)
    assert deviations == 2        # the update_model with a new alphabet and the recognize() right after it


# ---------------------------------------------------------------------------------------------------------------- G12
def test_g12_package_accepted_by_the_reference_loads_here():
    from danspeech_amd.deepspeech.model import DeepSpeech
    from danspeech_amd.pretrained_models import CustomModel
    path = os.path.join(GOLD, "g12_package.pth")
    m = DeepSpeech.load_model(path)
    assert (m.model_name, m.rnn_type, m.rnn_hidden_size, m.rnn_layers, m.conv_layers, m.bidirectional, m.context) == \
        ("g12", "lstm", 8, 2, 1, True, 20)
    assert m.labels == syn.DANSPEECH_LABELS and m.audio_conf["sampling_rate"] == 100 and not m.streaming_model
    sd = syn.make_state_dict(1, "lstm", 8, 2, seed=12, sample_rate=100)
    assert set(sd) <= set(m.state_dict())
    for k, v in sd.items():
        np.testing.assert_array_equal(np.asarray(m.state_dict()[k]), v)
    assert CustomModel(path).model_name == "g12"
