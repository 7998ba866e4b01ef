"""CPU-side tests of the drop-in surface: audio loading against the reference's golden (G6),
the update_decoder state machine, plugin registries, sharding plan, cache/md5 helpers.
No kernels run here (the compute path needs the MI355X)."""
import hashlib
import os

import numpy as np
import pytest

from danspeech_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "golden", "u0013002.wav")


def test_load_audio_matches_reference_g6(golden):
    from danspeech_amd.audio import load_audio
    g = golden("g6_audio")
    assert hashlib.sha256(open(WAV, "rb").read()).hexdigest() == str(g["wav_sha256"])
    y = load_audio(WAV)
    assert y.dtype == np.float64 and len(y) == int(g["n"]) == 66944
    assert y.min() == float(g["vmin"]) == -19241.0 and y.max() == float(g["vmax"]) == 15957.0
    assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == str(g["sha256"])
    # duration / offset follow the reference's 4096-frame chunk loop
    y2 = load_audio(WAV, duration=1.0)
    assert len(y2) == 3 * 4096 and np.array_equal(y2, y[:len(y2)])
    y3 = load_audio(WAV, offset=1.0)
    assert np.array_equal(y3, y[3 * 4096:])


def test_load_audio_wavpcm_is_channel_mean():
    from danspeech_amd.audio import load_audio_wavPCM
    import wave
    with wave.open(WAV, "rb") as w:
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2)
    np.testing.assert_array_equal(load_audio_wavPCM(WAV), raw.mean(axis=1))


def test_plugin_registries_match_reference_artefacts():
    from danspeech_amd import pretrained_models as pm, language_models as lm
    assert sorted(pm.REGISTRY) == sorted(["DanSpeechPrimary", "TestModel", "Baseline", "TransferLearned", "Folketinget",
                                          "EnglishLibrispeech", "CPUStreamingRNN", "GPUStreamingRNN"])
    assert pm.REGISTRY["DanSpeechPrimary"][:2] == ("DanSpeechPrimary.pth", "5bd08282d442e990c37481d5c61cf93c")
    assert pm.REGISTRY["TestModel"][1] == "c21438a33f847a9c8d4e08779e98bf31"
    assert lm.REGISTRY["DSL3gram"] == ("dsl_3gram.klm", "33ca3e2a8db3a036af6d7ad85972dbb0")
    assert len(lm.REGISTRY) == 9
    for name in list(pm.REGISTRY) + ["CustomModel", "get_model_from_string"]:
        assert callable(getattr(pm, name))
    for name in list(lm.REGISTRY) + ["CustomLanguageModel"]:
        assert callable(getattr(lm, name))
    assert lm.CustomLanguageModel("/x/y.klm") == "/x/y.klm"
    assert pm.get_model_from_string("nope") is None


def test_cache_layout_and_md5(tmp_path):
    from danspeech_amd.utils import data_utils as du
    f = tmp_path / "m.pth"
    f.write_bytes(b"hello")
    md5 = hashlib.md5(b"hello").hexdigest()
    assert du.validate_file(str(f), md5) and not du.validate_file(str(f), "0" * 32)
    assert du.get_model("m.pth", "http://x/m.pth", file_hash=md5, cache_dir=str(tmp_path)) == str(f)
    with pytest.raises(RuntimeError):    # absent + no network: fails loudly instead of downloading
        du.get_model("absent.pth", "http://x/absent.pth", file_hash=md5, cache_dir=str(tmp_path))
    assert du.subdir_mapper == {"acoustic_model": "models", "language_model": "lms"}


class _FakeModel:
    """Stands in for DeepSpeech so that the engine's host logic runs without a GPU."""
    audio_conf = {"sampling_rate": 16000, "window_size": 0.02, "window_stride": 0.01, "window": "hamming", "normalize": True}
    labels = syn.DANSPEECH_LABELS
    model_name = "fake"
    device = "cuda:0"

    def to(self, device):
        return self

    def eval(self):
        return self


def test_update_decoder_state_machine(capsys):
    """reference DanSpeechRecognizer.py:58-95 and Recognizer.py:97-130."""
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.decoder import GreedyDecoder, BeamCTCDecoder
    from danspeech_amd.errors.recognizer_errors import ModelNotInitialized
    with pytest.raises(ModelNotInitialized):
        Recognizer(lm="/some/lm.arpa")
    r = Recognizer()
    assert "Using device: cuda" in capsys.readouterr().out
    eng = r.danspeech_recognizer
    assert eng.model is None and eng.decoder is None and eng.lm is None
    assert (eng.alpha, eng.beta, eng.beam_width) == (1.3, 0.2, 64)
    r.update_model(_FakeModel())
    assert "DanSpeech model updated to: fake" in capsys.readouterr().out
    assert isinstance(eng.decoder, GreedyDecoder) and eng.lm == "greedy"
    assert eng.decoder.blank_index == 0 and eng.decoder.space_index == 32
    first = eng.decoder
    r.update_decoder()                       # nothing changed -> same decoder object
    assert eng.decoder is first
    assert "DanSpeech decoder updated " in capsys.readouterr().out
    r.update_decoder(alpha=0, beta=None)     # falsy values are ignored
    assert eng.decoder is first and eng.alpha == 1.3
    r.update_decoder(lm="/some/lm.arpa", alpha=1.2, beta=0.15, beam_width=10)
    d = eng.decoder
    assert isinstance(d, BeamCTCDecoder)
    assert (d.lm_path, d.alpha, d.beta, d.beam_width, d.cutoff_top_n, d.cutoff_prob, d.num_processes) == \
        ("/some/lm.arpa", 1.2, 0.15, 10, 40, 1.0, 6)
    r.update_decoder(lm="/some/lm.arpa")     # same lm -> no rebuild
    assert eng.decoder is d
    r.update_decoder(beam_width=20)
    assert eng.decoder is not d and eng.decoder.beam_width == 20
    r.update_decoder(lm="greedy")
    assert isinstance(eng.decoder, GreedyDecoder)


def test_deepspeech_ctor_contract():
    from danspeech_amd.deepspeech.model import DeepSpeech
    from danspeech_amd.errors.model_errors import ConvError
    with pytest.raises(ConvError, match="0 convolutional layers"):
        DeepSpeech("m", conv_layers=0)
    with pytest.raises(ConvError, match="Maximum amount"):
        DeepSpeech("m", conv_layers=4)
    m = DeepSpeech("m")
    assert m.labels == syn.DANSPEECH_LABELS and m.audio_conf["window"] == "hamming" and m.rnn_hidden_size == 768
    assert m.context == 20 and m.bidirectional and m.conv_layers == 2 and m.rnn_type == "gru"
    import torch
    assert m.get_seq_lens(torch.tensor([1001, 3001])).tolist() == [501, 1501]
    assert DeepSpeech("m", rnn_type=torch.nn.LSTM).rnn_type == "lstm"
    sd = syn.make_state_dict(2, "gru", 16, 2, seed=1)
    m2 = DeepSpeech("m", rnn_hidden_size=16, rnn_layers=2).load_state_dict(sd)
    n_ref = sum(int(np.prod(v.shape)) for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert DeepSpeech.get_param_size(m2) == n_ref
    with pytest.raises(RuntimeError):       # no CPU path: forward without .to('cuda') fails loudly
        m2(torch.zeros(1, 1, 161, 50), torch.tensor([50]))


def test_model_package_roundtrip(tmp_path):
    """serialize() -> load_model() here; the package the REFERENCE's loader accepts is tests/golden/g12_package.pth
    (tests/test_surface_golden.py::test_g12_package_accepted_by_the_reference_loads_here)."""
    import torch
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(2, "lstm", 16, 2, seed=2)
    m = DeepSpeech("pkg", rnn_type="lstm", rnn_hidden_size=16, rnn_layers=2).load_state_dict(sd)
    path = str(tmp_path / "pkg.pth")
    torch.save(m.serialize(), path)
    m2 = DeepSpeech.load_model(path)
    assert (m2.model_name, m2.rnn_type, m2.rnn_hidden_size, m2.rnn_layers) == ("pkg", "lstm", 16, 2)
    for k in sd:
        np.testing.assert_array_equal(np.asarray(m2.state_dict()[k]), sd[k])
    from danspeech_amd.pretrained_models import CustomModel
    assert CustomModel(path).model_name == "pkg"


class _NotATensor(object):
    pass


def test_model_package_with_foreign_objects_is_refused_with_instructions(tmp_path):
    """A .pth that needs a full unpickle is not loaded silently (the reference's torch.load would run it): the error says what
    to do, and the explicit route works."""
    import torch
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(2, "gru", 16, 1, seed=2)
    m = DeepSpeech("pkg", rnn_hidden_size=16, rnn_layers=1).load_state_dict(sd)
    package = m.serialize()
    package["extra"] = _NotATensor()
    path = str(tmp_path / "odd.pth")
    torch.save(package, path)
    with pytest.raises(RuntimeError, match="load_model_package"):
        DeepSpeech.load_model(path)
    m2 = DeepSpeech.load_model_package(torch.load(path, map_location="cpu", weights_only=False))
    assert m2.model_name == "pkg" and m2.rnn_hidden_size == 16


def test_decoder_base_helpers():
    from danspeech_amd.deepspeech.decoder import Decoder
    d = Decoder("_ab ")
    assert d.space_index == 3 and Decoder("_ab").space_index == 3
    assert d.wer("a b a", "a a") == 1 and d.cer("ab a", "abba") == 1


def test_plan_shards():
    from danspeech_amd.parallel import plan_shards
    lens = [5, 9, 1, 7, 3, 8]
    shards = plan_shards(lens, 2)
    assert sorted(np.concatenate(shards).tolist()) == list(range(6))
    for s in shards:
        l = [lens[i] for i in s]
        assert l == sorted(l, reverse=True)
    assert [lens[i] for i in shards[0]] == [9, 7, 3] and [lens[i] for i in shards[1]] == [8, 5, 1]


def test_transcribe_batches_cuts_large_batches_and_restores_the_callers_order():
    """``transcribe_batches``: a caller's batch of more than ``merge_clips`` clips runs as forwards of at most that many, longest
    clips first, and comes back as ONE list in the caller's order; smaller batches pass through untouched.  The pipeline below it
    is replaced by a stand-in that "transcribes" a clip to (its length, its first sample) and checks what it is handed."""
    from danspeech_amd.DanSpeechRecognizer import DanSpeechRecognizer
    eng = object.__new__(DanSpeechRecognizer)
    seen = []

    totals = []

    def forwards(batches, show_all=False, lanes=None, merge_clips=None, total=None):
        totals.append(total)
        for b in batches:
            seen.append([len(c) for c in b])
            yield [(len(c), float(c[0])) for c in b]
    eng._transcribe_forwards = forwards
    rng = np.random.default_rng(5)
    mk = lambda n: [np.full(int(rng.integers(5, 400)), float(rng.integers(0, 1000))) for _ in range(n)]
    batches = [mk(3), mk(150), [], mk(64), mk(65), mk(1)]
    out = list(eng.transcribe_batches(batches, merge_clips=64))
    assert [len(o) for o in out] == [3, 150, 0, 64, 65, 1]
    for b, o in zip(batches, out):
        assert o == [(len(c), float(c[0])) for c in b]
    assert [len(s) for s in seen] == [3, 64, 64, 22, 0, 64, 64, 1, 1]
    assert totals == [9]                                        # a sized source: the pipeline is told how many pieces will come
    for s in seen[1:4] + seen[6:8]:
        assert s == sorted(s, reverse=True)                     # every piece longest first ...
    assert seen[1][-1] >= seen[2][0] and seen[2][-1] >= seen[3][0]      # ... and the pieces of a batch in that order too
    # merge_clips=0 (the strictly sequential mode): nothing is cut
    seen.clear()
    assert [len(o) for o in eng.transcribe_batches([mk(150)], lanes=1, merge_clips=0)] == [150] and [len(s) for s in seen] == [150]
    # a consumer that stops early closes the pipeline below
    closed = []

    def forwards2(batches, **kw):
        try:
            for b in batches:
                yield [0] * len(b)
        finally:
            closed.append(True)
    eng._transcribe_forwards = forwards2
    g = eng.transcribe_batches([mk(200), mk(3)], merge_clips=64)
    next(g)
    g.close()
    assert closed == [True]


def test_pack_pcm_i16_says_exactly_when_a_clip_is_int16():
    """``dsmi_pack_pcm_i16`` (host only): float64 samples that are integers in int16's range come out as int16 and the call says 1;
    one sample that is not -- a fraction, beyond the range on either side, NaN, infinity -- and it says 0."""
    from danspeech_amd import _native
    L = _native.lib()
    rng = np.random.default_rng(3)
    x = np.round(rng.normal(0, 9000, 70001)).clip(-32768, 32767)
    x[:3] = (-32768.0, 32767.0, -0.0)
    d = np.full(len(x), 77, dtype=np.int16)
    assert L.dsmi_pack_pcm_i16(x.ctypes.data, len(x), d.ctypes.data) == 1 and np.array_equal(d.astype(np.float64), x)
    for at in (0, 4095, 4096, 70000):
        for bad in (0.5, -1e-9, 32768.0, -32769.0, 1e300, float("nan"), float("inf"), -float("inf")):
            y = x.copy()
            y[at] = bad
            assert L.dsmi_pack_pcm_i16(y.ctypes.data, len(y), d.ctypes.data) == 0, (at, bad)
    assert L.dsmi_pack_pcm_i16(x.ctypes.data, 0, d.ctypes.data) == 1           # nothing to say about no samples
    assert L.dsmi_pack_pcm_i16(None, 5, d.ctypes.data) == 0
