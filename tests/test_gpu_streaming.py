"""GPU parity of the chunked unidirectional path through the C ABI:
dsmi_stream_forward against the reference's own streaming model (tests/golden/g8_streaming.npz) and
against oracle/streaming.py; dsmi_features_stream against the oracle's restatement of
InferenceSpectrogramAudioParser.  Tolerance: 1e-4 on probabilities (BASELINE north_star), 2e-5 features."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def native():
    from danspeech_amd import _native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    _native.lib()
    return _native


def _cfg(kind, H, L, ctx):
    return dict(conv_layers=2, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, bidirectional=False, context=ctx)


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
def test_stream_forward_g8(native, golden, kind):
    g = golden("g8_streaming")
    tag = "%s_c2" % kind
    chunks = [int(v) for v in g["chunks_" + tag]]
    H, L, ctx = 32, 3, 6
    sd = syn.make_state_dict(2, kind, H, L, bidirectional=False, context=ctx, seed=81, fc_gain=4.0)
    m = native.NativeModel(_cfg(kind, H, L, ctx), sd)
    st = native.NativeStream(m)
    for utt in range(2):                 # the second utterance runs on the state is_last left behind
        for ci, T in enumerate(chunks):
            x = torch.from_numpy(syn.make_features(1, T, seed=8100 + 100 * utt + ci)).cuda()
            y = st.forward(x, ci == 0, ci == len(chunks) - 1)
            ref = g["probs_%s_u%d_k%d" % (tag, utt, ci)]
            if ci == 0:
                assert y is None and ref.size == 0
            else:
                assert tuple(y.shape[1:]) == ref.shape
                np.testing.assert_allclose(y[0].cpu().numpy(), ref, rtol=0, atol=1e-4)
    st.close(); m.close()


def test_stream_forward_vs_oracle_wide_model_many_chunks(native):
    """H = 200 (several k-blocks per wave, padded to 8), context 20, 12 chunks of the real-time size
    (39 spectrogram frames; Recognizer.py:598-612), odd and even chunk lengths (parity of the packed state)."""
    from oracle import streaming as ost
    H, L, ctx = 200, 2, 20
    sd = syn.make_state_dict(2, "gru", H, L, bidirectional=False, context=ctx, seed=82, fc_gain=4.0)
    cfg = _cfg("gru", H, L, ctx)
    m = native.NativeModel(cfg, sd)
    st = native.NativeStream(m)
    om = ost.StreamingModel(sd, cfg)
    chunks = [53] + [39, 40, 38, 39, 41, 39, 39, 44, 39, 39] + [17]
    worst = 0.0
    for ci, T in enumerate(chunks):
        x = syn.make_features(1, T, seed=8300 + ci)
        y = st.forward(torch.from_numpy(x).cuda(), ci == 0, ci == len(chunks) - 1)
        ref = om.forward(x, ci == 0, ci == len(chunks) - 1)
        assert (y is None) == (ref is None)
        if ref is not None:
            assert tuple(y.shape) == ref.shape
            worst = max(worst, float(np.abs(y.cpu().numpy() - ref).max()))
    print("streaming max |probs - oracle| = %.3g" % worst)
    assert worst < 1e-4
    # protocol errors: a chunk without is_first on a fresh / finished stream is refused and changes nothing
    with pytest.raises(native.DsmiError):
        st.forward(torch.from_numpy(syn.make_features(1, 39, seed=1)).cuda(), False, False)
    st.close()
    with pytest.raises(native.DsmiError):
        native.NativeStream(native.NativeModel(dict(cfg, bidirectional=True), syn.make_state_dict(2, "gru", 16, 1, seed=3)))
    m.close()


def test_features_stream_vs_oracle(native):
    from oracle import streaming as ost
    fe = native.NativeFrontend()
    p = ost.StreamingParser()
    state = np.zeros(3, dtype=np.float64)
    rng = np.random.default_rng(84)
    for n in (8640, 6250, 6240, 7001, 12000, 6240, 6240, 6240, 6240, 6240, 6240, 900):     # alpha crosses 1.0 at the 10th chunk
        part = np.round(rng.normal(0, 3000, n))
        y = p.frames(part).copy()                    # the host-side sample bookkeeping (parsers.py:112-133)
        p.buffer = p.buffer.copy()
        p2 = ost.StreamingParser(); p2.__dict__.update({k: v for k, v in p.__dict__.items()})
        p2.buffer = None
        ref = p2.parse_audio(y)                      # arithmetic half on exactly these samples
        p.input_mean, p.input_std, p.alpha = p2.input_mean, p2.input_std, p2.alpha
        got = fe.features_stream(torch.from_numpy(y).cuda(), state).cpu().numpy()
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        np.testing.assert_allclose(state, [p.input_mean, p.input_std, p.alpha], rtol=1e-6)
    with pytest.raises(native.DsmiError):
        fe.features_stream(torch.zeros(100, dtype=torch.float64).cuda(), state)
    fe.close()


# ---- the Python surface: DeepSpeech(streaming_inference_model=True), enable_real_time_streaming -------

def _stream_model(name, H, L, ctx, seed, kind="gru"):
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(2, kind, H, L, bidirectional=False, context=ctx, seed=seed, fc_gain=8.0)
    m = DeepSpeech(name, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, conv_layers=2, context=ctx, bidirectional=False,
                   streaming_inference_model=True).load_state_dict(sd)
    return m, sd, _cfg(kind, H, L, ctx)


def test_streaming_model_object_and_package_round_trip(tmp_path):
    from danspeech_amd.deepspeech.model import DeepSpeech
    from danspeech_amd.errors.model_errors import ConvError
    from oracle import streaming as ost
    m, sd, cfg = _stream_model("stream-pkg", 48, 2, 8, seed=85)
    import torch as _t
    path = str(tmp_path / "stream.pth")
    _t.save(m.serialize(), path)
    m2 = DeepSpeech.load_model(path).to("cuda")
    assert m2.streaming_model and not m2.bidirectional
    om = ost.StreamingModel(sd, cfg)
    for ci, T in enumerate([60, 39, 39, 20]):
        x = syn.make_features(1, T, seed=8500 + ci)
        y = m2(_t.from_numpy(x), ci == 0, ci == 3)
        ref = om.forward(x, ci == 0, ci == 3)
        assert (y is None) == (ref is None)
        if ref is not None:
            np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=1e-4)
    with pytest.raises(ConvError):
        DeepSpeech("bad", conv_layers=3, streaming_inference_model=True)


@pytest.mark.parametrize("string_parts", [True, False])
def test_real_time_streaming_over_a_recording_equals_oracle(string_parts):
    from danspeech_amd import Recognizer
    from oracle import streaming as ost
    m, sd, cfg = _stream_model("stream-rt", 64, 2, 20, seed=86)
    rec = Recognizer()
    rec.enable_real_time_streaming(streaming_model=m, string_parts=string_parts)
    audio = syn.make_clip(5, 16000 * 4 + 333)
    got = list(rec.stream_recording(audio, chunk_samples=1024))
    # the same chunking against the oracle pipeline
    o = ost.StreamingRecognizer(sd, cfg, syn.DANSPEECH_LABELS, string_parts=string_parts)
    general = 160 * 2 + 160 * ((20 - 1) * 2 - 1)
    first = general + 160 * 15
    want, data, pos, first_pass = [], audio[:0], 0, True
    while pos < len(audio):
        part = audio[pos:pos + 1024]; pos += len(part)
        last = pos >= len(audio)
        data = np.concatenate((data, part))
        out = None
        if first_pass:
            if not last and len(data) >= first:
                out = o.streaming_transcribe(data, False, True); first_pass = False; data = audio[:0]
        elif last or len(data) >= general:
            out = o.streaming_transcribe(data, last, False); data = audio[:0]
        if out:
            want.append((last, out))
    assert got == want and len(got) >= 3 and got[-1][0] is True
    rec.disable_real_time_streaming()
    assert type(rec.danspeech_recognizer.audio_parser).__name__ == "SpectrogramAudioParser"


def test_streaming_final_text_from_secondary_model_and_from_lm(tmp_path):
    """is_last hands the collected spectrograms to the secondary (bidirectional) model, or the collected
    probabilities to the LM decoder (DanSpeechRecognizer.py:190-210)."""
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.model import DeepSpeech
    m, sd, cfg = _stream_model("stream-sec", 64, 2, 20, seed=87)
    sd2 = syn.make_state_dict(2, "gru", 64, 2, seed=88, fc_gain=8.0)
    second = DeepSpeech("second", rnn_hidden_size=64, rnn_layers=2).load_state_dict(sd2)
    audio = syn.make_clip(6, 16000 * 3)
    rec = Recognizer()
    rec.enable_real_time_streaming(streaming_model=m, secondary_model=second)
    out = list(rec.stream_recording(audio, chunk_samples=2048))
    assert out[-1][0] is True
    # the secondary model saw the concatenation of the streaming parser's spectrograms
    from danspeech_amd.audio.parsers import InferenceSpectrogramAudioParser
    p = InferenceSpectrogramAudioParser(m.audio_conf)
    general = 160 * 2 + 160 * 37
    specs, data, pos, first_pass = [], audio[:0], 0, True
    while pos < len(audio):
        part = audio[pos:pos + 2048]; pos += len(part)
        last = pos >= len(audio)
        data = np.concatenate((data, part))
        if first_pass:
            if not last and len(data) >= general + 2400:
                specs.append(p.parse_audio(data, False)); first_pass = False; data = audio[:0]
        elif last or len(data) >= general:
            s = p.parse_audio(data, last)
            if len(s):
                specs.append(s)
            data = audio[:0]
    final = torch.cat(specs, dim=1)
    probs, sizes = second.to("cuda")(final.view(1, 1, final.size(0), final.size(1)), torch.IntTensor([final.size(1)]))
    from danspeech_amd.deepspeech.decoder import GreedyDecoder
    want = GreedyDecoder(syn.DANSPEECH_LABELS, blank_index=syn.DANSPEECH_LABELS.index("_")).decode(probs, sizes)[0][0][0]
    assert out[-1][1] == want
    rec.disable_real_time_streaming()
    # LM decoder on the concatenated streaming probabilities
    path = str(tmp_path / "lm.arpa")
    syn.make_arpa(path, order=3, n_words=300, seed=13, ngrams_per_order=800)
    rec2 = Recognizer(model=_stream_model("stream-lm", 64, 2, 20, seed=87)[0], lm=path)
    rec2.enable_real_time_streaming(streaming_model=rec2.danspeech_recognizer.model)
    out2 = list(rec2.stream_recording(audio, chunk_samples=2048))
    assert out2[-1][0] is True and isinstance(out2[-1][1], str)


def test_stream_forward_cpu_streaming_rnn_shape_vs_oracle(native):
    """The shape of pretrained_models.CPUStreamingRNN (2 conv, 5 x GRU 800 unidirectional, context 20)
    over the real-time chunk sizes: first pass 54 spectrogram frames, then 39 per pass, short last pass."""
    from oracle import streaming as ost
    H, L, ctx = 800, 5, 20
    sd = syn.make_state_dict(2, "gru", H, L, bidirectional=False, context=ctx, seed=89, fc_gain=6.0)
    cfg = _cfg("gru", H, L, ctx)
    m = native.NativeModel(cfg, sd)
    st = native.NativeStream(m)
    om = ost.StreamingModel(sd, cfg)
    chunks = [54, 39, 39, 39, 39, 21]
    worst = 0.0
    for ci, T in enumerate(chunks):
        x = syn.make_features(1, T, seed=8900 + ci)
        y = st.forward(torch.from_numpy(x).cuda(), ci == 0, ci == len(chunks) - 1)
        ref = om.forward(x, ci == 0, ci == len(chunks) - 1)
        assert (y is None) == (ref is None)
        if ref is not None:
            assert tuple(y.shape) == ref.shape
            worst = max(worst, float(np.abs(y.cpu().numpy() - ref).max()))
    print("CPUStreamingRNN shape: max |probs - oracle| = %.3g" % worst)
    assert worst < 1e-4
    st.close(); m.close()
