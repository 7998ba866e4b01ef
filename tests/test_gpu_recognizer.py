"""GPU end-to-end tests through the drop-in Python surface (Recognizer / DanSpeechRecognizer /
DeepSpeech / decoders), i.e. the BASELINE.json configs as parity cases:
  config 1  example clip u0013002.wav, greedy, B=1 through Recognizer.recognize()
  config 2  cfgA greedy batch (bench.py's workload, here checked for correctness on a sub-batch)
  config 3  cfgA + 3-gram LM, beam 64
against the CPU oracle on the same seeded weights (pretrained .pth/.klm artefacts are not
obtainable offline)."""
import os
import warnings

import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "golden", "u0013002.wav")


def _model(name, H, L, seed, kind="gru", conv=2):
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(conv, kind, H, L, seed=seed, fc_gain=8.0)
    m = DeepSpeech(name, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, conv_layers=conv).load_state_dict(sd)
    cfg = dict(conv_layers=conv, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, bidirectional=True, context=20)
    return m, sd, cfg


def _oracle_greedy(sd, cfg, clips):
    from oracle import model as om, features as of, decoder as od
    out = []
    for c in clips:
        x = of.spectrogram(c)[None, None]
        p, ol = om.forward(sd, cfg, x, [x.shape[-1]])
        out.append(od.greedy_decode(p, ol, syn.DANSPEECH_LABELS, 0)[0][0][0])
    return out


def test_config1_example_clip_greedy(capsys):
    """TestModel shape (2 conv, 5 x 400; reference test_model.py:14-16) with seeded weights."""
    from danspeech_amd import Recognizer
    from danspeech_amd.audio import load_audio
    from danspeech_amd.DanSpeechRecognizer import NoLmInstantiatedWarning
    model, sd, cfg = _model("TestModel", 400, 5, seed=11)
    rec = Recognizer(model=model)
    out = capsys.readouterr().out
    assert "Using device: cuda" in out and "DanSpeech model updated to: TestModel" in out
    audio = load_audio(WAV)
    text = rec.recognize(audio)
    assert isinstance(text, str)
    assert text == _oracle_greedy(sd, cfg, [audio])[0]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        allb = rec.recognize(audio, show_all=True)
    assert allb == [text] and any(issubclass(x.category, NoLmInstantiatedWarning) for x in w)


def test_recognize_batch_equals_single_and_oracle():
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("small", 64, 3, seed=12)
    rec = Recognizer(model=model)
    clips = [syn.make_clip(i, n) for i, n in enumerate([16000, 40000, 8000, 40000, 23456])]
    batch = rec.recognize_batch(clips)
    single = [rec.recognize(c) for c in clips]
    assert batch == single == _oracle_greedy(sd, cfg, clips)
    assert rec.recognize_batch([]) == []


def test_device_resident_clips_equal_host_arrays():
    """DeviceClips (clips back to back in GPU memory, longest first): the same results as the same clips handed over as host
    arrays, through recognize_batch and the pipelined recognize_batches; the order rule is enforced."""
    from danspeech_amd import Recognizer
    from danspeech_amd.audio.parsers import DeviceClips
    model, sd, cfg = _model("small", 64, 3, seed=12)
    rec = Recognizer(model=model)
    clips = [syn.make_clip(i, n) for i, n in enumerate([40000, 40000, 23456, 16000, 8000])]
    want = rec.recognize_batch(clips)
    n = np.array([len(c) for c in clips], dtype=np.int64)
    for dtype in (np.float64, np.int16):
        pcm = torch.from_numpy(np.concatenate(clips).astype(dtype)).cuda()
        dc = DeviceClips(pcm, n)
        assert rec.recognize_batch(dc) == want
        assert list(rec.recognize_batches([dc, dc.part(1, 4), dc])) == [want, want[1:4], want]
    # consecutive batches are merged into one forward: of one sample type only, host and device-resident batches apart, empty ones kept
    d64 = DeviceClips(torch.from_numpy(np.concatenate(clips)).cuda(), n)
    d16 = DeviceClips(torch.from_numpy(np.concatenate(clips).astype(np.int16)).cuda(), n)
    mixed = [d64, d16, clips, [], d16.part(0, 2), d16.part(2, 5), clips[:1], d64]
    assert list(rec.recognize_batches(mixed)) == [want, want, want, [], want[:2], want[2:], want[:1], want]
    assert list(rec.danspeech_recognizer.transcribe_batches(mixed, lanes=1, merge_clips=0)) == [want, want, want, [], want[:2], want[2:], want[:1], want]
    with pytest.raises(ValueError):
        DeviceClips(pcm, n[::-1].copy())
    with pytest.raises(ValueError):
        DeviceClips(pcm[:-1], n)


def test_forwards_in_flight_follow_the_first_forward():
    """Nobody named a lane count: four forwards in flight where a forward is one ring window (up to 64 clips of a model the ring
    kernel takes), two where it is more clips than that or the model's recurrent kernel takes the whole device; results equal
    either way, and a count that was named is taken as it is."""
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("small", 64, 3, seed=12)
    clips = [syn.make_clip(i, n) for i, n in enumerate([9000, 8000, 8000, 7000, 6000])]
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    want = rec.recognize_batch(clips)
    assert eng._lanes_that_pay(4, 64) == 4 and eng._lanes_that_pay(4, 65) == 2 and eng._lanes_that_pay(1, 65) == 1
    big = clips * 14                                             # 70 clips: one forward of more than a window where the caller allows it
    assert list(eng.transcribe_batches([big, clips, big], merge_clips=128)) == [want * 14, want, want * 14]
    assert len(eng._replicas) == 1
    # by default such a batch is cut into forwards of at most 64 clips (round 6), each one ring window: four in flight
    assert list(rec.recognize_batches([big, clips, big])) == [want * 14, want, want * 14]
    assert len(eng._replicas) == 3
    assert list(rec.recognize_batches([clips] * 6)) == [want] * 6
    assert len(eng._replicas) == 3
    wide, _, _ = _model("wide", 912, 1, seed=13)                 # 912 units: no ring form
    rec2 = Recognizer(model=wide)
    one = rec2.recognize_batch(clips)
    assert list(rec2.recognize_batches([clips] * 5)) == [one] * 5 and len(rec2.danspeech_recognizer._replicas) == 1
    assert list(rec2.danspeech_recognizer.transcribe_batches([clips] * 5, lanes=3)) == [one] * 5
    assert len(rec2.danspeech_recognizer._replicas) == 2


def test_sequential_mode_reads_a_batch_only_after_the_previous_result():
    """``lanes=1, merge_clips=0`` is what the docstring promises a feedback-driven source: batch k + 1 is asked for only after
    result k has been yielded, on the caller's own thread (the reference's ``recognize()`` is synchronous,
    danspeech/Recognizer.py:82-95).  Both decoders: a beam search has one more job in flight in the pipelined modes."""
    import threading
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.decoder import BeamCTCDecoder
    model, sd, cfg = _model("small", 64, 3, seed=12)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    clips = [syn.make_clip(i, n) for i, n in enumerate([9000, 8000, 7000])]
    want = rec.recognize_batch(clips)
    caller = threading.get_ident()
    for searching in (False, True):
        if searching:
            eng.decoder = BeamCTCDecoder(labels=eng.labels, beam_width=8, blank_index=0)
            want = rec.recognize_batch(clips)
        seen = []

        def source():
            for k in range(5):
                assert len(seen) == k, (k, len(seen))          # result k - 1 is out before batch k is asked for
                assert threading.get_ident() == caller
                yield clips if k != 2 else clips[:1]

        for out in eng.transcribe_batches(source(), lanes=1, merge_clips=0):
            seen.append(out)
        assert seen == [want, want, want[:1], want, want]


def test_staging_slots_are_sized_by_the_largest_forward_of_the_process():
    """A lane's pinned staging slot that has met only small forwards must not re-pin in the middle of a later call when its first
    large forward arrives (82 MB: 16 ms on the staging thread and a blocking upload behind it): every slot in use is as large as
    the largest forward any parser of the process has staged."""
    from danspeech_amd import Recognizer
    from danspeech_amd.audio.parsers import SpectrogramAudioParser
    model, sd, cfg = _model("small", 64, 3, seed=12)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    big = [syn.make_clip(i, 20000) for i in range(8)]
    small = big[:2]
    want_big, want_small = rec.recognize_batch(big), rec.recognize_batch(small)
    # three lanes, no merging: the large batch lands on lane 0 only, lanes 1 and 2 see small ones
    outs = list(eng.transcribe_batches([big, small, small, small, small, small], lanes=3, merge_clips=0))
    assert outs == [want_big] + [want_small] * 5
    high = SpectrogramAudioParser._stage_high
    assert high >= sum(len(c) for c in big) * 8
    parsers = [eng.audio_parser] + [r[1] for r in eng._replicas]
    used = [sl for p in parsers for sl in (getattr(p, "_slots", None) or []) if sl["buf"] is not None]
    assert len(used) >= 4 and all(sl["buf"].numel() >= high for sl in used[1:]), [sl["buf"].numel() for sl in used]


def test_short_calls_take_other_kernel_forms_and_say_the_same():
    """A call of one, two, three or four batches from a LIST (a sized source: the pipeline knows what is to come) runs its forwards
    on a lone batch's kernels or on two ring windows each (``dsmi_model_set_ring_windows``), a generator of the same batches on the
    forms of a long call: every result equals the single call's, nothing is recomputed, and the hints are given back at the end."""
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("small", 128, 3, seed=12)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    clips = [syn.make_clip(i, 9000 + 400 * (i % 5)) for i in range(32)]
    want = rec.recognize_batch(clips)
    assert want[:4] == _oracle_greedy(sd, cfg, clips[:4])
    for n in (1, 2, 3, 4, 6):
        assert list(rec.recognize_batches([clips] * n)) == [want] * n, n
        assert list(rec.recognize_batches(clips for _ in range(n))) == [want] * n, n
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    assert len(handles) == 4 and [h.recompute_count() for h in handles] == [0, 0, 0, 0]
    # the two-window form by hand, against the one-window form: the same probabilities within the parity bound
    from danspeech_amd import _native
    m = _native.NativeModel(cfg, sd)
    fe = _native.NativeFrontend()
    big = clips + clips
    order = np.argsort([-len(c) for c in big], kind="stable")
    n = np.array([len(big[i]) for i in order], dtype=np.int64)
    feat, frames = fe.features(torch.from_numpy(np.concatenate([big[i] for i in order])).cuda(), n)
    m.set_inflight(4)
    p1, ol = m.forward(feat, frames)
    p1 = p1.cpu().numpy()
    m.set_ring_windows(2)
    p2, _ = m.forward(feat, frames)
    p2 = p2.cpu().numpy()
    m.set_ring_windows(0)
    assert m.recompute_count() == 0
    assert max(float(np.abs(p1[b, :ol[b]] - p2[b, :ol[b]]).max()) for b in range(len(big))) < 5e-5
    m.close(); fe.close()


@pytest.mark.parametrize("hidden", [64, 128, 192, 256])
def test_small_models_in_the_pipeline(hidden):
    """Round 6 found the four-wave ring kernel wrong for GRUs of fewer than four k-blocks per wave (H < 224) IN A PIPELINE: with
    several windows of different handles running at once the last tile of a 64-clip window came out wrong now and then (garbage
    transcripts for its clips, from the third call of a process on; cause not found).  Those shapes now run the eight-wave form
    (``rnn_persist_ring4_tiles``); this is the scenario that showed it, repeated: every call's every batch equals the single call."""
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("small", hidden, 3, seed=12)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    eng.pipeline_balance_tail = False                      # (64-clip forwards to the end of a call)
    clips = [syn.make_clip(i, 9000 + 400 * (i % 5)) for i in range(32)]
    want = rec.recognize_batch(clips)
    for n in (6, 8, 6, 8, 6, 8, 6, 8):
        got = list(rec.recognize_batches([clips] * n))
        bad = [(k, i) for k in range(n) for i in range(32) if got[k][i] != want[i]]
        assert not bad, (hidden, n, bad[:8])
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    assert [h.recompute_count() for h in handles] == [0] * len(handles)


def test_a_ring_model_and_a_whole_device_model_in_flight_together():
    """Two engines used at once from two threads: one model the ring kernels take (slot-sized windows, ordered by events) and one
    whose recurrent kernel takes the whole device (turns through the device lock, api.hip) -- the two families are ordered
    against each other by events in enqueue order.  Every batch equals its engine's single call; no hand-off timed out."""
    import threading
    from danspeech_amd import Recognizer
    ring, _, _ = _model("ring", 128, 2, seed=21)
    wide, _, _ = _model("wide", 912, 2, seed=22)
    recs = [Recognizer(model=ring), Recognizer(model=wide)]
    clips = [syn.make_clip(i, 12000 + 300 * (i % 4)) for i in range(24)]
    want = [r.recognize_batch(clips) for r in recs]
    got, errs = [None, None], []

    def run(k):
        try:
            torch.cuda.set_device(0)
            got[k] = list(recs[k].recognize_batches([clips] * 6))
        except Exception as e:          # (reported below: an assertion in a thread would pass unnoticed)
            errs.append(repr(e))
    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    for k in range(2):
        assert got[k] == [want[k]] * 6, k
        eng = recs[k].danspeech_recognizer
        assert all(h.recompute_count() == 0 for h in [eng.model._native] + [r[0]._native for r in eng._replicas])


def test_float64_clips_travel_as_int16_where_that_is_exact():
    """``SpectrogramAudioParser.stage``: float64 clips whose samples are int16 integers (what ``load_audio`` returns for a file,
    reference resources.py:640) are uploaded as int16 -- and give the SAME features, bit for bit, as the float64 upload; one
    fractional sample anywhere in the batch and the whole batch travels as float64."""
    from danspeech_amd.audio.parsers import SpectrogramAudioParser
    clips = [syn.make_clip(i, n) for i, n in enumerate([48000, 40000, 40000, 33333, 16000, 16000, 9000, 8000, 8000])]
    packed, plain = SpectrogramAudioParser(device=0), SpectrogramAudioParser(device=0)
    plain.pack_int16 = False
    a = packed.stage(clips)
    b = plain.stage(clips)
    assert a.itemsize == 2 and b.itemsize == 8
    fa, na = packed.parse_batch(a)
    fb, nb = plain.parse_batch(b)
    torch.cuda.synchronize()
    assert np.array_equal(na, nb) and torch.equal(fa, fb)
    odd = [c.copy() for c in clips]
    odd[6][1234] += 0.25
    c = packed.stage(odd)
    assert c.itemsize == 8
    fc, _ = packed.parse_batch(c)
    fd, _ = plain.parse_batch(plain.stage(odd))
    torch.cuda.synchronize()
    assert torch.equal(fc, fd) and not torch.equal(fc, fa)


def test_wide_model_stream_of_batches_runs_clean():
    """Config 4's width (H = 1200: the tile-walking recurrent kernel, four tiles per workgroup) as a stream of 64-clip batches, with the
    forwards in flight the engine picks and with four: every batch equals the single call, and no hand-off of any handle timed out
    (a recomputed batch would give the same strings)."""
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("wide", 1200, 3, seed=5)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    clips = [syn.make_clip(i, 32000 + 500 * (i % 7)) for i in range(64)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        one = rec.recognize_batch(clips)
        for lanes in (None, 4):
            assert all(out == one for out in eng.transcribe_batches([clips] * 6, lanes=lanes))
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    assert len(handles) == 4 and [h.recompute_count() for h in handles] == [0, 0, 0, 0]
    assert not [x for x in w if "timed out" in str(x.message)]


def test_device_clips_batches_wait_for_their_producer():
    """Device-resident clips are produced by asynchronous work on the caller's stream (an RCCL scatter, a widening copy): every
    second batch of the pipeline runs on a side stream, which must wait for that producer.  Here the producer is a slow chain
    of kernels that ends by writing the real samples over a garbage buffer; batch 1 (the side stream's) read garbage before the
    pipeline ordered itself behind the caller's stream."""
    from danspeech_amd import Recognizer
    from danspeech_amd.audio.parsers import DeviceClips
    model, sd, cfg = _model("small", 64, 3, seed=12)
    rec = Recognizer(model=model)
    clips = [syn.make_clip(i, n) for i, n in enumerate([40000, 40000, 23456, 16000, 8000])]
    want = rec.recognize_batch(clips)
    n = np.array([len(c) for c in clips], dtype=np.int64)
    good = torch.from_numpy(np.concatenate(clips)).cuda()
    spin = torch.randn(4096, 4096, device="cuda")
    for rep in range(3):
        bufs = [torch.full_like(good, 1e4) for _ in range(4)]
        torch.cuda.synchronize()
        for _ in range(40):                    # ~tens of milliseconds of queued work in front of the copies
            spin = torch.tanh(spin @ spin * 1e-3)
        for b in bufs:
            b.copy_(good, non_blocking=True)
        got = list(rec.recognize_batches([DeviceClips(b, n) for b in bufs]))
        assert got == [want] * 4, rep


def test_config3_beam_with_lm_through_recognizer(tmp_path, capsys):
    """cfgA (2 conv, 5 x BiGRU 800) + synthetic 3-gram, alpha=1.3 beta=0.2 beam=64 (engine defaults)."""
    from danspeech_amd import Recognizer
    from danspeech_amd.language_models import CustomLanguageModel
    from oracle import model as om, features as of, beam as ob
    lm_path = str(tmp_path / "syn3.arpa")
    syn.make_arpa(lm_path, order=3, n_words=2000, seed=21, ngrams_per_order=5000)
    model, sd, cfg = _model("cfgA", 800, 5, seed=0)
    rec = Recognizer(model=model, lm=CustomLanguageModel(lm_path))
    assert "DanSpeech decoder updated " in capsys.readouterr().out
    eng = rec.danspeech_recognizer
    assert (eng.alpha, eng.beta, eng.beam_width) == (1.3, 0.2, 64)
    clips = [syn.make_clip(40, 48000), syn.make_clip(41, 30000)]
    beams = rec.recognize_batch(clips, show_all=True)
    scores = eng.decoder.last_scores
    best = rec.recognize_batch(clips)
    scorer = ob.Scorer(1.3, 0.2, lm_path, syn.DANSPEECH_LABELS)
    order = np.argsort([-len(c) for c in clips], kind="stable")
    for pos, i in enumerate(order):
        x = of.spectrogram(clips[i])[None, None]
        p, ol = om.forward(sd, cfg, x, [x.shape[-1]])
        ref = ob.ctc_beam_search(p[0, :ol[0]].astype(np.float64), syn.DANSPEECH_LABELS, 64, scorer=scorer)
        ref_strings = ["".join(syn.DANSPEECH_LABELS[c] for c in r[1]) for r in ref]
        assert len(beams[i]) == 64 and best[i] == beams[i][0]
        # the GPU probabilities differ from the oracle's by ~1e-6, so scores are compared at 1e-3 here
        # (the 1e-4 bound on identical inputs is tests/test_gpu_beam.py) and the top beams must coincide
        assert beams[i][:5] == ref_strings[:5], (beams[i][:5], ref_strings[:5])
        for k in range(5):
            assert abs(float(scores[pos, k]) - ref[k][0]) < 1e-3 * max(1.0, abs(ref[k][0]) / 100)


def test_lstm_three_conv_model_through_surface():
    from danspeech_amd import Recognizer
    model, sd, cfg = _model("lstm3", 48, 2, seed=13, kind="lstm", conv=3)
    rec = Recognizer(model=model)
    clips = [syn.make_clip(7, 20000)]
    assert rec.recognize_batch(clips) == _oracle_greedy(sd, cfg, clips)


def test_decoders_standalone_api(golden):
    """GreedyDecoder / BeamCTCDecoder keep the reference's decode(probs, sizes) -> (strings, offsets) contract."""
    from danspeech_amd.deepspeech.decoder import GreedyDecoder, BeamCTCDecoder
    g = golden("g5_greedy")
    labels = syn.DANSPEECH_LABELS
    dec = GreedyDecoder(labels, blank_index=labels.index("_"))
    strings, offsets = dec.decode(torch.from_numpy(g["probs"]), torch.from_numpy(g["sizes"]))
    assert [s[0] for s in strings] == [str(s) for s in g["strings"]]
    assert all(isinstance(o[0], torch.Tensor) and o[0].dtype == torch.int32 for o in offsets)
    b = BeamCTCDecoder(labels, beam_width=8, blank_index=0)
    strings, offsets = b.decode(torch.from_numpy(g["probs"]), torch.from_numpy(g["sizes"]))
    assert len(strings) == g["probs"].shape[0] and all(len(s) == 8 for s in strings)
    assert strings[0][0] == str(g["strings"][0])      # peaky probabilities: best beam = greedy path


# ---- the audio front door on the device ---------------------------------------

def _wav_bytes(samples, width):
    """Little-endian frames of the given sample width from int64 sample values (interleaved if 2-D)."""
    v = np.asarray(samples, dtype=np.int64).reshape(-1)
    if width == 1:
        return (v + 128).astype(np.uint8).tobytes()
    if width == 2:
        return v.astype("<i2").tobytes()
    if width == 4:
        return v.astype("<i4").tobytes()
    u = (v & 0xFFFFFF).astype(np.uint32)
    return np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], axis=1).astype(np.uint8).tobytes()


@pytest.mark.parametrize("width,channels", [(1, 1), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2)])
def test_raw_wav_frames_decoded_on_device_equal_host_load_audio(width, channels):
    """dsmi_features on a file's raw frames == dsmi_features on load_audio's float64 output, bit for bit
    (same STFT kernel; only the sample decode and the saturating L+R fold move to the GPU)."""
    from danspeech_amd import _native
    from danspeech_amd.audio.resources import _frames_to_int
    rng = np.random.default_rng(50 + 10 * width + channels)
    lim = 1 << (8 * width - 1)
    raws, refs, n = [], [], []
    for frames in (4000, 1777):
        v = rng.integers(-lim, lim, size=(frames, channels))
        v[:50] = lim - 1          # L+R overflows: the fold must saturate, both ways
        v[50:100] = -lim
        raw = _wav_bytes(v, width)
        d = _frames_to_int(raw, width)
        if channels == 2:
            d = d.reshape(-1, 2)
            d = np.clip(d[:, 0] + d[:, 1], -lim, lim - 1)
        raws.append(raw); refs.append(d.astype(np.float64)); n.append(frames)
    fe = _native.NativeFrontend()
    n = np.array(n, dtype=np.int64)
    a, fa = fe.features(torch.from_numpy(np.frombuffer(b"".join(raws), dtype=np.uint8).copy()).cuda(), n, wav_format=(width, channels))
    b, fb = fe.features(torch.from_numpy(np.concatenate(refs)).cuda(), n)
    assert np.array_equal(fa, fb)
    assert torch.equal(a, b)
    fe.close()


def test_recognize_files_equals_recognize_of_load_audio():
    """Config 1 through the device front door: the stereo example file's frames go to the GPU undecoded."""
    from danspeech_amd import Recognizer
    from danspeech_amd.audio import load_audio
    m, sd, cfg = _model("frontdoor", 64, 2, seed=71)
    rec = Recognizer(model=m)
    want = rec.recognize(load_audio(WAV))
    got = rec.recognize_files([WAV, WAV])
    assert got == [want, want] and isinstance(want, str)
    with pytest.raises(_feature_error()):
        from danspeech_amd import _native
        fe = _native.NativeFrontend()
        fe.features(torch.zeros(8, dtype=torch.uint8).cuda(), np.array([4], dtype=np.int64), wav_format=(1, 2))


def _feature_error():
    from danspeech_amd import _native
    return (ValueError, _native.DsmiError)


# ---- offline long-form segmentation --------------------------------------------

def _long_recording(seconds=50, seed=90):
    """Noise floor well under the gate with speech-like bursts of various lengths, some shorter than
    the phrase threshold, some separated by pauses shorter than the pause threshold, one at t=0 and one
    running into the end of the recording (dropped, as in the script)."""
    rng = np.random.default_rng(seed)
    n = seconds * 16000
    x = rng.normal(0, 60, n)
    t = 0
    spans = [(0, 9000)]
    t = 30000
    while t < n - 40000:
        dur = int(rng.choice([1500, 4000, 12000, 30000, 70000]))
        spans.append((t, t + dur))
        t += dur + int(rng.choice([3000, 7000, 12000, 25000]))
    spans.append((n - 9000, n))
    for a, b in spans:
        b = min(b, n)
        x[a:b] += rng.normal(0, 2500, b - a)
    return np.round(x)


@pytest.mark.parametrize("step", [1024, 512, 2048])
def test_segmentation_equals_the_script_restatement(step):
    from danspeech_amd import _native
    from oracle import segmentation as oseg
    audio = _long_recording()
    want, e_want = oseg.segment(audio, energy_threshold=600, step=step)
    fe = _native.NativeFrontend()
    hop_s = step / 16000.0
    kw = dict(energy_threshold=600, step=step, pause_hops=int(np.ceil(0.55 / hop_s)), phrase_hops=int(np.ceil(0.2 / hop_s)))
    got, e_got = fe.segment(torch.from_numpy(audio).cuda(), return_energies=True, **kw)
    assert np.array_equal(e_got, e_want)                      # bit-exact: numpy's summation order is reproduced
    assert len(want) >= 3 and [tuple(int(v) for v in r) for r in got] == want
    # int16 samples on the device give the same phrases
    got16 = fe.segment(torch.from_numpy(audio.astype(np.int16)).cuda(), **kw)
    assert np.array_equal(got16, got)
    # degenerate inputs: shorter than one hop, and silence
    assert len(fe.segment(torch.zeros(step, dtype=torch.float64).cuda(), **kw)) == 0
    assert len(fe.segment(torch.zeros(40 * step, dtype=torch.float64).cuda(), **kw)) == 0
    with pytest.raises(_native.DsmiError):
        fe.segment(torch.from_numpy(audio).cuda(), step=1000)
    fe.close()


def test_segmentation_equals_the_reference_script_golden():
    """G9 (tests/golden/g9_segments.json): the sample ranges the reference's own example script handed to recognize() on
    seeded signals (tools/gen_golden_segments.py ran the script itself) -- dsmi_segment on the int16 samples must give
    exactly those, including the overlapping lead-in, phrases at the very start and the offset start."""
    import hashlib
    import json
    from danspeech_amd import _native
    fe = _native.NativeFrontend()
    cases = json.load(open(os.path.join(HERE, "golden", "g9_segments.json"), encoding="utf-8"))["cases"]
    total = 0
    for c in cases:
        pcm = syn.gated_signal(c["plan"], c["seed"])
        assert hashlib.sha256(pcm.tobytes()).hexdigest() == c["sha256"]
        pcm = pcm[c["offset_seconds"] * 16000:]
        got = fe.segment(torch.from_numpy(pcm.copy()).cuda())                  # the script's defaults: 600, 1024, 0.55 s, 0.2 s
        assert [[int(a), int(b)] for a, b in got] == c["segments"], c["name"]
        got64 = fe.segment(torch.from_numpy(pcm.astype(np.float64)).cuda())     # as recognize_long passes load_audio's array
        assert np.array_equal(got64, got)
        total += len(got)
    assert total >= 12
    fe.close()


def test_recognize_long_equals_per_phrase_recognize():
    from danspeech_amd import Recognizer
    from oracle import segmentation as oseg
    audio = _long_recording(seconds=30, seed=91)
    m, sd, cfg = _model("longform", 64, 2, seed=72)
    rec = Recognizer(model=m)
    got = rec.recognize_long(audio, max_batch=3)
    want_segs, _ = oseg.segment(audio)
    assert [(a, b) for a, b, _ in got] == want_segs and len(got) >= 3
    for a, b, text in got:
        assert text == rec.recognize(audio[a:b])


def test_recognize_files_groups_by_format_and_segment_takes_raw_wav_frames(tmp_path):
    """Files of different sample formats in one recognize_files call (one batch per format), and dsmi_segment on
    a stereo file's raw frames (same phrases as on load_audio's fold of it)."""
    import wave
    from danspeech_amd import Recognizer, _native
    from danspeech_amd.audio import load_audio
    from danspeech_amd.audio.resources import read_wav_frames
    mono = (_long_recording(seconds=6, seed=92)).astype(np.int16)
    p_mono = str(tmp_path / "mono16.wav")
    with wave.open(p_mono, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(mono.astype("<i2").tobytes())
    m, sd, cfg = _model("mixed-formats", 64, 2, seed=73)
    rec = Recognizer(model=m)
    got = rec.recognize_files([WAV, p_mono, WAV])
    assert got == [rec.recognize(load_audio(p)) for p in (WAV, p_mono, WAV)]
    # segmentation straight on the stereo example file's frames
    raw, width, nch = read_wav_frames(WAV)
    fe = _native.NativeFrontend()
    a = fe.segment(torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda(), energy_threshold=300, wav_format=(width, nch))
    b = fe.segment(torch.from_numpy(load_audio(WAV)).cuda(), energy_threshold=300)
    assert np.array_equal(a, b)
    fe.close()


# ---- the reference's own lines around librosa / ctcdecode, pinned by tools/gen_golden_surface.py (G10 - G12) --------------
def test_g10_parsers_on_the_gpu_equal_the_reference_post_stft(golden):
    """dsmi_features / dsmi_features_stream against what the REFERENCE's parsers (parsers.py:50-72,102-164) make of the
    documented STFT: log1p, float32, unbiased std; hop carry-over and drifting statistics.  (The STFT itself: unpinned.)"""
    import json
    from danspeech_amd.audio import load_audio
    from danspeech_amd.audio.parsers import SpectrogramAudioParser, InferenceSpectrogramAudioParser
    g = golden("g10_parsers")
    clips = {"wav": load_audio(WAV), "c0": syn.make_clip(0, 16000), "c1": syn.make_clip(1, 4321), "short": syn.make_clip(2, 700)}
    for name, y in clips.items():
        for normalize in (True, False):
            want = g["spect_%s_%d" % (name, normalize)]
            got = SpectrogramAudioParser({"normalize": normalize}).parse_audio(y).cpu().numpy()
            assert got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-5)
    p = InferenceSpectrogramAudioParser()
    for ui, (name, parts) in enumerate(json.loads(str(g["stream_plan"]))):
        y, pos = clips[name], 0
        for k, n in enumerate(parts):
            s = p.parse_audio(y[pos:pos + n], is_last=(k == len(parts) - 1))
            pos += n
            want = g["stream_u%d_p%d" % (ui, k)]
            if want.size == 0:
                assert len(s) == 0
            else:
                np.testing.assert_allclose(s.cpu().numpy(), want, rtol=0, atol=3e-5)
            np.testing.assert_allclose([p.input_mean, p.input_std, p.alpha], g["stream_u%d_p%d_state" % (ui, k)], rtol=2e-6, atol=1e-6)
        p.reset()


def test_g11_recognize_end_to_end_equals_the_reference_run():
    """SURVEY 8(c) G6 part 2: Recognizer.recognize on the GPU against the REFERENCE's own Recognizer.recognize (torch CPU model,
    seeded weights, the documented STFT), for the greedy steps of the recorded scenario; probabilities against the
    reference model's."""
    import json
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.model import DeepSpeech
    gold = json.load(open(os.path.join(HERE, "golden", "g11_surface.json"), encoding="utf-8"))
    z = np.load(os.path.join(HERE, "golden", "g11_surface.npz"))
    cfg = gold["cfg"]
    sd = syn.make_state_dict(2, "gru", cfg["rnn_hidden_size"], cfg["rnn_layers"], seed=gold["seed"], **syn.TALKATIVE)

    def model(name, labels):
        return DeepSpeech(name, labels=labels, rnn_hidden_size=cfg["rnn_hidden_size"], rnn_layers=cfg["rnn_layers"]).load_state_dict(sd)

    models = {"m1": model("golden-m1", gold["labels"]), "m2": model("golden-m2", gold["other_labels"])}
    clips = [syn.make_clip(*c) for c in gold["clip_ids"]]
    rec, checked = None, 0
    for ev in gold["events"]:
        op = ev["op"]
        if op[0] == "new":
            kw = dict(op[1])
            name = kw.pop("model")
            kw.pop("lm", None)                     # "/other/lm.klm" does not exist: the beam steps belong to test_gpu_beam.py
            rec = Recognizer(model=models[name], **kw)
        elif op[0] == "update_model":
            rec.update_model(models[op[1]])
        elif op[0] == "update_decoder" and op[1].get("lm") in (None, "greedy"):
            rec.update_decoder(**op[1])
        elif op[0] == "recognize" and ev["state"]["decoder"] == "GreedyDecoder" and rec.danspeech_recognizer.lm == "greedy":
            got = rec.recognize(clips[op[1]], show_all=op[2])
            assert got == ev["result"], op
            checked += 1
    assert checked >= 3
    for k, y in enumerate(clips):
        eng = Recognizer(model=models["m1"]).danspeech_recognizer
        feats, frames = eng.audio_parser.parse_batch([y])
        probs, sizes = eng.model(feats, torch.from_numpy(frames.astype(np.int32)))
        assert sizes.tolist() == z["sizes%d" % k].tolist()
        np.testing.assert_allclose(probs.cpu().numpy(), z["probs%d" % k], rtol=0, atol=1e-4)


def test_g12_reference_accepted_package_runs_on_the_gpu(golden):
    from danspeech_amd.deepspeech.model import DeepSpeech
    g = golden("g12_package_forward")
    m = DeepSpeech.load_model(os.path.join(HERE, "golden", "g12_package.pth")).to("cuda")
    probs, out_lens = m(torch.from_numpy(g["x"]), torch.from_numpy(g["out_lens"] * 0 + g["x"].shape[-1]))
    assert out_lens.tolist() == g["out_lens"].tolist()
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=0, atol=1e-4)
