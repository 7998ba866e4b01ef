"""GPU: the fused C-ABI entries a host without the Python layer uses -- dsmi_session_* / dsmi_recognize_* (recognize.hip)
and dsmi_comm_* (comm.hip, RCCL bound at run time) -- against the Python surface on the same handles' weights, which the
other GPU tests hold to the oracle and the reference's goldens."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _engine(H=64, L=3, seed=12, lm_path=None, kind="gru"):
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(2, kind, H, L, seed=seed, **syn.TALKATIVE)
    model = DeepSpeech("m", rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, conv_layers=2).load_state_dict(sd)
    if lm_path:
        from danspeech_amd.language_models import CustomLanguageModel
        rec = Recognizer(model=model, lm=CustomLanguageModel(lm_path))
    else:
        rec = Recognizer(model=model)
    return rec, sd


def _session(rec, sd, kind="gru", H=64, L=3):
    """A SECOND set of native handles on the same weights (what a non-Python host would create), bound into a session."""
    from danspeech_amd import _native
    eng = rec.danspeech_recognizer
    cfg = eng.model._cfg()
    model = _native.NativeModel(cfg, sd, device=0)
    conf = eng.audio_config
    fe = _native.NativeFrontend(dict(sampling_rate=conf["sampling_rate"], window_size=conf["window_size"], window_stride=conf["window_stride"],
                                     window=conf["window"], normalize=conf.get("normalize", True)), device=0)
    dec = _native.NativeDecoder(eng.labels, blank_index=eng.labels.index("_"), device=0)
    return _native.NativeSession(fe, model, dec), (fe, model, dec)


def _clips(lengths, dtype=np.int16, seed0=0):
    out = []
    for i, n in enumerate(lengths):
        c = syn.make_clip(seed0 + i, n)
        if dtype == np.int16:
            c = np.clip(np.round(c), -32768, 32767).astype(np.int16)
        else:
            c = c.astype(dtype)
        out.append(c)
    return out


@pytest.mark.parametrize("dtype", [np.int16, np.float64])
def test_fused_greedy_equals_python_surface(dtype):
    rec, sd = _engine()
    ses, keep = _session(rec, sd)
    clips = _clips([16000, 40000, 8000, 40000, 23456, 31999, 12345], dtype)
    want = rec.recognize_batch(clips)
    got, nbytes, scores = ses.recognize_batch(clips)
    assert got == want and max(len(t) for t in want) >= 10
    assert list(nbytes) == [len(t.encode("utf-8")) for t in want] and not scores.any()
    # a short text buffer: cut at a label boundary, full length reported
    short, nb2, _ = ses.recognize_batch(clips, text_stride=8)
    for s_, w, n in zip(short, want, nb2):
        assert n == len(w.encode("utf-8")) and w.encode("utf-8").startswith(s_.encode("utf-8")) and len(s_.encode("utf-8")) <= 7
    ses.close()


def test_two_sessions_in_flight_and_errors():
    from danspeech_amd import _native
    rec, sd = _engine(seed=13)
    a, keep_a = _session(rec, sd)
    b, keep_b = _session(rec, sd)
    for _, m, _ in (keep_a, keep_b):
        m.set_inflight(2)
    ca, cb = _clips([30000, 9000, 20000]), _clips([8000, 45000], seed0=10)
    a.enqueue(ca)
    b.enqueue(cb)
    with pytest.raises(_native.DsmiError) as e:
        a.enqueue(cb)                                   # one batch per session at a time
    assert e.value.code == _native.DSMI_ERR_INVALID and "not been collected" in e.value.msg
    ta, tb = a.collect()[0], b.collect()[0]
    assert ta == rec.recognize_batch(ca) and tb == rec.recognize_batch(cb)
    with pytest.raises(_native.DsmiError):
        a.collect()                                     # nothing enqueued
    with pytest.raises(_native.DsmiError) as e:
        a.enqueue([np.zeros(0, dtype=np.int16)])
    assert "empty clip" in e.value.msg
    a.close(); b.close()


def test_fused_beam_with_lm_equals_python_surface(tmp_path):
    lm_path = str(tmp_path / "syn3.arpa")
    syn.make_arpa(lm_path, order=3, n_words=300, seed=21, ngrams_per_order=800)
    rec, sd = _engine(seed=14, lm_path=lm_path)
    eng = rec.danspeech_recognizer
    ses, (fe, model, dec) = _session(rec, sd)
    dec.set_lm(lm_path, eng.alpha, eng.beta)
    clips = _clips([24000, 16000, 36000, 16000])
    want = rec.recognize_batch(clips, show_all=True)
    got, _, scores = ses.recognize_batch(clips, beam_width=eng.beam_width, cutoff_top_n=40, cutoff_prob=1.0)
    assert got == [beams[0] for beams in want]
    assert np.isfinite(scores).all() and scores.any()
    ses.close()


def test_comm_world1_scatter_recognise_gather():
    """One rank: RCCL is opened and a communicator of size 1 created; scatter -> device-resident shard (longest first) ->
    dsmi_recognize_enqueue_device -> gather restores the caller's order.  (N > 1: tests/test_parallel_gloo.py covers the
    same plan and order restoration with two processes on CPU.)"""
    from danspeech_amd import _native
    rec, sd = _engine(seed=15)
    ses, keep = _session(rec, sd)
    comm = _native.NativeComm(_native.NativeComm.unique_id(), 0, 1, 0)
    clips = _clips([9000, 30000, 9000, 41000, 16000])
    dev, n, pos, code, total = comm.scatter(clips)
    assert total == 5 and code == 0 and list(pos) == [3, 1, 4, 0, 2] and list(n) == [41000, 30000, 16000, 9000, 9000]
    ses.enqueue_device(dev, n, code)
    text, nbytes, _ = ses.collect(raw=True)
    out = comm.gather_text(text, pos, total)
    assert out == rec.recognize_batch(clips)
    with pytest.raises(_native.DsmiError) as e:
        ses.enqueue_device(dev, n[::-1].copy(), code)
    assert e.value.code == _native.DSMI_ERR_UNSORTED
    comm.close(); ses.close()


def test_c_host_example_equals_python_surface(tmp_path):
    """examples/host_recognize.c, built with gcc against include/dsmi.h + libdsmi.so only, run as its own process on the
    reference's example recording (stereo 16-bit WAV, raw frames handed over) and on a second copy of it in one batch."""
    import os
    import shutil
    import subprocess
    import sys
    from danspeech_amd import _native
    from danspeech_amd.audio import load_audio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import export_weights
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    wav = os.path.join(root, "tests", "golden", "u0013002.wav")
    # only one PROCESS per GPU runs persistent kernels (the child would get the per-step path, this process the persistent
    # one): put both on the per-step path so that the two transcripts come from the same arithmetic
    os.environ["DSMI_RNN_MODE"] = "steps"
    try:
        rec, sd = _engine(H=96, L=2, seed=16)
        want = rec.recognize(load_audio(wav))
    finally:
        del os.environ["DSMI_RNN_MODE"]
    eng = rec.danspeech_recognizer
    pack = str(tmp_path / "m.dsmiw")
    export_weights.write_pack(pack, sd, eng.model._cfg(), eng.labels, eng.audio_config)
    exe = str(tmp_path / "host_recognize")
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "host_recognize.c"),
                    "-o", exe, "-L", libdir, "-ldsmi", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe, pack, wav, wav], capture_output=True, text=True, timeout=600, env=dict(os.environ, DSMI_RNN_MODE="steps"))
    assert r.returncode == 0, r.stderr
    assert len(want) >= 10 and r.stdout.splitlines() == [want, want]


@pytest.mark.parametrize("world", [2, 4])
def test_comm_two_ranks_over_mock_transport(tmp_path, world):
    """The N > 1 loops of comm.hip (plan, slice offsets, grouped sends / receives, gather rows, fewer clips than ranks) with
    two and FOUR rank processes on this box's single GPU.  RCCL refuses two ranks on one device, so the transport is
    tests/mock_rccl.c -- same entry points, messages as files -- loaded through DSMI_RCCL_LIBRARY; everything above it
    (device buffers, staging, the session's device-resident path) is the product code."""
    import os
    import shutil
    import subprocess
    import sys
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    here = os.path.dirname(os.path.abspath(__file__))
    mock = str(tmp_path / "libmock_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(here, "mock_rccl.c"), "-o", mock,
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    work = tmp_path / "work"
    work.mkdir()
    # (one process per GPU gets the persistent kernels: both ranks on the per-step path, so that every transcript comes
    # from the same arithmetic whichever rank computed it)
    env = dict(os.environ, DSMI_RCCL_LIBRARY=mock, MOCK_RCCL_DIR=str(work), DSMI_RNN_MODE="steps")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "_comm_rank.py"), str(r), str(world), str(work)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert (work / "ok").exists()
