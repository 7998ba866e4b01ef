"""GPU tests of the failure and guard paths around the persistent recurrent kernels:

* a hand-off wait that times out (forced with the DSMI_DEBUG_DROP_SIGNAL / DSMI_DEBUG_SPIN_LIMIT test
  hooks: one workgroup never signals one step) must be reported by the SAME forward -- the batch is
  recomputed on the per-step path inside ``dsmi_forward_status`` / ``dsmi_rnn_layer`` and the results
  equal the oracle's;
* weights outside the fp16 range of the split operands (|w| >= 60000) route the affected stage to the
  fp32-MFMA kernels at load time; tiny weights (fp16-subnormal ``hi`` terms) and saturating gates
  (the hardware exp / rcp cell) stay within the parity bound.
"""
import os
import warnings

import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def native():
    from danspeech_amd import _native
    assert torch.cuda.is_available()
    _native.lib()
    return _native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cfg(H, L, kind="gru", cl=2):
    return dict(conv_layers=cl, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, bidirectional=True, context=20)


def _batch(B=5, T=161, seed=3):
    lens = np.sort(np.random.default_rng(seed).integers(T // 2, T + 1, size=B))[::-1].astype(np.int32)
    lens[0] = T
    x = syn.make_features(B, T, seed=seed)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    return x, lens


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("H", [64, 40])          # 64: rnn_persist16 (H % 16 == 0); 40: first-generation kernel
def test_timeout_is_reported_and_recomputed_by_the_same_forward(native, H):
    from oracle import torch_port as tp
    cfg = _cfg(H, 3)
    sd = syn.make_state_dict(2, "gru", H, 3, seed=21, **syn.TALKATIVE)
    x, lens = _batch()
    ref, ol_ref = tp.forward(sd, cfg, x, lens)
    # workgroup 1 of direction 0 never signals step 7 of layer 1; a wait gives up after 3000 polls
    with _env(DSMI_DEBUG_DROP_SIGNAL="1:1:7", DSMI_DEBUG_SPIN_LIMIT="3000"):
        m = native.NativeModel(cfg, sd)
    xd = _dev(x)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p, ol = m.forward(xd, lens, check=False)
        assert m.status() is True                       # this very forward reports it ...
    assert any("timed out" in str(x.message) for x in w)
    assert m.recompute_count() == 1
    assert np.array_equal(ol, ol_ref)
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=0, atol=1e-4)     # ... and its results are the recomputed ones
    # the handle stays usable (per-step path from now on), nothing left to report
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p2, _ = m.forward(xd, lens)
    assert not w and m.recompute_count() == 1
    np.testing.assert_allclose(p2.cpu().numpy(), ref, rtol=0, atol=1e-4)
    m.close()


def test_uncollected_timeout_fails_the_next_forward(native):
    cfg = _cfg(64, 2)
    sd = syn.make_state_dict(2, "gru", 64, 2, seed=22)
    x, lens = _batch()
    with _env(DSMI_DEBUG_DROP_SIGNAL="0:0:3", DSMI_DEBUG_SPIN_LIMIT="3000"):
        m = native.NativeModel(cfg, sd)
    xd = _dev(x)
    m.forward(xd, lens, check=False)
    torch.cuda.synchronize()
    with pytest.raises(native.DsmiError) as e:         # status never collected: loud failure, not silent garbage
        m.forward(xd, lens, check=False)
    assert e.value.code == native.DSMI_ERR_TIMEOUT
    p, _ = m.forward(xd, lens)                          # recovered: per-step path
    from oracle import torch_port as tp
    np.testing.assert_allclose(p.cpu().numpy(), tp.forward(sd, cfg, x, lens)[0], rtol=0, atol=1e-4)
    m.close()


def test_rnn_layer_entry_recomputes_after_timeout(native):
    from oracle import model as om
    H = 64
    cfg = _cfg(H, 2, cl=1)
    audio_conf = dict(sampling_rate=100, window_size=0.02)       # n_freq = 2 -> layer-0 input size 32
    sd = syn.make_state_dict(1, "gru", H, 2, seed=23, sample_rate=100)
    with _env(DSMI_DEBUG_DROP_SIGNAL="1:2:4", DSMI_DEBUG_SPIN_LIMIT="3000"):
        m = native.NativeModel(cfg, sd, audio_conf=audio_conf)
    rng = np.random.default_rng(5)
    lens = np.array([30, 22, 9], dtype=np.int32)
    x = rng.standard_normal((30, 3, H)).astype(np.float32)
    y = m.rnn_layer(1, _dev(x), lens).cpu().numpy()
    ref = om.batch_rnn(sd, 1, "gru", x, lens, True, True)
    np.testing.assert_allclose(y, ref, rtol=0, atol=5e-6)
    assert m.recompute_count() == 1
    m.close()


@pytest.mark.parametrize("where", ["conv", "w_ih", "w_hh", "bn"])
def test_fp16_range_guard_falls_back_to_fp32_kernels(native, where):
    """One weight beyond fp16's range: the stage must leave the split-fp16 kernels (a split would produce inf)."""
    from oracle import torch_port as tp
    H = 64
    cfg = _cfg(H, 2)
    sd = syn.make_state_dict(2, "gru", H, 2, seed=24)
    if where == "conv":
        sd["conv.seq_module.3.weight"][3, 5, 7, 2] = 7.0e4
    elif where == "w_ih":
        sd["rnns.1.rnn.weight_ih_l0"][10, 3] = -7.0e4
    elif where == "w_hh":
        sd["rnns.0.rnn.weight_hh_l0_reverse"][100, 17] = 6.6e4
    else:
        sd["rnns.1.batch_norm.module.weight"][5] = 5.0e4        # 2|a| + |b| bound of the GEMM's A operand
    x, lens = _batch(B=4, T=121, seed=8)
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    ref, _ = tp.forward(sd, cfg, x, lens)
    pn = p.cpu().numpy()
    assert np.isfinite(pn).all()
    np.testing.assert_allclose(pn, ref, rtol=0, atol=1e-4)
    m.close()


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
def test_tiny_weights_and_saturating_gates(native, kind):
    """|w| ~ 1e-7 (the fp16 ``hi`` term is subnormal or zero, the value lives in ``lo``) in one layer, and biases of
    +-100 in another (exp overflows to inf / underflows to 0 in the hardware-exp cell: sigmoid and tanh must
    saturate to exactly 0 / 1 / -1, never NaN)."""
    from oracle import torch_port as tp
    from oracle import model as om
    H = 64
    cfg = _cfg(H, 3, kind=kind)
    sd = syn.make_state_dict(2, kind, H, 3, seed=25, **syn.TALKATIVE)
    for sfx in ("", "_reverse"):
        sd["rnns.1.rnn.weight_hh_l0" + sfx] = (sd["rnns.1.rnn.weight_hh_l0" + sfx] * np.float32(1e-6)).astype(np.float32)
        sd["rnns.1.rnn.weight_ih_l0" + sfx] = (sd["rnns.1.rnn.weight_ih_l0" + sfx] * np.float32(1e-6)).astype(np.float32)
        b = sd["rnns.2.rnn.bias_hh_l0" + sfx]
        b[0::7] = 100.0
        b[3::7] = -100.0
    x, lens = _batch(B=3, T=141, seed=9)
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    pn = p.cpu().numpy()
    assert np.isfinite(pn).all()
    ref, _ = tp.forward(sd, cfg, x, lens)
    np.testing.assert_allclose(pn, ref, rtol=0, atol=1e-4)
    # the layer with the tiny weights on its own, at stage tolerance
    audio_conf = dict(sampling_rate=100, window_size=0.02)
    sd1 = syn.make_state_dict(1, kind, H, 2, seed=26, sample_rate=100)
    for sfx in ("", "_reverse"):
        sd1["rnns.1.rnn.weight_hh_l0" + sfx] = (sd1["rnns.1.rnn.weight_hh_l0" + sfx] * np.float32(3e-6)).astype(np.float32)
    m1 = native.NativeModel(_cfg(H, 2, kind=kind, cl=1), sd1, audio_conf=audio_conf)
    rng = np.random.default_rng(6)
    l1 = np.array([40, 33, 12], dtype=np.int32)
    x1 = rng.standard_normal((40, 3, H)).astype(np.float32)
    y = m1.rnn_layer(1, _dev(x1), l1).cpu().numpy()
    np.testing.assert_allclose(y, om.batch_rnn(sd1, 1, kind, x1, l1, True, True), rtol=0, atol=5e-6)
    m.close(); m1.close()


@pytest.mark.parametrize("variant", ["ring", "paired", "half", "whole"])
def test_two_batches_in_flight_on_two_handles(native, variant):
    """Two handles, two streams, forwards enqueued back to back without waiting.  The kernel variant follows from what the
    caller says (set_inflight) and the batch: two batches in flight of 17+ clips -> the paired-tile kernel, each batch on its
    own half of the chip; of at most 16 clips -> half-CU workgroups, the two batches' kernels co-resident on the same CUs;
    one batch in flight -> whole-CU workgroups, which the gate chains.  Since round 4 batches in flight run the ring kernel, each
    on its own gate slot ("ring"); DSMI_RNN_KERNEL=duo keeps the older kernels reachable ("paired", "half").  Either way both
    batches must equal the oracle, repeatedly."""
    from oracle import torch_port as tp
    cfg = _cfg(800, 2)
    sd = syn.make_state_dict(2, "gru", 800, 2, seed=31, **syn.TALKATIVE)
    with _env(**({} if variant in ("ring", "whole") else dict(DSMI_RNN_KERNEL="duo"))):
        models = [native.NativeModel(cfg, sd) for _ in range(2)]
    for m in models:
        m.set_inflight(1 if variant == "whole" else 2)
    streams = [torch.cuda.Stream() for _ in range(2)]
    batches = [_batch(B=16, T=301, seed=40), _batch(B=11, T=257, seed=41)] if variant == "half" else \
        [_batch(B=32, T=301, seed=40), _batch(B=20, T=257, seed=41)]
    refs = [tp.forward(sd, cfg, x, lens)[0] for x, lens in batches]
    xs = [_dev(x) for x, _ in batches]
    torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs.append(models[k].forward(xs[k], batches[k][1], check=False))
        for k in range(2):
            assert models[k].status() is False
            p, ol = outs[k]
            pn = p.cpu().numpy()
            for b in range(pn.shape[0]):
                np.testing.assert_allclose(pn[b, :ol[b]], refs[k][b, :ol[b]], rtol=0, atol=1e-4)
    for m in models:
        assert m.recompute_count() == 0
        m.close()


@pytest.mark.parametrize("kind,H,B", [("gru", 800, 32), ("gru", 64, 17), ("lstm", 512, 48), ("rnn", 96, 64), ("gru", 896, 40), ("lstm", 64, 32),
                                      ("gru", 800, 96), ("gru", 800, 72)])
def test_paired_tile_kernel_equals_oracle_and_the_single_tile_kernels(native, kind, H, B):
    """rnn_persist_duo (the default for 17+ clips when the shape fits): all cell types, one pair / two pairs of tiles, an odd
    tile count (the last half B idle), a partial last tile, the seven-k-block shape, ragged lengths -- against the oracle
    and against the same batch with one batch in flight (up to 32 clips: rnn_persist16's whole-CU workgroups; more: the
    same kernel).  H = 800 with 96 / 72 clips: three tile pairs of 100 workgroups each, i.e. two launches per layer (windows
    of two pairs and one; with 72 clips the last pair's second half idle)."""
    from oracle import torch_port as tp
    cfg = _cfg(H, 2, kind=kind)
    sd = syn.make_state_dict(2, kind, H, 2, seed=61, **syn.TALKATIVE)
    x, lens = _batch(B=B, T=181, seed=62)
    ref, ol_ref = tp.forward(sd, cfg, x, lens)
    outs = []
    for inflight in (2, 1):
        with _env(DSMI_RNN_KERNEL="duo"):          # (batches in flight run the ring kernel by default: tests/test_gpu_ring.py)
            m = native.NativeModel(cfg, sd)
        m.set_inflight(inflight)
        p, ol = m.forward(_dev(x), lens)
        assert np.array_equal(ol, ol_ref) and m.recompute_count() == 0
        outs.append(p.cpu().numpy())
        m.close()
    for b in range(B):
        np.testing.assert_allclose(outs[0][b, :ol_ref[b]], ref[b, :ol_ref[b]], rtol=0, atol=1e-4)
        np.testing.assert_allclose(outs[0][b, :ol_ref[b]], outs[1][b, :ol_ref[b]], rtol=0, atol=5e-5)


@pytest.mark.parametrize("H,B,drop", [(1200, 64, "1:5:9"), (1024, 104, "0:60:17")])
def test_tile_walking_kernel_timeout_is_recomputed(native, H, B, drop):
    """The tile-walking kernel (more 16-clip tiles than fit side by side: rnn_persist16_pipe_kernel) with its wave roles: a
    feeder wave's poll never answers (one workgroup does not signal one step of tile 0), the feeders give up, the cell waves
    run on behind the barrier, the forward reports it and its results are the recomputed ones.  H = 1200: four tiles per
    workgroup, 150 workgroups; H = 1024 with 104 clips: two groups of workgroups walking four and three tiles (the second
    in the one-role form) in one launch, sharing the error word."""
    from oracle import torch_port as tp
    cfg = _cfg(H, 2)
    sd = syn.make_state_dict(2, "gru", H, 2, seed=65, **syn.TALKATIVE)
    x, lens = _batch(B=B, T=81, seed=66)
    ref, _ = tp.forward(sd, cfg, x, lens)
    with _env(DSMI_DEBUG_DROP_SIGNAL=drop, DSMI_DEBUG_SPIN_LIMIT="3000"):
        m = native.NativeModel(cfg, sd)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p, ol = m.forward(_dev(x), lens, check=False)
        assert m.status() is True
    assert m.recompute_count() == 1 and any("timed out" in str(x.message) for x in w)
    pn = p.cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(pn[b, :ol[b]], ref[b, :ol[b]], rtol=0, atol=1e-4)
    # the same handle, nothing dropped any more?  (the hook is read when the handle is made: a fresh one runs clean)
    m.close()
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    assert m.recompute_count() == 0
    m.close()


@pytest.mark.parametrize("H,B,drop", [(64, 24, "1:2:9"), (800, 96, "1:3:11")])
def test_paired_tile_kernel_timeout_is_recomputed(native, H, B, drop):
    """The paired-tile kernel's own hand-off timeout (DSMI_RNN_KERNEL=duo, two batches in flight): one pair on the handle's lane,
    and H = 800 with 96 clips = three tile pairs in two windows, several launches sharing one error word and one counter array."""
    from oracle import torch_port as tp
    cfg = _cfg(H, 2)
    sd = syn.make_state_dict(2, "gru", H, 2, seed=63, **syn.TALKATIVE)
    x, lens = _batch(B=B, T=161, seed=64)
    ref, _ = tp.forward(sd, cfg, x, lens)
    with _env(DSMI_DEBUG_DROP_SIGNAL=drop, DSMI_DEBUG_SPIN_LIMIT="3000", DSMI_RNN_KERNEL="duo"):
        m = native.NativeModel(cfg, sd)
    m.set_inflight(2)
    m.set_profiling(2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p, ol = m.forward(_dev(x), lens, check=False)
        assert m.status() is True
    assert m.recompute_count() == 1
    pn = p.cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(pn[b, :ol[b]], ref[b, :ol[b]], rtol=0, atol=1e-4)
    m.close()


def _lm_recognizer(tmp_path, H=64, L=2, seed=31):
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.model import DeepSpeech
    lm = str(tmp_path / "lm3.arpa")
    syn.make_arpa(lm, order=3, n_words=300, seed=9, ngrams_per_order=800)
    sd = syn.make_state_dict(2, "gru", H, L, seed=seed, **syn.TALKATIVE)
    model = DeepSpeech("m", rnn_hidden_size=H, rnn_layers=L).load_state_dict(sd)
    return Recognizer(model=model, lm=lm, beam_width=16)


def test_pipelined_beam_search_survives_a_recomputed_batch(native, tmp_path):
    """recognize_batches with a language model keeps a search in flight per batch on alternating decoder handles.  A batch
    whose persistent kernel timed out is recomputed by its collect and must then be decoded again on ITS OWN handle: handle 0
    may hold the next batch's search by then."""
    rec = _lm_recognizer(tmp_path)
    batches = [[syn.make_clip(10 * b + i, 16000 + 800 * i) for i in range(3)] for b in range(4)]
    want = [rec.recognize_batch(b, show_all=True) for b in batches]
    with _env(DSMI_DEBUG_DROP_SIGNAL="1:1:5", DSMI_DEBUG_SPIN_LIMIT="3000"):
        rec2 = _lm_recognizer(tmp_path)
        eng = rec2.danspeech_recognizer
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = list(rec2.recognize_batches(batches, show_all=True))
    assert got == want
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    assert sum(h.recompute_count() for h in handles) >= 1 and any("timed out" in str(x.message) for x in w)
    assert rec2.recognize_batch(batches[0], show_all=True) == want[0]


def test_leaving_the_batch_pipeline_early_leaves_the_engine_usable(native, tmp_path):
    """A caller that stops consuming recognize_batches (break, or an exception in its loop) must not leave a forward or a
    beam-search ticket uncollected: the next call on the same engine works."""
    rec = _lm_recognizer(tmp_path, seed=32)
    batches = [[syn.make_clip(50 + 10 * b + i, 12000 + 500 * i) for i in range(2)] for b in range(5)]
    want = [rec.recognize_batch(b) for b in batches]
    for k, res in enumerate(rec.recognize_batches(batches)):
        assert res == want[k]
        if k == 1:
            break
    assert rec.recognize_batch(batches[4]) == want[4]
    assert list(rec.recognize_batches(batches[:3])) == want[:3]
    with pytest.raises(ZeroDivisionError):
        for res in rec.recognize_batches(batches):
            1 / 0
    assert rec.recognize(batches[2][0]) == want[2][0]
