"""GPU tests of the ring kernel: one workgroup = 32 hidden units of one direction walking every 16-clip tile of its window, the
tiles' packed states staged through an LDS ring by LDS-DMA.  Two forms: four waves, one per SIMD on the whole register file
(rnn_persist_ring4.hip: windows of three tiles or more), eight waves (rnn_persist_ring.hip: windows of one or two tiles);
``DSMI_RNN_KERNEL=ring4`` / ``ring8`` runs one form on every window.  Reference semantics:
``BatchRNN.forward`` (danspeech/deepspeech/model.py:114-122) through the whole ``DeepSpeech.forward`` (:496-515); the
checker is oracle/torch_port.py (itself pinned to the reference's goldens, tests/test_oracle_torch_port.py)."""
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


# (kind, H, B, inflight): inflight 2 = one window on the handle's own gate slot; inflight 1 with more than 32 clips = the
# tiles spread over up to four windows side by side.  B 64 = two 32-clip batches in one launch (four real tiles); 17 / 32 =
# one or two real tiles (the eight-wave form by itself; padded with phantom tiles in the four-wave form); 40 / 72 = a partial
# last tile; 128 with inflight 2 = two launches of four tiles one after the other; H 800 / 896 = 13 / 14 k-blocks per wave of the
# four-wave form (the benchmarked shape and the widest), 7 of the eight-wave form; 48 units = an odd number of 16-unit groups
# (the last workgroup's second group idle); H 16 = one k-block (waves without any).
CASES = [("gru", 800, 64, 2), ("gru", 800, 32, 2), ("gru", 64, 17, 2), ("lstm", 512, 48, 2), ("rnn", 96, 64, 2),
         ("gru", 896, 40, 2), ("lstm", 64, 32, 2), ("gru", 800, 128, 2), ("gru", 800, 72, 1), ("gru", 800, 128, 1),
         ("gru", 48, 70, 2), ("gru", 16, 33, 2), ("rnn", 160, 100, 1)]


# form: "auto" = what the engine picks by the window's tiles; "ring4" / "ring8" = that form on every window (the four-wave form
# then also walks windows of one or two real tiles padded with phantom ones, the eight-wave form also windows of four)
@pytest.mark.parametrize("form", ["auto", "ring4", "ring8"])
@pytest.mark.parametrize("kind,H,B,inflight", CASES)
def test_ring_kernel_equals_oracle_and_the_older_kernels(native, kind, H, B, inflight, form):
    from oracle import torch_port as tp
    cfg = _cfg(H, 2, kind=kind)
    sd = syn.make_state_dict(2, kind, H, 2, seed=71, **syn.TALKATIVE)
    x, lens = _batch(B=B, T=181, seed=72)
    ref, ol_ref = tp.forward(sd, cfg, x, lens)
    with _env(**({} if form == "auto" else {"DSMI_RNN_KERNEL": form})):
        m = native.NativeModel(cfg, sd)
    m.set_inflight(inflight)
    m.set_profiling(2)
    p, ol = m.forward(_dev(x), lens)
    assert np.array_equal(ol, ol_ref) and m.recompute_count() == 0
    ring = p.cpu().numpy()
    m.close()
    with _env(DSMI_RNN_KERNEL="duo"):
        m = native.NativeModel(cfg, sd)
    m.set_inflight(inflight)
    p, ol = m.forward(_dev(x), lens)
    old = p.cpu().numpy()
    m.close()
    for b in range(B):
        np.testing.assert_allclose(ring[b, :ol_ref[b]], ref[b, :ol_ref[b]], rtol=0, atol=1e-4)
        np.testing.assert_allclose(ring[b, :ol_ref[b]], old[b, :ol_ref[b]], rtol=0, atol=5e-5)


# The reference's unidirectional models (model.py:399-407: ``bidirectional=False`` + ``Lookahead(context)``): one direction, grid.y = 1,
# chains d * ntiles + tile with d = 0 only, half as many CUs per window.
UNI_CASES = [("gru", 800, 64, 2), ("lstm", 256, 40, 2), ("gru", 320, 100, 1), ("rnn", 96, 33, 2)]


@pytest.mark.parametrize("form", ["auto", "ring4"])
@pytest.mark.parametrize("kind,H,B,inflight", UNI_CASES)
def test_ring_kernel_unidirectional_with_lookahead(native, kind, H, B, inflight, form):
    from oracle import torch_port as tp
    cfg = dict(conv_layers=2, rnn_type=kind, rnn_hidden_size=H, rnn_layers=2, bidirectional=False, context=20)
    sd = syn.make_state_dict(2, kind, H, 2, bidirectional=False, context=20, seed=91, **syn.TALKATIVE)
    x, lens = _batch(B=B, T=181, seed=92)
    ref, ol_ref = tp.forward(sd, cfg, x, lens)
    with _env(**({} if form == "auto" else {"DSMI_RNN_KERNEL": form})):
        m = native.NativeModel(cfg, sd)
    m.set_inflight(inflight)
    m.set_profiling(2)
    p, ol = m.forward(_dev(x), lens)
    assert np.array_equal(ol, ol_ref) and m.recompute_count() == 0
    assert m.kernel_stats()["rnn_layer_persistent"]["launches"] >= 2
    pn = p.cpu().numpy()
    m.close()
    for b in range(B):
        np.testing.assert_allclose(pn[b, :ol_ref[b]], ref[b, :ol_ref[b]], rtol=0, atol=1e-4)


def test_unidirectional_model_through_the_pipeline(native):
    """``transcribe_batches`` on a unidirectional model: merged forwards on several handles, equal to the lone calls."""
    from danspeech_amd import Recognizer
    from danspeech_amd.deepspeech.model import DeepSpeech
    sd = syn.make_state_dict(2, "gru", 320, 3, bidirectional=False, context=20, seed=93, **syn.TALKATIVE)
    model = DeepSpeech("uni", rnn_type="gru", rnn_hidden_size=320, rnn_layers=3, conv_layers=2, bidirectional=False, context=20).load_state_dict(sd)
    rec = Recognizer(model=model)
    eng = rec.danspeech_recognizer
    batches = [[syn.make_clip(10 * k + i, 24000 + 900 * ((i + k) % 9)) for i in range(20)] for k in range(6)]
    lone = [rec.recognize_batch(b) for b in batches]
    assert list(rec.recognize_batches(batches)) == lone
    handles = [eng.model._native] + [r[0]._native for r in eng._replicas]
    assert len(handles) == 4 and [h.recompute_count() for h in handles] == [0, 0, 0, 0]
    assert all(isinstance(t, str) for b in lone for t in b) and len({t for b in lone for t in b}) > 1


def test_two_models_of_different_widths_share_the_ring_slots(native):
    """Ring slots are counted per model (n_cus / CUs of its window) but index one set of events per device: H = 800 (50 CUs a
    window, five slots) beside H = 896 (56 CUs, four slots) must never be admitted beyond the device's 256 CUs -- windows that are
    not all resident spin to their timeout and are recomputed.  Nine handles in flight, none recomputed, results right."""
    from oracle import torch_port as tp
    specs = [(800, 5), (896, 4)]
    models, refs, xs, batches = [], [], [], []
    for H, count in specs:
        cfg = _cfg(H, 2)
        sd = syn.make_state_dict(2, "gru", H, 2, seed=95 + H, **syn.TALKATIVE)
        x, lens = _batch(B=64, T=161, seed=96 + H)
        ref = tp.forward(sd, cfg, x, lens)[0]
        for _ in range(count):
            m = native.NativeModel(cfg, sd)
            m.set_inflight(4)
            models.append(m); refs.append(ref); xs.append(_dev(x)); batches.append(lens)
    streams = [torch.cuda.Stream() for _ in models]
    torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        for k, m in enumerate(models):
            with torch.cuda.stream(streams[k]):
                outs.append(m.forward(xs[k], batches[k], check=False))
        for k, m in enumerate(models):
            assert m.status() is False
            p, ol = outs[k]
            pn = p.cpu().numpy()
            for b in range(0, pn.shape[0], 7):
                np.testing.assert_allclose(pn[b, :ol[b]], refs[k][b, :ol[b]], rtol=0, atol=1e-4)
    for m in models:
        assert m.recompute_count() == 0
        m.close()


def test_ring_kernel_is_what_runs(native):
    """Two batches in flight, cfgA's width: one launch per layer on H / 32 x 2 = 50 workgroups."""
    cfg = _cfg(800, 2)
    sd = syn.make_state_dict(2, "gru", 800, 2, seed=73)
    x, lens = _batch(B=64, T=101, seed=74)
    m = native.NativeModel(cfg, sd)
    m.set_inflight(2)
    m.set_profiling(2)
    m.forward(_dev(x), lens)
    ks = m.kernel_stats()["rnn_layer_persistent"]
    assert ks["launches"] == 2, ks
    m.close()


@pytest.mark.parametrize("form", ["auto", "ring8"])
def test_ring_kernel_timeout_is_recomputed(native, form):
    """A workgroup that never signals one step of chain 0: the poll of that step times out, the error word is raised, the
    forward's status reports it and the SAME batch is recomputed on the per-step path (api.hip collect_oldest).  Both forms
    (56 clips = four tiles: the four-wave form by itself)."""
    from oracle import torch_port as tp
    cfg = _cfg(64, 2)
    sd = syn.make_state_dict(2, "gru", 64, 2, seed=75, **syn.TALKATIVE)
    x, lens = _batch(B=56, T=161, seed=76)
    ref, _ = tp.forward(sd, cfg, x, lens)
    with _env(DSMI_DEBUG_DROP_SIGNAL="1:1:9", DSMI_DEBUG_SPIN_LIMIT="3000", **({} if form == "auto" else {"DSMI_RNN_KERNEL": form})):
        m = native.NativeModel(cfg, sd)
    m.set_inflight(2)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        p, ol = m.forward(_dev(x), lens, check=False)
        assert m.status() is True
    assert m.recompute_count() == 1
    pn = p.cpu().numpy()
    for b in range(56):
        np.testing.assert_allclose(pn[b, :ol[b]], ref[b, :ol[b]], rtol=0, atol=1e-4)
    m.close()


def test_four_ring_layers_in_flight_on_four_handles(native):
    """Four handles with batches in flight: each layer is one window on the handle's own gate slot, four side by side."""
    from oracle import torch_port as tp
    cfg = _cfg(800, 2)
    sd = syn.make_state_dict(2, "gru", 800, 2, seed=77, **syn.TALKATIVE)
    models = [native.NativeModel(cfg, sd) for _ in range(4)]
    for m in models:
        m.set_inflight(2)
    streams = [torch.cuda.Stream() for _ in range(4)]
    batches = [_batch(B=B, T=T, seed=80 + k) for k, (B, T) in enumerate([(64, 201), (32, 257), (48, 181), (20, 301)])]
    refs = [tp.forward(sd, cfg, x, lens)[0] for x, lens in batches]
    xs = [_dev(x) for x, _ in batches]
    torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        for k in range(4):
            with torch.cuda.stream(streams[k]):
                outs.append(models[k].forward(xs[k], batches[k][1], check=False))
        for k in range(4):
            assert models[k].status() is False
            p, ol = outs[k]
            pn = p.cpu().numpy()
            for b in range(pn.shape[0]):
                np.testing.assert_allclose(pn[b, :ol[b]], refs[k][b, :ol[b]], rtol=0, atol=1e-4)
    for m in models:
        assert m.recompute_count() == 0
        m.close()
