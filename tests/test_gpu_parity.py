"""GPU parity tests: the HIP path (through the C ABI, include/dsmi.h) against the golden
vectors captured from the reference and against the CPU oracle on seeded inputs.

Tolerances: the forward pass is fp32 end to end on both sides but sums in a different order
(MFMA k-chains vs BLAS), so stage outputs agree to ~1e-5 and softmax probabilities are held
to 1e-4 (BASELINE.json north_star); greedy transcripts and offsets must be identical.
"""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def native():
    from danspeech_amd import _native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    _native.lib()  # fail loudly if libdsmi.so is missing
    return _native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cfg(cl, kind, H, L, bidir=True, context=20):
    return dict(conv_layers=cl, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, bidirectional=bidir, context=context)


def test_seq_lens_g1(native, golden):
    g = golden("g1_seq_lens")
    for cl in (1, 2, 3):
        m = native.NativeModel(_cfg(cl, "gru", 8, 1), syn.make_state_dict(cl, "gru", 8, 1, seed=1))
        assert np.array_equal(m.seq_lens(g["T"]), g["conv%d" % cl])
        m.close()


@pytest.mark.parametrize("cl", [1, 2, 3])
def test_conv_stack_g2(native, golden, cl):
    g = golden("g2_conv%d" % cl)
    sd = syn.make_state_dict(cl, "gru", 8, 1, seed=int(g["seed"]))
    m = native.NativeModel(_cfg(cl, "gru", 8, 1), sd)
    lens = g["lens"]
    x = syn.make_features(len(lens), int(lens[0]), seed=int(g["x_seed"]))
    for i, L in enumerate(lens):
        x[i, :, :, L:] = 0
    y = m.conv_stack(_dev(x), lens).cpu().numpy()
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=3e-5)
    for i, L in enumerate(g["out_lens"]):
        assert not y[i, :, :, L:].any()
    m.close()


def test_conv_stack_long_ragged_vs_oracle(native):
    """Several time tiles, tiles fully past a clip's length, odd T."""
    from oracle import model as om
    cl = 2
    sd = syn.make_state_dict(cl, "gru", 8, 1, seed=5)
    m = native.NativeModel(_cfg(cl, "gru", 8, 1), sd)
    lens = np.array([333, 150, 20], dtype=np.int32)
    x = syn.make_features(3, 333, seed=6)
    for i, L in enumerate(lens):
        x[i, :, :, L:] = 0
    y = m.conv_stack(_dev(x), lens).cpu().numpy()
    ref = om.conv_stack(sd, x, om.get_seq_lens(lens, cl), cl)
    np.testing.assert_allclose(y, ref, rtol=0, atol=3e-5)
    m.close()


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("bidir", [1, 0])
def test_batch_rnn_g3(native, golden, kind, bidir):
    """BatchRNN golden vectors driven through dsmi_rnn_layer: layer 0 (no BN, I=32) and
    layer 1 (BN, I=H=16) of a 1-conv model whose audio_conf gives n_freq=2."""
    g = golden("g3_batch_rnn")
    H = 16
    audio_conf = dict(sampling_rate=100, window_size=0.02)
    sd = syn.make_state_dict(1, kind, H, 2, bidirectional=bool(bidir), context=3, seed=2)
    for bn in (0, 1):
        tag = "%s_bn%d_bi%d" % (kind, bn, bidir)
        for k in g.files:
            if k.startswith("w_%s__" % tag):
                sd["rnns.%d.%s" % (bn, k.split("__", 1)[1])] = g[k]
    m = native.NativeModel(_cfg(1, kind, H, 2, bool(bidir), 3), sd, audio_conf=audio_conf)
    for bn in (0, 1):
        tag = "%s_bn%d_bi%d" % (kind, bn, bidir)
        y = m.rnn_layer(bn, _dev(g["x_bn%d" % bn]), g["lens"]).cpu().numpy()
        np.testing.assert_allclose(y, g["y_" + tag], rtol=0, atol=5e-6)
        for b, L in enumerate(g["lens"]):
            assert not y[L:, b].any()
    m.close()


def _small_cases():
    for kind in ("gru", "lstm", "rnn"):
        for bidir in (True, False):
            for cl in (1, 2, 3):
                if cl != 2 and not (kind == "gru" and bidir):
                    continue
                yield kind, bidir, cl


@pytest.mark.parametrize("kind,bidir,cl", list(_small_cases()))
def test_forward_small_g4(native, golden, kind, bidir, cl):
    g = golden("g4_forward_small")
    tag = "%s_bi%d_c%d" % (kind, bidir, cl)
    wseed, xseed = [int(v) for v in g["seeds_" + tag]]
    sd = syn.make_state_dict(cl, kind, 32, 3, bidirectional=bidir, context=6, seed=wseed)
    m = native.NativeModel(_cfg(cl, kind, 32, 3, bidir, 6), sd)
    lens = g["lens"]
    x = syn.make_features(3, 120, seed=xseed)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    p, ol = m.forward(_dev(x), lens)
    assert np.array_equal(ol, g["outlens_" + tag])
    np.testing.assert_allclose(p.cpu().numpy(), g["probs_" + tag], rtol=0, atol=1e-4)
    m.close()


def test_forward_full_cfgA_and_greedy_g4_g7(native, golden):
    """Full-size cfgA (2 conv, 5 x BiGRU 800), ragged B=2: probs within 1e-4 of the reference,
    greedy transcript + offsets identical (sharpened logits, margin >= 1e-3)."""
    g = golden("g4_forward_full")
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, fc_gain=8.0)
    m = native.NativeModel(_cfg(2, "gru", 800, 5), sd)
    lens = g["lens"]
    x = syn.make_features(2, 1001, seed=7)
    x[1, :, :, 777:] = 0
    p, ol = m.forward(_dev(x), lens)
    assert np.array_equal(ol, g["out_lens"])
    pn = p.cpu().numpy()
    err = np.abs(pn - g["probs"]).max()
    print("cfgA max |probs - reference| = %.3g" % err)
    assert err < 1e-4
    labels = syn.DANSPEECH_LABELS
    gd = native.NativeDecoder(labels, blank_index=0)
    dec = gd.greedy(p, ol)
    strings = ["".join(labels[i] for i in ids) for ids, _ in dec]
    assert strings == [str(s) for s in g["strings"]]
    assert np.array_equal(dec[0][1], g["off0"]) and np.array_equal(dec[1][1], g["off1"])
    # batch invariance: clip 1 alone gives the same probabilities (MaskConv's purpose, model.py:57-58)
    p1, _ = m.forward(_dev(x[1:2, :, :, :777].copy()), lens[1:2])
    np.testing.assert_allclose(p1.cpu().numpy()[0, :ol[1]], pn[1, :ol[1]], rtol=0, atol=2e-6)
    m.close()


def test_greedy_g5(native, golden):
    g = golden("g5_greedy")
    labels = syn.DANSPEECH_LABELS
    m = native.NativeDecoder(labels, blank_index=labels.index("_"))
    for sizes, skey, okey in ((g["sizes"], "strings", "offsets"), (None, "strings_nosize", "offsets_nosize")):
        dec = m.greedy(_dev(g["probs"]), sizes)
        strings = ["".join(labels[i] for i in ids) for ids, _ in dec]
        assert strings == [str(s) for s in g[skey]]
        for b, (_, off) in enumerate(dec):
            ref = g[okey][b]
            assert np.array_equal(off, ref[ref >= 0])
    m.close()


def test_features_vs_oracle(native):
    from oracle import features as of
    m = native.NativeFrontend()
    clips = [syn.make_clip(0, 160000), syn.make_clip(1, 66944), syn.make_clip(2, 4000), syn.make_clip(3, 161)]
    n = np.array([len(c) for c in clips], dtype=np.int64)
    for dtype in (np.float64, np.float32, np.int16):
        pcm = _dev(np.concatenate(clips).astype(dtype))
        feat, frames = m.features(pcm, n)
        feat = feat.cpu().numpy()
        for b, c in enumerate(clips):
            ref = of.spectrogram(c)
            assert frames[b] == ref.shape[1]
            np.testing.assert_allclose(feat[b, 0, :, :frames[b]], ref, rtol=0, atol=2e-5)
            assert not feat[b, 0, :, frames[b]:].any()
    m.close()


def test_errors(native):
    with pytest.raises(native.DsmiError) as e:
        native.NativeModel(_cfg(4, "gru", 8, 1), {})
    assert e.value.code == native.DSMI_ERR_CONV
    with pytest.raises(native.DsmiError) as e:
        native.NativeModel(_cfg(0, "gru", 8, 1), {})
    assert e.value.code == native.DSMI_ERR_CONV and "0 convolutional layers" in e.value.msg
    with pytest.raises(native.DsmiError) as e:
        native.NativeModel(_cfg(2, "gru", 8, 1), {})
    assert e.value.code == native.DSMI_ERR_NOT_READY
    m = native.NativeModel(_cfg(2, "gru", 8, 1), syn.make_state_dict(2, "gru", 8, 1, seed=1))
    with pytest.raises(native.DsmiError) as e:
        m.forward(_dev(syn.make_features(2, 30)), [20, 30])
    assert e.value.code == native.DSMI_ERR_UNSORTED
    m.close()


def test_features_window_and_pad_variants(native):
    """hann / constant-padding front ends against the oracle's generic path."""
    from oracle import features as of
    import scipy.signal.windows as W
    clip = syn.make_clip(5, 20000)
    n = np.array([len(clip)], dtype=np.int64)
    fe = native.NativeFrontend(pad_mode="constant")
    feat, fr = fe.features(_dev(clip), n)
    np.testing.assert_allclose(feat.cpu().numpy()[0, 0], of.spectrogram(clip, pad_mode="constant"), rtol=0, atol=2e-5)
    fe.close()
    fe = native.NativeFrontend(dict(normalize=False))
    feat, fr = fe.features(_dev(clip), n)
    np.testing.assert_allclose(feat.cpu().numpy()[0, 0], of.spectrogram(clip, normalize=False), rtol=0, atol=2e-5)
    fe.close()


def test_features_both_stft_kernels(native):
    """n_fft = 320 with float64 / float32 / int16 samples runs on the float64 matrix pipe (stft_mfma_kernel); every other window
    length takes the direct kernel (stft_logmag_kernel): 8 kHz audio_conf -> n_fft = 160, against the oracle; and a clip whose frame
    count is one past a 64-frame workgroup and a 16-frame tile of the matrix-pipe form."""
    from oracle import features as of
    clip = syn.make_clip(7, 30000)
    n = np.array([len(clip)], dtype=np.int64)
    fe = native.NativeFrontend(dict(sampling_rate=8000))
    feat, fr = fe.features(_dev(clip), n)
    ref = of.spectrogram(clip, sample_rate=8000)
    assert fr[0] == ref.shape[1]
    np.testing.assert_allclose(feat.cpu().numpy()[0, 0, :, :fr[0]], ref, rtol=0, atol=2e-5)
    fe.close()
    fe = native.NativeFrontend()
    for ns in (64 * 160, 64 * 160 - 1, 80 * 160, 16 * 160 + 5):          # 65, 64, 81, 17 frames
        c = syn.make_clip(8, ns)
        feat, fr = fe.features(_dev(c), np.array([ns], dtype=np.int64))
        ref = of.spectrogram(c)
        assert fr[0] == ref.shape[1]
        np.testing.assert_allclose(feat.cpu().numpy()[0, 0, :, :fr[0]], ref, rtol=0, atol=2e-5)
    fe.close()


@pytest.mark.parametrize("H,why,gen1_launches", [
    (904, "not a multiple of 16: first-generation persistent kernel, 10 k-pairs per wave", 1),
    (1200, "config 4 width: second generation, 75 workgroups x 2 directions, batch tiles walked in turn", None),
    (1200, "persist8: first generation, 2 x 150 workgroups > 256 CUs, one launch per direction", 2),
    (100, "H % 8 != 0: per-step fp32-MFMA fallback", None),
    (1288, "H > 1280: per-step fp32-MFMA fallback", None)])
def test_wide_and_odd_layers_vs_oracle(native, monkeypatch, H, why, gen1_launches):
    from oracle import model as om
    if why.startswith("persist8"):
        monkeypatch.setenv("DSMI_RNN_MODE", "persist8")
    sd = syn.make_state_dict(2, "gru", H, 1, seed=33, fc_gain=4.0)
    cfg = _cfg(2, "gru", H, 1)
    m = native.NativeModel(cfg, sd)
    m.set_profiling(2)
    lens = np.array([60, 41], dtype=np.int32)
    x = syn.make_features(2, 60, seed=34)
    x[1, :, :, 41:] = 0
    p, ol = m.forward(_dev(x), lens)
    ref, ol_ref = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, ol_ref)
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=0, atol=1e-4)
    ks = m.kernel_stats()
    if "fallback" in why:
        assert ks["rnn_step"]["launches"] == int(ol[0]) and "rnn_layer_persistent" not in ks
    else:
        assert ks["rnn_layer_persistent"]["launches"] == (gen1_launches or 1) and "rnn_step" not in ks
        assert m.recompute_count() == 0
    m.close()


def test_first_and_second_generation_persistent_kernels_agree(native, monkeypatch):
    """DSMI_RNN_MODE=persist8 (8 units per workgroup, 32-clip tiles) vs the default second generation (16 units,
    16-clip tiles side by side) on a 40-clip ragged batch: same split-fp16 products, different summation splits."""
    sd = syn.make_state_dict(2, "gru", 96, 2, seed=37, fc_gain=4.0)
    cfg = _cfg(2, "gru", 96, 2)
    rng = np.random.default_rng(38)
    lens = np.sort(rng.integers(5, 80, size=40))[::-1].astype(np.int32).copy()
    lens[0] = 80
    x = syn.make_features(40, 80, seed=39)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    m2 = native.NativeModel(cfg, sd)
    p2, _ = m2.forward(_dev(x), lens)
    monkeypatch.setenv("DSMI_RNN_MODE", "persist8")
    m1 = native.NativeModel(cfg, sd)
    p1, _ = m1.forward(_dev(x), lens)
    np.testing.assert_allclose(p1.cpu().numpy(), p2.cpu().numpy(), rtol=0, atol=2e-5)
    m1.close(); m2.close()


def test_step_path_and_persistent_path_agree(native, monkeypatch):
    """DSMI_RNN_MODE=steps (fp32 MFMA, one launch per step) vs the default split-fp16 kernels."""
    sd = syn.make_state_dict(2, "lstm", 64, 2, seed=35, fc_gain=4.0)
    cfg = _cfg(2, "lstm", 64, 2)
    lens = np.array([90, 77, 30], dtype=np.int32)
    x = syn.make_features(3, 90, seed=36)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    m1 = native.NativeModel(cfg, sd)
    p1, _ = m1.forward(_dev(x), lens)
    monkeypatch.setenv("DSMI_RNN_MODE", "steps")
    monkeypatch.setenv("DSMI_DENSE_MODE", "f32")
    m2 = native.NativeModel(cfg, sd)
    p2, _ = m2.forward(_dev(x), lens)
    np.testing.assert_allclose(p1.cpu().numpy(), p2.cpu().numpy(), rtol=0, atol=2e-5)
    m1.close(); m2.close()


@pytest.mark.parametrize("kind,B", [("gru", 70), ("lstm", 64), ("rnn", 33)])
def test_multi_tile_batches_in_the_persistent_kernel_vs_oracle(native, kind, B):
    """B > 32: every workgroup of the persistent kernel walks several 32-clip tiles per step
    (ragged last tile, ragged lengths); compared with the oracle and with the per-step path."""
    from oracle import model as om
    H = 48
    sd = syn.make_state_dict(2, kind, H, 2, seed=41, fc_gain=4.0)
    cfg = _cfg(2, kind, H, 2)
    rng = np.random.default_rng(42)
    lens = rng.integers(9, 50, size=B).astype(np.int32)
    lens[0] = 50
    lens = np.sort(lens)[::-1].copy()
    x = syn.make_features(B, 50, seed=43)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    ref, ol_ref = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, ol_ref)
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=0, atol=1e-4)
    m.close()


def test_extreme_lengths_vs_oracle(native):
    """One-frame and very short clips next to a 30 s clip (T = 3001, T' = 1501: BASELINE config 5's length) in one
    ragged batch: counters, masks and packed-sequence zeros at both ends of the length range."""
    from oracle import model as om
    H = 40
    sd = syn.make_state_dict(2, "gru", H, 2, seed=51, fc_gain=4.0)
    cfg = _cfg(2, "gru", H, 2)
    lens = np.array([3001, 700, 12, 2, 1], dtype=np.int32)
    x = syn.make_features(len(lens), 3001, seed=52)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    ref, ol_ref = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, ol_ref) and ol[0] == 1501 and ol[-1] == 1
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=0, atol=1e-4)
    # a batch of one single-frame clip
    p1, ol1 = m.forward(_dev(x[4:5, :, :, :1].copy()), lens[4:5])
    np.testing.assert_allclose(p1.cpu().numpy()[0, :1], ref[4, :1], rtol=0, atol=1e-5)
    m.close()


@pytest.mark.parametrize("kind,H,L,cl,bidir", [("lstm", 800, 2, 2, True),        # full-width LSTM: 4 gates x 8 units fill the MFMA row tile
                                               ("rnn", 800, 2, 2, True),         # tanh RNN at full width
                                               ("gru", 1200, 3, 3, True),        # docstring-Primary shape: 3 conv (96-channel third layer), wide layers
                                               ("gru", 800, 2, 2, False)])       # unidirectional + Lookahead(context 20) at full width
def test_full_width_variants_vs_oracle(native, kind, H, L, cl, bidir):
    from oracle import model as om
    sd = syn.make_state_dict(cl, kind, H, L, bidirectional=bidir, context=20, seed=61, fc_gain=6.0)
    cfg = _cfg(cl, kind, H, L, bidir, 20)
    lens = np.array([201, 150, 64], dtype=np.int32)
    x = syn.make_features(3, 201, seed=62)
    for b, Lb in enumerate(lens):
        x[b, :, :, Lb:] = 0
    m = native.NativeModel(cfg, sd)
    p, ol = m.forward(_dev(x), lens)
    ref, ol_ref = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, ol_ref)
    err = np.abs(p.cpu().numpy() - ref).max()
    print("%s H=%d L=%d conv=%d bidir=%d: max |probs - oracle| = %.3g" % (kind, H, L, cl, bidir, err))
    assert err < 1e-4
    m.close()


@pytest.mark.parametrize("kind,B,why,H,T", [
    ("gru", 20, "two tiles per workgroup: deferred signalling", 1024, 44),
    ("gru", 40, "three tiles per workgroup: the tile-walking kernel, cell waves and feeder waves", 1024, 44),
    ("gru", 64, "four tiles: a feeder requests the state two instances ahead", 1024, 44),
    ("gru", 104, "seven tiles, an odd number of tile instances", 1024, 46),
    ("rnn", 72, "five tiles, one gate", 1024, 44),
    ("lstm", 56, "four tiles, LSTM cell state carried per tile (one operand set: every wave polls)", 1024, 44),
    ("gru", 56, "config 4's width: five k-blocks per wave, eight W_hh fragments per wave in LDS, ragged last tile", 1200, 44),
    ("gru", 40, "config 4's width, three tiles and an odd number of steps: the last instance stands alone", 1200, 46),
    ("gru", 128, "config 4's width, eight tiles per workgroup (as many as a workgroup carries)", 1200, 30)])
def test_second_generation_walks_several_tiles_vs_oracle(native, kind, B, why, H, T):
    """H = 1024 / 1200: the workgroups of both directions leave room for ONE tile group on 256 CUs, so every workgroup of
    rnn_persist16.hip walks ceil(B / 16) batch tiles per step (ragged lengths, ragged last tile)."""
    from oracle import model as om
    sd = syn.make_state_dict(2, kind, H, 1, seed=44, fc_gain=4.0)
    cfg = _cfg(2, kind, H, 1)
    rng = np.random.default_rng(45)
    lens = np.sort(rng.integers(7, T, size=B))[::-1].astype(np.int32).copy()
    lens[0] = T
    x = syn.make_features(B, T, seed=46)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    m = native.NativeModel(cfg, sd)
    m.set_profiling(2)
    p, ol = m.forward(_dev(x), lens)
    ref, ol_ref = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, ol_ref)
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=0, atol=1e-4)
    assert m.kernel_stats()["rnn_layer_persistent"]["launches"] == 1
    assert m.recompute_count() == 0         # (a hand-off that times out is recomputed: right results would hide a wrong schedule)
    m.close()
