"""Pin the CPU oracle against golden vectors captured from the reference itself
(tools/gen_golden.py).  CPU-only."""
import hashlib

import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import model as om
from oracle import decoder as od


def sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def test_g1_seq_lens(golden):
    g = golden("g1_seq_lens")
    for cl in (1, 2, 3):
        assert np.array_equal(om.get_seq_lens(g["T"], cl), g["conv%d" % cl])
    assert om.get_seq_lens([1001], 2)[0] == 501 and om.get_seq_lens([3001], 2)[0] == 1501


@pytest.mark.parametrize("cl", [1, 2, 3])
def test_g2_conv_stack(golden, cl):
    g = golden("g2_conv%d" % cl)
    sd = syn.make_state_dict(cl, "gru", 8, 1, seed=int(g["seed"]))
    assert sd_hash(sd) == str(g["sd_sha"]), "synthetic weight generator drifted"
    lens = g["lens"]
    x = syn.make_features(len(lens), int(lens[0]), seed=int(g["x_seed"]))
    for i, L in enumerate(lens):
        x[i, :, :, L:] = 0
    out_lens = om.get_seq_lens(lens, cl)
    assert np.array_equal(out_lens, g["out_lens"])
    y = om.conv_stack(sd, x, out_lens, cl)
    assert y.shape == g["y"].shape
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=2e-5)
    # masked region is exactly zero in both
    for i, L in enumerate(out_lens):
        assert not y[i, :, :, L:].any() and not g["y"][i, :, :, L:].any()


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("bn", [0, 1])
@pytest.mark.parametrize("bidir", [0, 1])
def test_g3_batch_rnn(golden, kind, bn, bidir):
    g = golden("g3_batch_rnn")
    tag = "%s_bn%d_bi%d" % (kind, bn, bidir)
    sd = {}
    for k in g.files:
        if k.startswith("w_%s__" % tag):
            name = k.split("__", 1)[1]
            sd["rnns.0." + name] = g[k]
    y = om.batch_rnn(sd, 0, kind, g["x_bn%d" % bn], g["lens"], bool(bidir), bool(bn))
    ref = g["y_" + tag]
    if not bidir:
        assert ref.shape[2] == 16
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6)
    for b, L in enumerate(g["lens"]):
        assert not y[L:, b].any()


def _small_cases():
    i = 0
    for kind in ("gru", "lstm", "rnn"):
        for bidir in (True, False):
            for cl in (1, 2, 3):
                if cl != 2 and not (kind == "gru" and bidir):
                    continue
                i += 1
                yield kind, bidir, cl


@pytest.mark.parametrize("kind,bidir,cl", list(_small_cases()))
def test_g4_forward_small(golden, kind, bidir, cl):
    g = golden("g4_forward_small")
    tag = "%s_bi%d_c%d" % (kind, bidir, cl)
    wseed, xseed = [int(v) for v in g["seeds_" + tag]]
    sd = syn.make_state_dict(cl, kind, 32, 3, bidirectional=bidir, context=6, seed=wseed)
    assert sd_hash(sd) == str(g["sha_" + tag])
    cfg = dict(conv_layers=cl, rnn_type=kind, rnn_hidden_size=32, rnn_layers=3, bidirectional=bidir, context=6)
    lens = g["lens"]
    x = syn.make_features(3, 120, seed=xseed)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    p, ol = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, g["outlens_" + tag])
    np.testing.assert_allclose(p, g["probs_" + tag], rtol=0, atol=1e-5)


def test_g4_forward_full_and_greedy(golden):
    """cfgA full size (2 conv, 5 x BiGRU 800), ragged B=2; G7 sharpened logits."""
    g = golden("g4_forward_full")
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, fc_gain=8.0)
    assert sd_hash(sd) == str(g["sha"])
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
    lens = g["lens"]
    x = syn.make_features(2, 1001, seed=7)
    x[1, :, :, 777:] = 0
    p, ol = om.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, g["out_lens"])
    np.testing.assert_allclose(p, g["probs"], rtol=0, atol=1e-4)
    strings, offsets = od.greedy_decode(p, ol, syn.DANSPEECH_LABELS, 0)
    assert [s[0] for s in strings] == [str(s) for s in g["strings"]]
    assert np.array_equal(offsets[0][0], g["off0"]) and np.array_equal(offsets[1][0], g["off1"])


def test_unsorted_lengths_raise():
    sd = syn.make_state_dict(2, "gru", 8, 1, seed=1)
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=8, rnn_layers=1, bidirectional=True, context=20)
    with pytest.raises(RuntimeError):
        om.forward(sd, cfg, syn.make_features(2, 30), [20, 30])


def test_g5_greedy(golden):
    g = golden("g5_greedy")
    labels = syn.DANSPEECH_LABELS
    strings, offsets = od.greedy_decode(g["probs"], g["sizes"], labels, labels.index("_"))
    assert [s[0] for s in strings] == [str(s) for s in g["strings"]]
    for b, o in enumerate(offsets):
        ref = g["offsets"][b]
        assert np.array_equal(o[0], ref[ref >= 0])
    strings, offsets = od.greedy_decode(g["probs"], None, labels, labels.index("_"))
    assert [s[0] for s in strings] == [str(s) for s in g["strings_nosize"]]
    for b, o in enumerate(offsets):
        ref = g["offsets_nosize"][b]
        assert np.array_equal(o[0], ref[ref >= 0])


def test_flops_formula():
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
    assert abs(om.flops_per_clip(cfg, 501) / 1e9 - 51.85) < 0.05
    cfgb = dict(cfg, rnn_hidden_size=1200, rnn_layers=7)
    assert abs(om.flops_per_clip(cfgb, 501) / 1e9 - 132.9) < 0.1
