"""Pin oracle/torch_port.py (the torch-CPU restatement used for the large workloads and for
bench.py's cpu_baseline) to the golden vectors captured from the reference itself.  CPU-only."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import torch_port as tp
from oracle import model as om

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("cl", [1, 2, 3])
def test_g2_conv_stack(golden, cl):
    g = golden("g2_conv%d" % cl)
    sd = syn.make_state_dict(cl, "gru", 8, 1, seed=int(g["seed"]))
    lens = g["lens"]
    x = syn.make_features(len(lens), int(lens[0]), seed=int(g["x_seed"]))
    for i, L in enumerate(lens):
        x[i, :, :, L:] = 0
    y = tp.conv_stack(sd, torch.from_numpy(x), om.get_seq_lens(lens, cl), cl).numpy()
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
@pytest.mark.parametrize("bn", [0, 1])
@pytest.mark.parametrize("bidir", [0, 1])
def test_g3_batch_rnn(golden, kind, bn, bidir):
    g = golden("g3_batch_rnn")
    tag = "%s_bn%d_bi%d" % (kind, bn, bidir)
    sd = {"rnns.0." + k.split("__", 1)[1]: g[k] for k in g.files if k.startswith("w_%s__" % tag)}
    y = tp.batch_rnn(sd, 0, kind, torch.from_numpy(g["x_bn%d" % bn]), g["lens"], bool(bidir), bool(bn)).numpy()
    np.testing.assert_allclose(y, g["y_" + tag], rtol=0, atol=1e-6)


def _small_cases():
    for kind in ("gru", "lstm", "rnn"):
        for bidir in (True, False):
            for cl in (1, 2, 3):
                if cl != 2 and not (kind == "gru" and bidir):
                    continue
                yield kind, bidir, cl


@pytest.mark.parametrize("kind,bidir,cl", list(_small_cases()))
def test_g4_forward_small(golden, kind, bidir, cl):
    g = golden("g4_forward_small")
    tag = "%s_bi%d_c%d" % (kind, bidir, cl)
    wseed, xseed = [int(v) for v in g["seeds_" + tag]]
    sd = syn.make_state_dict(cl, kind, 32, 3, bidirectional=bidir, context=6, seed=wseed)
    cfg = dict(conv_layers=cl, rnn_type=kind, rnn_hidden_size=32, rnn_layers=3, bidirectional=bidir, context=6)
    lens = g["lens"]
    x = syn.make_features(3, 120, seed=xseed)
    for b, L in enumerate(lens):
        x[b, :, :, L:] = 0
    p, ol = tp.forward(sd, cfg, x, lens)
    assert np.array_equal(ol, g["outlens_" + tag])
    np.testing.assert_allclose(p, g["probs_" + tag], rtol=0, atol=2e-6)


def test_g4_forward_full_cfgA(golden):
    """Full-size cfgA, ragged B = 2, against the reference's own probabilities."""
    g = golden("g4_forward_full")
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, fc_gain=8.0)
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)
    x = syn.make_features(2, 1001, seed=7)
    x[1, :, :, 777:] = 0
    p, ol = tp.forward(sd, cfg, x, g["lens"])
    assert np.array_equal(ol, g["out_lens"])
    assert np.abs(p - g["probs"]).max() < 1e-5


def test_unsorted_lengths_raise():
    sd = syn.make_state_dict(2, "gru", 8, 1, seed=1)
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=8, rnn_layers=1, bidirectional=True, context=20)
    with pytest.raises(RuntimeError):
        tp.forward(sd, cfg, syn.make_features(2, 50), [40, 50])
