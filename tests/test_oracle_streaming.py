"""oracle/streaming.py against the reference's own streaming model (tests/golden/g8_streaming.npz)."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import streaming as ost


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
def test_streaming_forward_g8(golden, kind):
    g = golden("g8_streaming")
    tag = "%s_c2" % kind
    chunks = [int(v) for v in g["chunks_" + tag]]
    H, L, ctx = 32, 3, 6
    sd = syn.make_state_dict(2, kind, H, L, bidirectional=False, context=ctx, seed=81, fc_gain=4.0)
    cfg = dict(conv_layers=2, rnn_type=kind, rnn_hidden_size=H, rnn_layers=L, bidirectional=False, context=ctx)
    m = ost.StreamingModel(sd, cfg)
    for utt in range(2):
        for ci, T in enumerate(chunks):
            x = syn.make_features(1, T, seed=8100 + 100 * utt + ci)
            y = m.forward(x, ci == 0, ci == len(chunks) - 1)
            ref = g["probs_%s_u%d_k%d" % (tag, utt, ci)]
            if ci == 0:
                assert y is None and ref.size == 0
            else:
                assert y.shape[1:] == ref.shape
                np.testing.assert_allclose(y[0], ref, rtol=0, atol=2e-6)


def test_streaming_parser_framing_and_adaptive_stats():
    p = ost.StreamingParser()
    rng = np.random.default_rng(3)
    a = np.round(rng.normal(0, 3000, 8640))
    s0 = p.parse_audio(a)
    # 8640 samples: no remainder, 1 + (8640 - 320)//160 = 53 frames; the last hop is kept for the next pass
    assert s0.shape == (161, 53) and len(p.buffer) == 160 and abs(p.alpha - 0.1) < 1e-12
    b = np.round(rng.normal(0, 3000, 6250))          # 160 + 6250 = 6410 = 40*160 + 10 -> 10 extra samples ride along
    s1 = p.parse_audio(b)
    assert s1.shape == (161, 1 + (6400 - 320) // 160) and len(p.buffer) == 170
    # the mix of dataset and input statistics (parsers.py:147-158)
    raw = np.log1p(np.abs(np.fft.rfft(a[np.arange(320)[None] + 160 * np.arange(53)[:, None]] * p.window, axis=1).T.astype(np.complex64)).astype(np.float32))
    m0 = (0 + raw.mean()) / 2
    sd0 = (0 + raw.std()) / 2
    want = (raw - (m0 * 0.1 + 0.9 * p.DATASET_MEAN)) / (sd0 * 0.1 + 0.9 * p.DATASET_STD)
    np.testing.assert_allclose(s0, want, rtol=0, atol=1e-5)
    assert p.parse_audio(np.zeros(100), is_last=True) == [] and p.buffer is None and p.alpha == 0
