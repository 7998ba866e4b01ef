"""oracle.features cross-checked against an independent STFT (torch.stft); parity vs
librosa itself is UNPINNED (librosa absent, no reference test covers it)."""
import numpy as np
import torch

from danspeech_amd import synthetic as syn
from oracle import features as of


def _torch_spect(y, pad_mode):
    win = torch.from_numpy(of.hamming_sym(320))
    D = torch.stft(torch.from_numpy(y), n_fft=320, hop_length=160, win_length=320, window=win,
                   center=True, pad_mode=pad_mode, return_complex=True)
    s = torch.log1p(D.abs().float())
    return ((s - s.mean()) / s.std()).numpy()


def test_shape_and_torch_stft_agreement():
    y = syn.make_clip(0, 160000)
    for pad_mode in ("reflect", "constant"):
        s = of.spectrogram(y, pad_mode=pad_mode)
        assert s.shape == (161, 1001) and s.dtype == np.float32
        np.testing.assert_allclose(s, _torch_spect(y, pad_mode), rtol=0, atol=2e-5)
        assert abs(float(s.mean())) < 1e-5 and abs(float(s.std(ddof=1)) - 1) < 1e-5


def test_short_and_odd_lengths():
    for n in (161, 320, 4000, 66944):
        y = syn.make_clip(3, n)
        s = of.spectrogram(y)
        assert s.shape == (161, 1 + n // 160)
        np.testing.assert_allclose(s, _torch_spect(y, "reflect"), rtol=0, atol=2e-5)


def test_window_matches_scipy():
    import scipy.signal.windows as W
    np.testing.assert_allclose(of.hamming_sym(320), W.hamming(320), rtol=0, atol=1e-15)
