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


def test_scipy_stft_agreement():
    """A second independent STFT (scipy.signal.stft, its 'even' boundary extension = numpy/librosa 'reflect' padding,
    'zeros' = librosa >= 0.10's constant padding), un-scaled by the window sum."""
    import scipy.signal as ss
    win = of.hamming_sym(320)
    for n in (160000, 66944, 4000):
        y = syn.make_clip(5, n)
        for pad_mode, boundary in (("reflect", "even"), ("constant", "zeros")):
            _, _, Z = ss.stft(y, fs=16000, window=win, nperseg=320, noverlap=160, nfft=320, boundary=boundary, padded=False,
                              return_onesided=True)
            Z = Z[:, :1 + n // 160] * win.sum()
            s = np.log1p(np.abs(Z.astype(np.complex64))).astype(np.float32)
            s = (s - s.mean()) / s.std(ddof=1)
            got = of.spectrogram(y, pad_mode=pad_mode)
            assert got.shape == s.shape
            np.testing.assert_allclose(got, s, rtol=0, atol=3e-5)
