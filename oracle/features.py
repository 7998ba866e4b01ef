"""numpy restatement of ``SpectrogramAudioParser.parse_audio``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows reference
danspeech/audio/parsers.py:43-72:

    D = librosa.stft(y, n_fft=320, hop_length=160, win_length=320, window=scipy hamming)
    spect = log1p(|D|) -> FloatTensor -> (spect - mean) / std   (torch unbiased std)

librosa itself is absent from the build image (PARITY vs librosa UNPINNED); the
librosa semantics restated here are: ``center=True`` pads n_fft//2 samples on both
sides (``reflect`` for the librosa <= 0.9 that danspeech 1.0.4 was written against,
``constant`` for >= 0.10); a *callable* window is evaluated as ``window(n_fft)`` and
is therefore the symmetric Hamming window; frames start at ``t * hop``; the rFFT is
taken in float64 (``y`` is float64) and stored as complex64; ``np.abs`` / ``np.log1p``
then run in float32.
"""
import numpy as np


def hamming_sym(n):
    k = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))


def n_frames(n_samples, hop=160):
    return 1 + n_samples // hop


def spectrogram(y, sample_rate=16000, window_size=0.02, window_stride=0.01, normalize=True,
                pad_mode="reflect"):
    """float64[N] -> float32[n_fft//2+1, 1 + N//hop]."""
    n_fft = int(sample_rate * window_size)   # parsers.py:47
    hop = int(sample_rate * window_stride)   # parsers.py:48
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    T = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(T)[None, :]
    frames = yp[idx] * hamming_sym(n_fft)[:, None]
    D = np.fft.rfft(frames, axis=0).astype(np.complex64)
    spect = np.log1p(np.abs(D)).astype(np.float32)          # parsers.py:62-64
    if normalize:                                            # parsers.py:66-70
        mean = np.float32(spect.mean(dtype=np.float64))
        std = np.float32(spect.std(dtype=np.float64, ddof=1))
        spect = ((spect - mean) / std).astype(np.float32)
    return spect
