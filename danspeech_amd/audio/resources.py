"""Audio file loading for the drop-in surface (reference danspeech/audio/resources.py:22-82).

Only what ``recognize()`` needs: ``load_audio`` for PCM WAV files with the reference's semantics
(4096-frame chunks, optional duration/offset, stereo -> mono as the *saturating sum* L+R of
``audioop.tomono(buf, width, 1, 1)`` at resources.py:302-303, samples returned as float64 at
their integer scale, resources.py:630-640) and ``load_audio_wavPCM`` (channel *mean*).
AIFF/FLAC decoding, the microphone classes and ``AudioData`` are outside the hot path.
"""
import wave

import numpy as np


def _frames_to_int(buf, width):
    if width == 1:      # unsigned 8-bit; AudioData.get_raw_data biases by -128 (resources.py:551-554)
        return np.frombuffer(buf, dtype=np.uint8).astype(np.int64) - 128
    if width == 2:
        return np.frombuffer(buf, dtype="<i2").astype(np.int64)
    if width == 4:
        return np.frombuffer(buf, dtype="<i4").astype(np.int64)
    if width == 3:
        a = np.frombuffer(buf, dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        v = a[:, 0] | (a[:, 1] << 8) | (a[:, 2] << 16)
        return np.where(v >= 1 << 23, v - (1 << 24), v)
    raise ValueError("unsupported sample width %d" % width)


def read_wav_frames(path, duration=None, offset=None):
    """The byte stream ``load_audio`` decodes, undecoded: (raw frames, sample width, channels).

    Same chunked reading as resources.py:41-57 (4096-frame chunks, optional duration/offset at
    the pinned 16 kHz).  The bytes can go to the GPU as they are: ``dsmi_features`` decodes the
    sample width and folds two channels (``Recognizer.recognize_files``)."""
    try:
        reader = wave.open(path, "rb")
    except (wave.Error, EOFError):
        raise ValueError("Audio file could not be read as PCM WAV; AIFF/FLAC decoding is outside this package's scope")
    with reader:
        nch, width = reader.getnchannels(), reader.getsampwidth()
        assert 1 <= nch <= 2, "Audio must be mono or stereo"
        chunk = 4096
        seconds_per_buffer = (chunk + 0.0) / 16000      # SpeechFile pins sampling_rate = 16000 (resources.py:192)
        elapsed_time = 0
        offset_time = 0
        offset_reached = False
        parts = []
        while True:                                      # resources.py:41-57
            if offset and not offset_reached:
                offset_time += seconds_per_buffer
                if offset_time > offset:
                    offset_reached = True
            buf = reader.readframes(chunk)
            if len(buf) == 0:
                break
            if offset_reached or not offset:
                elapsed_time += seconds_per_buffer
                if duration and elapsed_time > duration:
                    break
                parts.append(buf)
    if nch == 2 and width == 1:
        raise ValueError("8-bit stereo WAV is not supported")
    return b"".join(parts), width, nch


def load_audio(path, duration=None, offset=None):
    """PCM WAV -> float64 numpy array ready for ``Recognizer.recognize``."""
    raw, width, nch = read_wav_frames(path, duration=duration, offset=offset)
    data = _frames_to_int(raw, width)
    if nch == 2:
        data = data.reshape(-1, 2)
        lim = 1 << (8 * width - 1)
        data = np.clip(data[:, 0] + data[:, 1], -lim, lim - 1)     # audioop.tomono(buf, width, 1, 1) saturates
    return data.astype(float)


def load_audio_wavPCM(path):
    """resources.py:64-82: scipy-style read, multi-channel averaged."""
    import scipy.io.wavfile as wav
    _, sound = wav.read(path)
    if len(sound.shape) > 1:
        if sound.shape[1] == 1:
            sound = sound.squeeze()
        else:
            sound = sound.mean(axis=1)
    return sound.astype(float)
