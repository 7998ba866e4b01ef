"""``SpectrogramAudioParser`` of the drop-in surface (reference danspeech/audio/parsers.py:37-72).

``parse_audio(recording)`` returns the normalised log-magnitude spectrogram ``[n_freq, T]`` as a
float32 torch tensor; the STFT runs in libdsmi.so (float64 DFT on the GPU, features.hip) and the
tensor stays on the device (the engine's ``.to(device)`` is then a no-op).
"""
import numpy as np


class AudioParser(object):
    def __init__(self, audio_config=None):
        self.audio_config = audio_config
        if not self.audio_config:
            self.audio_config = {}
        self.normalize = self.audio_config.get("normalize", True)
        self.sampling_rate = self.audio_config.get("sampling_rate", 16000)
        self.window = self.audio_config.get("window", "hamming")
        self.window_stride = self.audio_config.get("window_stride", 0.01)
        self.window_size = self.audio_config.get("window_size", 0.02)

    def parse_audio(self, recording):
        raise NotImplementedError


class SpectrogramAudioParser(AudioParser):
    def __init__(self, audio_config=None, device=0, pad_mode="reflect"):
        super(SpectrogramAudioParser, self).__init__(audio_config)
        self.n_fft = int(self.sampling_rate * self.window_size)
        self.hop_length = int(self.sampling_rate * self.window_stride)
        self.device = device
        self.pad_mode = pad_mode          # librosa <= 0.9 'reflect' (era of danspeech 1.0.4), >= 0.10 'constant'
        self._native = None

    def _frontend(self):
        if self._native is None:
            from .. import _native
            conf = dict(sampling_rate=self.sampling_rate, window_size=self.window_size, window_stride=self.window_stride,
                        window=self.window, normalize=self.normalize)
            self._native = _native.NativeFrontend(conf, device=self.device, pad_mode=self.pad_mode)
        return self._native

    def parse_batch(self, recordings):
        """list of 1-D arrays -> (features [B,1,F,Tmax] CUDA float32, frames int32[B]); batched extension."""
        import torch
        n = np.array([len(r) for r in recordings], dtype=np.int64)
        total = int(n.sum())
        # one pinned staging buffer (grow-only): clips are copied into it once and cross PCIe asynchronously
        if getattr(self, "_stage", None) is None or self._stage.numel() < total:
            self._stage = torch.empty(max(total, 1), dtype=torch.float64).pin_memory()
        host = self._stage[:total].numpy()
        off = 0
        for r, k in zip(recordings, n):
            host[off:off + k] = r              # converts to float64 like the reference's parser input (resources.py:640)
            off += int(k)
        pcm = self._stage[:total].to("cuda:%d" % self.device, non_blocking=True)
        out = self._frontend().features(pcm, n)
        torch.cuda.current_stream(self.device).synchronize()      # the staging buffer is reused by the next call
        return out

    def parse_wav_frames(self, raws, width, channels):
        """Raw PCM WAV frames (``read_wav_frames``; one common sample width / channel count) ->
        same result as ``parse_batch([load_audio(f) for f in files])`` with the decoding on the GPU."""
        import torch
        n = np.array([len(r) // (width * channels) for r in raws], dtype=np.int64)
        buf = np.frombuffer(b"".join(raws), dtype=np.uint8)
        pcm = torch.from_numpy(buf.copy()).to("cuda:%d" % self.device)
        return self._frontend().features(pcm, n, wav_format=(width, channels))

    def parse_audio(self, recording):
        feat, frames = self.parse_batch([recording])
        return feat[0, 0, :, :int(frames[0])]


class InferenceSpectrogramAudioParser(AudioParser):
    """The streaming parser (reference danspeech/audio/parsers.py:75-170): spectrograms of consecutive
    parts of a recording with one hop carried over, normalised with statistics that move from the NST
    dataset's to the input's over the first second.

    The sample bookkeeping (parsers.py:112-133) is host logic on a few hundred samples and stays here;
    the STFT (no centre padding), log1p and the adaptive normalisation (parsers.py:136-161) run in
    ``dsmi_features_stream`` and the spectrogram stays on the GPU."""

    def __init__(self, audio_config=None, device=0):
        super(InferenceSpectrogramAudioParser, self).__init__(audio_config)
        self.n_fft = int(self.sampling_rate * self.window_size)
        self.hop_length = int(self.sampling_rate * self.window_stride)
        self.device = device
        self.dataset_mean = 5.492418704733003        # applied inside dsmi_features_stream (parsers.py:89-90)
        self.dataset_std = 1.7552755216970917
        self.alpha_increment = 0.1
        self._state = np.zeros(3, dtype=np.float64)  # input_mean, input_std, alpha
        self.buffer = None
        self.has_buffer = False
        self._native = None

    input_mean = property(lambda self: float(self._state[0]))
    input_std = property(lambda self: float(self._state[1]))
    alpha = property(lambda self: float(self._state[2]))

    def _frontend(self):
        if self._native is None:
            from .. import _native
            conf = dict(sampling_rate=self.sampling_rate, window_size=self.window_size, window_stride=self.window_stride,
                        window=self.window, normalize=self.normalize)
            self._native = _native.NativeFrontend(conf, device=self.device)
        return self._native

    def parse_audio(self, part_of_recording, is_last=False):
        import torch
        if is_last and len(part_of_recording) < self.n_fft:            # parsers.py:106-110
            self.reset()
            return []
        part_of_recording = np.asarray(part_of_recording, dtype=np.float64)
        if self.has_buffer:
            part_of_recording = np.concatenate((self.buffer, part_of_recording), axis=None)
        extra_samples = len(part_of_recording) % self.hop_length
        if extra_samples != 0:
            extra_samples_array = part_of_recording[-extra_samples:]
            part_of_recording = part_of_recording[:-extra_samples]
        self.buffer = part_of_recording[-self.hop_length:]
        if extra_samples != 0:
            self.buffer = np.concatenate((self.buffer, extra_samples_array), axis=None)
        self.has_buffer = True
        pcm = torch.from_numpy(np.ascontiguousarray(part_of_recording)).to("cuda:%d" % self.device)
        return self._frontend().features_stream(pcm, self._state)

    def reset(self):
        self.buffer = None
        self.has_buffer = False
        self._state[:] = 0
