"""``SpectrogramAudioParser`` of the drop-in surface (reference danspeech/audio/parsers.py:37-72).

``parse_audio(recording)`` returns the normalised log-magnitude spectrogram ``[n_freq, T]`` as a
float32 torch tensor; the STFT runs in libdsmi.so (float64 DFT on the GPU, features.hip) and the
tensor stays on the device (the engine's ``.to(device)`` is then a no-op).
"""

import numpy as np


_COPY_STREAMS = {}


def _shared_copy_stream(device):
    """ONE upload stream per device for the parsers of a pipeline that also runs beam searches on a decode stream: HIP
    gives a process a handful of hardware queues (four by default) and maps streams onto them round robin, so every
    further stream is a chance that two of the pipeline's streams -- a forward and the other batch's beam search, say --
    share a queue and run one after the other (measured: 23 ms per batch instead of 15).  A greedy pipeline has no such
    kernel on its decode stream and is 7 % faster with an upload stream per parser (8.99 against 9.64 ms per batch)."""
    import torch
    if device not in _COPY_STREAMS:
        _COPY_STREAMS[device] = torch.cuda.Stream(device=device)
    return _COPY_STREAMS[device]


class StagedClips(object):
    """Host clips on their way to the device: ``SpectrogramAudioParser.stage`` has copied them into a pinned buffer and started
    the upload on the copy stream; ``parse_batch`` makes the current stream wait for it and runs the spectrograms.  Lets a
    pipeline start a batch's upload before it has a model handle free for it."""

    def __init__(self, pcm, n_samples, itemsize, done, slot=None):
        self.pcm, self.n_samples, self.itemsize, self.done, self.slot = pcm, n_samples, itemsize, done, slot

    def __len__(self):
        return len(self.n_samples)


class DeviceClips(object):
    """A batch whose clips already lie back to back in GPU memory, longest first: ``pcm`` = 1-D CUDA tensor (int16, float32
    or float64 samples), ``n_samples`` = their lengths.  What an RCCL scatter delivers (``parallel.recognize_sharded``) and
    what a caller that keeps its audio on the device passes to ``recognize_batch`` / ``recognize_batches`` instead of a
    list of host arrays: no staging, no upload."""

    def __init__(self, pcm, n_samples):
        self.pcm = pcm
        self.n_samples = np.asarray(n_samples, dtype=np.int64)
        if self.n_samples.ndim != 1 or (np.diff(self.n_samples) > 0).any():
            raise ValueError("DeviceClips: clips must be ordered longest first (pack_padded_sequence's order, reference model.py:117)")
        if int(self.n_samples.sum()) != pcm.numel():
            raise ValueError("DeviceClips: the lengths do not add up to the buffer")

    def __len__(self):
        return len(self.n_samples)

    def part(self, lo, hi):
        """Clips lo..hi-1 as a batch of their own (a view)."""
        offs = np.concatenate(([0], np.cumsum(self.n_samples)))
        return DeviceClips(self.pcm[int(offs[lo]):int(offs[hi])], self.n_samples[lo:hi])

    @staticmethod
    def merge(batches):
        """Several batches (one sample type) as one, longest first (a stable merge: every batch is ordered already); one gather on
        the current stream."""
        import torch
        n = np.concatenate([b.n_samples for b in batches])
        order = np.argsort(-n, kind="stable")
        pieces, k = [], 0
        for b in batches:
            offs = np.concatenate(([0], np.cumsum(b.n_samples)))
            pieces += [b.pcm[int(offs[i]):int(offs[i + 1])] for i in range(len(b))]
        merged = DeviceClips(torch.cat([pieces[i] for i in order]), n[order])
        merged.order = order            # merged clip i is clip order[i] of the batches laid end to end
        return merged


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

    # sample types dsmi_features reads directly; anything else (and mixed batches) goes up as float64, the type
    # load_audio hands to recognize() (reference resources.py:640) -- the conversion is exact for all three
    _NATIVE_PCM = (np.dtype(np.int16), np.dtype(np.float32), np.dtype(np.float64))
    # staging slots of every parser of the process are sized alike (see _staging): the mark, its ceiling and its lock
    pack_int16 = True      # stage(): float64 clips whose samples are int16 integers are uploaded as int16 (False: always as they come)
    _stage_high = 1 << 20
    _STAGE_SHARED_CAP = 256 << 20
    _stage_lock = __import__("threading").Lock()

    def _staging(self, nbytes):
        """Two pinned host buffers used alternately, each with the event of the upload that last read it: the host
        fills one while the copy engine still drains the other, and nothing waits for the GPU's compute stream."""
        import torch
        if getattr(self, "_slots", None) is None:
            self._slots = [dict(buf=None, done=None, dev=None, used=None), dict(buf=None, done=None, dev=None, used=None)]
            self._turn = 0
            self._copy_stream = None         # made when an upload asks for it: every stream of the process takes a place on the hardware queues
        slot = self._slots[self._turn]
        self._turn ^= 1
        if slot["done"] is not None:
            slot["done"].synchronize()                    # the upload issued two batches ago
        if getattr(self, "upload_on_compute_stream", False) and slot["used"] is not None:
            slot["used"].synchronize()                    # ... or the forward that uploaded from this buffer itself
        # Sized by the largest forward ANY parser of the process has staged, not by this slot's own history: a pipeline's lanes take
        # forwards of different sizes in turn (merged pairs, single batches at a call's end), and a slot that met only small ones
        # re-pinned 82 MB (16 ms on the staging thread, the device buffer and a blocking first upload behind it) in the middle of a
        # later call, when its first large forward arrived (profiles/r05_fill_drain.txt).  The shared mark stops at _STAGE_SHARED_CAP
        # (a pipeline forward of 64 clips x 30 s as float64 is 246 MB): one very long recording staged as a single clip sizes the slot
        # it went through, not every slot of every parser for the rest of the process.  Helper threads stage concurrently: locked.
        cls = SpectrogramAudioParser
        with cls._stage_lock:
            if nbytes <= cls._STAGE_SHARED_CAP:
                cls._stage_high = max(cls._stage_high, nbytes)
            high = max(cls._stage_high, nbytes)
        if slot["buf"] is None or slot["buf"].numel() < high:
            slot["buf"] = torch.empty(high, dtype=torch.uint8).pin_memory()
        if slot["dev"] is not None and slot["dev"].numel() < high:
            if slot["used"] is not None:
                slot["used"].synchronize()   # the kernels that read the buffer being given back
            slot["dev"] = None               # (re-made at `high` where it is next needed)
        slot["high"] = high
        return slot

    def _upload(self, dst, src_pinned, stream):
        """Pinned host bytes -> device bytes on ``stream`` by ``dsmi_upload`` (a kernel that reads the pinned buffer over the bus): the
        pipeline's uploads do not go through hipMemcpyAsync, whose first copies on a process's streams hold the calling thread for
        6-12 ms each (profiles/r06_second_call_stall.txt)."""
        from .. import _native
        rc = _native.lib().dsmi_upload(int(self.device), dst.data_ptr(), src_pinned.data_ptr(), int(src_pinned.numel()), int(stream.cuda_stream))
        if rc != 0:
            raise _native.DsmiError(rc, "dsmi_upload failed")

    def stage(self, recordings):
        """list of 1-D arrays -> ``StagedClips``: the clips copied back to back into a pinned buffer (two buffers, used
        alternately) and their upload started on the copy stream.  Touches no stream but the copy stream."""
        import torch
        recordings = [np.asarray(r) for r in recordings]
        kinds = {r.dtype for r in recordings}
        dtype = kinds.pop() if len(kinds) == 1 and next(iter(kinds)) in self._NATIVE_PCM else np.dtype(np.float64)
        n = np.array([len(r) for r in recordings], dtype=np.int64)
        total = int(n.sum())
        slot = self._staging(total * dtype.itemsize)
        offs = np.concatenate(([0], np.cumsum(n)))

        def fill(work):
            if total * dtype.itemsize >= (8 << 20) and len(recordings) >= 8:
                # tens of megabytes of float64 per batch: four host threads fill the pinned buffer (5 ms -> 1.5 ms for 32 x 10 s)
                if getattr(self, "_pool", None) is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=4)
                cut = np.linspace(0, len(recordings), 5).astype(int)
                return list(self._pool.map(lambda ab: work(*ab), zip(cut[:-1], cut[1:])))
            return [work(0, len(recordings))]

        # float64 clips -- what load_audio hands to recognize() (reference resources.py:640) -- are integers in int16's range for every
        # audio file: they travel as int16, a quarter of the bytes (82 MB at 14 GB/s is 5.8 ms of a 64-clip forward's life, and all of
        # a short call's start), and dsmi_features widens them exactly: the same features bit for bit.  dsmi_pack_pcm_i16 says whether
        # a clip qualifies while it converts it; one that does not sends the whole batch as float64.
        packed = False
        if dtype == np.float64 and self.pack_int16 and all(r.ndim == 1 and r.flags.c_contiguous for r in recordings):
            from .. import _native
            pack = _native.lib().dsmi_pack_pcm_i16
            host16 = slot["buf"][:total * 2].numpy().view(np.int16)

            def pack_some(lo, hi):
                for i in range(lo, hi):
                    if not pack(recordings[i].ctypes.data, int(n[i]), host16[offs[i]:].ctypes.data):       # (ctypes releases the GIL)
                        return False
                return True
            if all(fill(pack_some)):
                dtype, packed = np.dtype(np.int16), True
        if not packed:
            host = slot["buf"][:total * dtype.itemsize].numpy().view(dtype)

            def copy(lo, hi):
                for i in range(lo, hi):
                    host[offs[i]:offs[i + 1]] = recordings[i]          # (converts to `dtype`; numpy releases the GIL while it copies)
            fill(copy)
        # the parser's own upload stream, or the one all parsers of the device share (share_copy_stream: set by a pipeline
        # that also keeps a decode stream busy -- see _shared_copy_stream)
        nbytes = total * dtype.itemsize
        if getattr(self, "upload_on_compute_stream", False):
            # the pipeline's choice: no copy stream at all -- parse_batch uploads on the stream that runs the forward (1.7 ms of a
            # 40 ms forward, while the other forwards in flight keep the device busy).  A forward enqueued behind a cross-stream wait
            # for an upload on another stream found that upload taking 20 ms; measured again at the end of round 4 with
            # five streams in the process (no two on one hardware queue): one shared upload stream 7.33 ms per batch, one per parser
            # 7.58, this 5.89-5.98.
            if slot["dev"] is None or slot["dev"].numel() < nbytes:
                slot["dev"] = torch.empty(max(nbytes, slot["high"]), dtype=torch.uint8, device="cuda:%d" % self.device)
            up = getattr(self, "upload_stream", None)
            if up is not None:
                # the lane's stream is known: the upload goes into it from HERE (the staging thread), behind the lane's running
                # forward and ahead of this one's kernels -- it starts the moment that forward ends, and a hipMemcpyAsync that
                # holds its caller (see DanSpeechRecognizer.transcribe_batches) holds this thread, not the one that feeds the lanes
                pcm = slot["dev"][:nbytes]
                self._upload(pcm, slot["buf"][:nbytes], up)
                return StagedClips(pcm, n, dtype.itemsize, None, slot)
            return StagedClips(slot["buf"][:nbytes], n, dtype.itemsize, None, slot)
        if getattr(self, "share_copy_stream", False):
            up = _shared_copy_stream(self.device)
        else:
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            up = self._copy_stream
        with torch.cuda.stream(up):
            # the slot's own device buffer (not a fresh allocation per batch: tens of megabytes allocated on the copy stream and
            # released on the compute stream go round the caching allocator's cross-stream bookkeeping every batch)
            if slot["dev"] is None or slot["dev"].numel() < nbytes:
                slot["dev"] = torch.empty(max(nbytes, slot["high"]), dtype=torch.uint8, device="cuda:%d" % self.device)
            if slot["used"] is not None:
                up.wait_event(slot["used"])                  # the kernels that read this buffer two batches ago
            pcm = slot["dev"][:nbytes]
            pcm.copy_(slot["buf"][:nbytes], non_blocking=True)
            slot["done"] = torch.cuda.Event()
            slot["done"].record(up)
        return StagedClips(pcm, n, dtype.itemsize, slot["done"], slot)

    def parse_batch(self, recordings):
        """list of 1-D arrays (or ``StagedClips`` / ``DeviceClips``) -> (features [B,1,F,Tmax] CUDA float32, frames int32[B]);
        batched extension.  Asynchronous: the upload runs on a copy stream, the kernels on the current stream behind it."""
        import torch
        if isinstance(recordings, DeviceClips):
            return self._frontend().features(recordings.pcm, recordings.n_samples)
        staged = recordings if isinstance(recordings, StagedClips) else self.stage(recordings)
        main = torch.cuda.current_stream(self.device)
        if not staged.pcm.is_cuda:                               # staged only: the upload runs here, on the forward's own stream
            dev = staged.slot["dev"][:staged.pcm.numel()]
            self._upload(dev, staged.pcm, main)
            staged = StagedClips(dev, staged.n_samples, staged.itemsize, None, staged.slot)
        if staged.done is not None:
            main.wait_event(staged.done)
        out = self._frontend().features(staged.pcm.view({2: torch.int16, 4: torch.float32, 8: torch.float64}[staged.itemsize]), staged.n_samples,
                                        device="cuda:%d" % self.device)
        if staged.slot is not None:
            staged.slot["used"] = torch.cuda.Event()
            staged.slot["used"].record(main)                 # the staging slot's device buffer may be overwritten behind this point
        else:
            staged.pcm.record_stream(main)
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

    The sample bookkeeping (which samples are analysed now, which are carried to the next call) is host logic on a
    few hundred samples and stays here;
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
        """Spectrogram of the next part of the utterance.  The samples not yet analysed are the carry of the previous
        call plus this part; whole hops of them are analysed now, and the last analysed hop plus the remainder are
        carried over so that consecutive parts tile the signal exactly like one long STFT without centre padding.
        A closing part shorter than one window ends the utterance with nothing."""
        import torch
        if is_last and len(part_of_recording) < self.n_fft:
            self.reset()
            return []
        samples = np.asarray(part_of_recording, dtype=np.float64).reshape(-1)
        if self.has_buffer:
            samples = np.concatenate((self.buffer, samples))
        usable = len(samples) - len(samples) % self.hop_length
        self.buffer = samples[max(usable - self.hop_length, 0):].copy()
        self.has_buffer = True
        pcm = torch.from_numpy(np.ascontiguousarray(samples[:usable])).to("cuda:%d" % self.device)
        return self._frontend().features_stream(pcm, self._state)

    def reset(self):
        self.buffer = None
        self.has_buffer = False
        self._state[:] = 0
