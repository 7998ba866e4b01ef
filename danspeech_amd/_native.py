"""ctypes binding of libdsmi.so (include/dsmi.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing, importing this module's users works (so that pure-host
logic can be tested without a GPU) but the first call that needs a kernel raises
``NativeLibraryMissing`` with the build command.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSMI_LIBRARY: another build of the same ABI -- tools/exp/ load lib/libdsmi_exp.so (`make -C danspeech_amd/csrc exp`: the timing
# instantiations and A/B switches that libdsmi.so does not carry)
LIB_PATH = os.environ.get("DSMI_LIBRARY") or os.path.join(_HERE, "lib", "libdsmi.so")

RNN_TYPES = {"gru": 0, "lstm": 1, "rnn": 2}
WINDOWS = {"hamming": 0, "hann": 1, "blackman": 2, "bartlett": 3}
PCM_DTYPES = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}
PAD_MODES = {"reflect": 0, "constant": 1}

DSMI_ERR_INVALID = -1
DSMI_ERR_CONV = -2
DSMI_ERR_NOT_READY = -3
DSMI_ERR_UNSORTED = -4
DSMI_ERR_CAPACITY = -8
DSMI_ERR_TIMEOUT = -9
DSMI_ERR_COMM = -10
DSMI_RECOMPUTED = 1


class NativeLibraryMissing(RuntimeError):
    pass


class DsmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdsmi error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class ModelDesc(C.Structure):
    _fields_ = [
        ("conv_layers", C.c_int32), ("rnn_type", C.c_int32), ("rnn_hidden_size", C.c_int32),
        ("rnn_layers", C.c_int32), ("bidirectional", C.c_int32), ("context", C.c_int32),
        ("n_labels", C.c_int32), ("sample_rate", C.c_int32), ("window_size", C.c_double),
    ]


class FrontendDesc(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32), ("window_size", C.c_double), ("window_stride", C.c_double),
        ("window", C.c_int32), ("normalize", C.c_int32), ("pad_mode", C.c_int32),
    ]


_lib = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p

_PROTOS = {
    "dsmi_model_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_vp)]),
    "dsmi_model_load_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "dsmi_model_finalize": (C.c_int, [_vp]),
    "dsmi_reserve": (C.c_int, [_vp, C.c_int, C.c_int]),
    "dsmi_model_destroy": (None, [_vp]),
    "dsmi_last_error": (C.c_char_p, [_vp]),
    "dsmi_seq_lens": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "dsmi_frontend_create": (C.c_int, [C.POINTER(FrontendDesc), C.c_int, C.POINTER(_vp)]),
    "dsmi_frontend_destroy": (None, [_vp]),
    "dsmi_frontend_last_error": (C.c_char_p, [_vp]),
    "dsmi_features": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "dsmi_features_stream": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, C.c_int, _vp, _vp]),
    "dsmi_stream_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dsmi_stream_destroy": (None, [_vp]),
    "dsmi_stream_last_error": (C.c_char_p, [_vp]),
    "dsmi_stream_reset": (C.c_int, [_vp]),
    "dsmi_stream_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "dsmi_segment": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, _vp, _vp, C.c_int,
                               C.POINTER(C.c_int), _vp, _vp]),
    "dsmi_decoder_create": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(_vp)]),
    "dsmi_decoder_destroy": (None, [_vp]),
    "dsmi_decoder_last_error": (C.c_char_p, [_vp]),
    "dsmi_decoder_set_lm": (C.c_int, [_vp, C.c_char_p, C.c_double, C.c_double]),
    "dsmi_beam": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "dsmi_forward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "dsmi_forward_ready": (C.c_int, [_vp]),
    "dsmi_forward_status": (C.c_int, [_vp]),
    "dsmi_recompute_count": (C.c_int, [_vp]),
    "dsmi_model_set_inflight": (C.c_int, [_vp, C.c_int]),
    "dsmi_model_set_ring_windows": (C.c_int, [_vp, C.c_int]),
    "dsmi_pack_pcm_i16": (C.c_int, [_vp, C.c_int64, _vp]),
    "dsmi_upload": (C.c_int, [C.c_int, _vp, _vp, C.c_int64, _vp]),
    "dsmi_conv_stack": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "dsmi_rnn_layer": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "dsmi_greedy": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "dsmi_greedy_enqueue": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "dsmi_greedy_collect": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dsmi_set_profiling": (C.c_int, [_vp, C.c_int]),
    "dsmi_stage_time_us": (C.c_double, [_vp, C.c_int]),
    "dsmi_kernel_stats": (C.c_int, [_vp, C.c_int, _i64p, _i64p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dsmi_reset_kernel_stats": (C.c_int, [_vp]),
    "dsmi_debug_persist_stamps": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64]),
    "dsmi_debug_step_stamps": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64]),
    "dsmi_last_forward_stats": (C.c_int, [_vp, _i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dsmi_lm_open": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "dsmi_lm_close": (None, [_vp]),
    "dsmi_lm_last_error": (C.c_char_p, [_vp]),
    "dsmi_lm_info": (C.c_int, [_vp, C.POINTER(C.c_int), _i64p, C.POINTER(C.c_int)]),
    "dsmi_lm_word_index": (C.c_int, [_vp, C.c_char_p]),
    "dsmi_lm_lookup": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "dsmi_lm_cond_log10": (C.c_double, [_vp, _vp, C.c_int]),
    "dsmi_beam_enqueue": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp]),
    "dsmi_beam_collect": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "dsmi_decoder_beam_stats": (C.c_int, [_vp, _vp]),
    "dsmi_debug_beam_stamps": (C.c_int, [_vp, _vp, C.c_int64]),
    "dsmi_model_info": (C.c_int, [_vp, C.POINTER(ModelDesc), C.POINTER(C.c_int)]),
    "dsmi_frontend_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dsmi_decoder_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dsmi_decoder_label": (C.c_char_p, [_vp, C.c_int]),
    "dsmi_session_create": (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp)]),
    "dsmi_session_destroy": (None, [_vp]),
    "dsmi_session_last_error": (C.c_char_p, [_vp]),
    "dsmi_recognize_batch": (C.c_int, [_vp, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, _vp]),
    "dsmi_recognize_enqueue": (C.c_int, [_vp, C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "dsmi_recognize_enqueue_device": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "dsmi_recognize_collect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp, _vp]),
    "dsmi_plan_shards": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "dsmi_comm_unique_id": (C.c_int, [_vp]),
    "dsmi_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "dsmi_comm_destroy": (None, [_vp]),
    "dsmi_comm_last_error": (C.c_char_p, [_vp]),
    "dsmi_comm_scatter": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _vp, C.c_int, C.c_int, C.POINTER(_vp), _vp, _vp, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
    "dsmi_comm_gather_text": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
}


def declared_symbols():
    """Every function include/dsmi.h declares (used by the CPU-side export test)."""
    return sorted(_PROTOS)


_hw_queue_note = [False]


def want_hw_queues(n=8):
    """The batch pipeline keeps four forwards in flight on four streams beside a decode stream; the ROCm runtime maps a process's
    streams onto FOUR hardware queues by default, and two streams that share one run one after the other (a forward's persistent
    recurrent kernel behind another forward's dense kernels: 9.7 against 5.9 ms per batch, tools/exp/pipeline_lanes.py).  The
    runtime reads GPU_MAX_HW_QUEUES once, at the process's first GPU call: an engine asks for ``n`` when it is made
    (``DanSpeechRecognizer.__init__``), which takes effect if nothing has touched the GPU yet; a value the caller has set is kept;
    otherwise one warning says what the pipeline will cost."""
    if os.environ.get("GPU_MAX_HW_QUEUES"):
        return
    touched = _lib is not None and _handles[0] > 0
    try:
        import torch
        touched = touched or torch.cuda.is_initialized()
    except ImportError:
        pass
    if not touched:
        os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    elif not _hw_queue_note[0]:
        _hw_queue_note[0] = True
        import warnings
        warnings.warn("danspeech_amd: the GPU runtime was initialised before the recognizer was made and GPU_MAX_HW_QUEUES is not set: "
                      "the batch pipeline's streams will share four hardware queues (recognize_batches runs up to 1.6x slower). "
                      "Set GPU_MAX_HW_QUEUES=8 in the environment, or create the Recognizer before the first GPU call.", RuntimeWarning)


_handles = [0]          # native handles made so far (any of them has initialised the GPU runtime)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `make -C danspeech_amd/csrc` (hipcc, gfx950) or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _np_ptr(a):
    return a.ctypes.data_as(_vp)


class NativeModel:
    """Owns one dsmi_model handle (one GPU)."""

    def __init__(self, cfg, state_dict, device=0, audio_conf=None, n_labels=33):
        L = lib()
        ac = audio_conf or {}
        d = ModelDesc()
        d.conv_layers = int(cfg["conv_layers"])
        d.rnn_type = RNN_TYPES[cfg["rnn_type"]]
        d.rnn_hidden_size = int(cfg["rnn_hidden_size"])
        d.rnn_layers = int(cfg["rnn_layers"])
        d.bidirectional = int(bool(cfg["bidirectional"]))
        d.context = int(cfg.get("context", 20))
        d.n_labels = int(n_labels)
        d.sample_rate = int(ac.get("sampling_rate", 16000))
        d.window_size = float(ac.get("window_size", 0.02))
        self.desc = d
        self.n_labels = int(n_labels)
        self.device = device
        h = _vp()
        _handles[0] += 1
        rc = L.dsmi_model_create(C.byref(d), device, C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (L.dsmi_last_error(None) or b"").decode())
        self._h = h
        for name, t in state_dict.items():
            a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
            if a.dtype == np.int64:   # num_batches_tracked
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(L.dsmi_model_load_tensor(self._h, name.encode(), _np_ptr(a), shape, a.ndim))
        self._check(L.dsmi_model_finalize(self._h))

    def _check(self, rc):
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().dsmi_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host arithmetic
    def seq_lens(self, lens):
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        out = np.empty_like(lens)
        self._check(lib().dsmi_seq_lens(self._h, _np_ptr(lens), len(lens), _np_ptr(out)))
        return out

    def reserve(self, max_batch, max_frames):
        self._check(lib().dsmi_reserve(self._h, int(max_batch), int(max_frames)))

    # ---- device entry points (torch tensors only as containers)
    def _stream(self):
        return _stream(self.device)

    def _on_device(self, t):
        assert t.is_cuda and t.device.index == self.device, "tensor on %s, handle on cuda:%d" % (t.device, self.device)

    def forward(self, feat, lens, out=None, check=True):
        """feat: CUDA float32 [B,1,F,T] contiguous; lens sorted descending. -> (probs [B,T',C], out_lens).

        The kernels are enqueued asynchronously.  With ``check=True`` (default) the call then waits for them and
        collects the forward's status (``dsmi_forward_status``: a batch whose persistent recurrent kernel timed
        out is recomputed before this returns), so the probabilities are valid for any consumer.  A pipelining
        caller passes ``check=False`` and calls ``status()`` itself before consuming ``probs``."""
        import torch
        assert feat.dtype == torch.float32 and feat.is_contiguous()
        self._on_device(feat)
        B, T = feat.shape[0], feat.shape[-1]
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        To = int(self.seq_lens(np.array([T], dtype=np.int32))[0])
        probs = out if out is not None else torch.empty((B, To, self.n_labels), dtype=torch.float32, device=feat.device)
        out_lens = np.empty(B, dtype=np.int32)
        self._check(lib().dsmi_forward(self._h, feat.data_ptr(), _np_ptr(lens), B, T, probs.data_ptr(),
                                       _np_ptr(out_lens), self._stream()))
        if not hasattr(self, "_inflight"):
            self._inflight = []
        self._inflight.append((feat, probs))      # dsmi_forward_status may recompute from / into these buffers
        del self._inflight[:-4]
        if check:
            self.status()
        return probs, out_lens

    def status(self):
        """Wait for the oldest forward whose status has not been collected and collect it.  True when the batch had
        to be recomputed on the per-step path (results are valid either way; raises if the recompute failed)."""
        rc = lib().dsmi_forward_status(self._h)
        if getattr(self, "_inflight", None):
            self._inflight.pop(0)
        if rc == DSMI_RECOMPUTED:
            import warnings
            warnings.warn((lib().dsmi_last_error(self._h) or b"").decode(), RuntimeWarning)
            return True
        self._check(rc)
        return False

    def ready(self):
        """Has the oldest uncollected forward finished on the device (``status()`` would not block)?  Never blocks."""
        rc = lib().dsmi_forward_ready(self._h)
        if rc < 0:
            self._check(rc)
        return rc == 1

    def recompute_count(self):
        return int(lib().dsmi_recompute_count(self._h))

    def set_inflight(self, batches):
        """How many batches the caller keeps in flight on this device (one handle + stream each): 2 selects the
        throughput variant of the recurrent kernel."""
        self._check(lib().dsmi_model_set_inflight(self._h, int(batches)))

    def set_ring_windows(self, windows):
        """Ring windows the next forwards' recurrent layers take side by side (0: what ``set_inflight`` implies)."""
        self._check(lib().dsmi_model_set_ring_windows(self._h, int(windows)))

    def conv_stack(self, feat, lens):
        import torch
        B, T = feat.shape[0], feat.shape[-1]
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        To = int(self.seq_lens(np.array([T], dtype=np.int32))[0])
        from .synthetic import CONV_SPECS, conv_out_freq
        cl = self.desc.conv_layers
        n_freq = int(self.desc.sample_rate * self.desc.window_size) // 2 + 1
        C_ = CONV_SPECS[cl - 1][1]
        F_ = conv_out_freq(n_freq, cl)
        out = torch.empty((B, C_, F_, To), dtype=torch.float32, device=feat.device)
        self._check(lib().dsmi_conv_stack(self._h, feat.data_ptr(), _np_ptr(lens), B, T, out.data_ptr(), self._stream()))
        return out

    def rnn_layer(self, layer, x, out_lens):
        """x: CUDA [T,B,I] -> [T,B,H] (one BatchRNN of the model)."""
        import torch
        T, B = x.shape[0], x.shape[1]
        out_lens = np.ascontiguousarray(out_lens, dtype=np.int32)
        y = torch.empty((T, B, self.desc.rnn_hidden_size), dtype=torch.float32, device=x.device)
        self._check(lib().dsmi_rnn_layer(self._h, int(layer), x.data_ptr(), _np_ptr(out_lens), B, T, y.data_ptr(), self._stream()))
        return y

    KERNEL_KINDS = ["unused", "conv1", "conv2", "conv3", "gemm_l0", "gemm", "rnn_step", "head", "greedy", "beam", "rnn_layer_persistent"]

    def set_profiling(self, level):
        self._check(lib().dsmi_set_profiling(self._h, int(level)))

    def kernel_stats(self):
        """dict kind -> dict(launches, samples, avg_us, flops_per_launch, bytes_per_launch)."""
        out = {}
        for k, name in enumerate(self.KERNEL_KINDS):
            n = C.c_int64(); sm = C.c_int64(); us = C.c_double(); fl = C.c_double(); by = C.c_double()
            self._check(lib().dsmi_kernel_stats(self._h, k, C.byref(n), C.byref(sm), C.byref(us), C.byref(fl), C.byref(by)))
            if n.value:
                out[name] = dict(launches=n.value, samples=sm.value, avg_us=us.value,
                                 flops_per_launch=fl.value, bytes_per_launch=by.value)
        return out

    def reset_kernel_stats(self):
        self._check(lib().dsmi_reset_kernel_stats(self._h))

    def stage_time_us(self, stage):
        return float(lib().dsmi_stage_time_us(self._h, int(stage)))

    def last_forward_stats(self):
        n = C.c_int64(); a = C.c_double(); b = C.c_double()
        self._check(lib().dsmi_last_forward_stats(self._h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value


def _stream(device=None):
    """The torch stream current on ``device`` (the HANDLE's device, not torch's current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NativeFrontend:
    """Owns one dsmi_frontend handle: SpectrogramAudioParser on one GPU."""

    def __init__(self, audio_conf=None, device=0, pad_mode="reflect"):
        L = lib()
        ac = audio_conf or {}
        d = FrontendDesc()
        d.sample_rate = int(ac.get("sampling_rate", 16000))
        d.window_size = float(ac.get("window_size", 0.02))
        d.window_stride = float(ac.get("window_stride", 0.01))
        d.window = WINDOWS[ac.get("window", "hamming")]
        d.normalize = int(bool(ac.get("normalize", True)))
        d.pad_mode = PAD_MODES[pad_mode]
        self.desc = d
        self.device = device
        self.n_fft = int(d.sample_rate * d.window_size)
        self.hop = int(d.sample_rate * d.window_stride)
        self.n_freq = self.n_fft // 2 + 1
        h = _vp()
        _handles[0] += 1
        rc = L.dsmi_frontend_create(C.byref(d), device, C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (L.dsmi_frontend_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().dsmi_frontend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    PCM_STEREO = 16
    WAV_WIDTH_DTYPE = {1: 3, 2: 0, 3: 4, 4: 5}       # sample width in bytes -> DSMI_PCM_{U8,I16,I24,I32}

    def features(self, pcm_dev, n_samples, t_stride=None, wav_format=None, device=None):
        """pcm_dev: 1-D CUDA tensor (int16/float32/float64), clips back to back; or, with
        ``wav_format=(sample_width, channels)``, a uint8 tensor holding the raw frames of PCM WAV
        files back to back (``n_samples`` then counts frames; stereo is folded on the device).
        -> (feat [B,1,F,t_stride] float32 CUDA, frames int32[B])."""
        import torch
        n_samples = np.ascontiguousarray(n_samples, dtype=np.int64)
        B = len(n_samples)
        frames = 1 + n_samples // self.hop
        if t_stride is None:
            t_stride = int(frames.max())
        if wav_format is None:
            dt = {torch.int16: 0, torch.float32: 1, torch.float64: 2}[pcm_dev.dtype]
        else:
            width, channels = wav_format
            if pcm_dev.dtype != torch.uint8 or width not in self.WAV_WIDTH_DTYPE or channels not in (1, 2):
                raise ValueError("raw WAV frames: uint8 tensor, sample width 1..4, one or two channels")
            if int(n_samples.sum()) * width * channels != pcm_dev.numel():
                raise ValueError("frame counts do not add up to the size of the byte buffer")
            dt = self.WAV_WIDTH_DTYPE[width] | (self.PCM_STEREO if channels == 2 else 0)
        feat = torch.empty((B, 1, self.n_freq, t_stride), dtype=torch.float32, device=device if device is not None else pcm_dev.device)
        fr = np.empty(B, dtype=np.int32)
        rc = lib().dsmi_features(self._h, pcm_dev.data_ptr(), dt, _np_ptr(n_samples), B, feat.data_ptr(),
                                 int(t_stride), _np_ptr(fr), _stream(self.device))
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_frontend_last_error(self._h) or b"").decode())
        return feat, fr


def _segment(self, pcm_dev, energy_threshold=600, step=1024, pause_hops=9, phrase_hops=4, wav_format=None,
             max_segments=None, return_energies=False):
    """dsmi_segment over ONE recording resident on the GPU -> int64 [n,2] sample ranges [start, end)
    (and the float64 hop energies)."""
    import torch
    if wav_format is None:
        dt = {torch.int16: 0, torch.float32: 1, torch.float64: 2}[pcm_dev.dtype]
        n = pcm_dev.numel()
    else:
        width, channels = wav_format
        dt = self.WAV_WIDTH_DTYPE[width] | (self.PCM_STEREO if channels == 2 else 0)
        n = pcm_dev.numel() // (width * channels)
    nhops = max((n - 1) // step, 0) if n > step else 0
    cap = int(max_segments) if max_segments is not None else nhops // 2 + 1
    st = np.zeros(max(cap, 1), dtype=np.int64); en = np.zeros(max(cap, 1), dtype=np.int64)
    e = np.zeros(max(nhops, 1), dtype=np.float64)
    found = C.c_int(0)
    rc = lib().dsmi_segment(self._h, pcm_dev.data_ptr(), dt, n, int(step), float(energy_threshold), int(pause_hops),
                            int(phrase_hops), _np_ptr(st), _np_ptr(en), cap, C.byref(found), _np_ptr(e), _stream(self.device))
    if rc != 0:
        raise DsmiError(rc, (lib().dsmi_frontend_last_error(self._h) or b"").decode())
    seg = np.stack([st[:found.value], en[:found.value]], axis=1)
    return (seg, e[:nhops]) if return_energies else seg


NativeFrontend.segment = _segment


def _features_stream(self, pcm_dev, state):
    """dsmi_features_stream: one chunk of the streaming parser.  ``state`` = float64[3] (input_mean, input_std,
    alpha), updated in place.  -> feat [n_freq, frames] float32 CUDA."""
    import torch
    dt = {torch.int16: 0, torch.float32: 1, torch.float64: 2}[pcm_dev.dtype]
    n = pcm_dev.numel()
    n_fft = 2 * (self.n_freq - 1)
    nfr = 1 + (n - n_fft) // self.hop if n >= n_fft else 0
    feat = torch.empty((self.n_freq, max(nfr, 1)), dtype=torch.float32, device=pcm_dev.device)
    fr = np.zeros(1, dtype=np.int32)
    rc = lib().dsmi_features_stream(self._h, pcm_dev.data_ptr(), dt, n, _np_ptr(state), feat.data_ptr(), feat.shape[1],
                                    _np_ptr(fr), _stream(self.device))
    if rc != 0:
        raise DsmiError(rc, (lib().dsmi_frontend_last_error(self._h) or b"").decode())
    return feat[:, :int(fr[0])]


NativeFrontend.features_stream = _features_stream


class NativeStream:
    """Owns one dsmi_stream handle: the carried state of one utterance streamed through a unidirectional model."""

    def __init__(self, model):
        self.model = model
        h = _vp()
        rc = lib().dsmi_stream_create(model._h, C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_stream_last_error(None) or b"").decode())
        self._h = h

    def _check(self, rc):
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_stream_last_error(self._h) or b"").decode())

    def forward(self, feat, is_first, is_last):
        """feat: CUDA float32 [F,T] (or [1,1,F,T]) -> probs [1,T_out,C] CUDA, or None while the lookahead buffers."""
        import torch
        feat = feat.reshape(feat.shape[-2], feat.shape[-1]).contiguous()
        assert feat.is_cuda and feat.dtype == torch.float32
        T = feat.shape[1]
        cap = T + 4 * int(self.model.desc.context) + 2048 if not hasattr(self, "_cap") else self._cap
        while True:
            probs = torch.empty((cap, self.model.n_labels), dtype=torch.float32, device=feat.device)
            tout = np.zeros(1, dtype=np.int32)
            rc = lib().dsmi_stream_forward(self._h, feat.data_ptr(), T, int(bool(is_first)), int(bool(is_last)),
                                           probs.data_ptr(), cap, _np_ptr(tout), _stream(self.model.device))
            if rc == DSMI_ERR_CAPACITY and cap < (1 << 24):
                # nothing was consumed: the capacity check precedes every state update of the lookahead
                cap *= 4
                continue
            self._check(rc)
            break
        n = int(tout[0])
        return probs[:n].unsqueeze(0) if n > 0 else None

    def reset(self):
        self._check(lib().dsmi_stream_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().dsmi_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeDecoder:
    """Owns one dsmi_decoder handle: greedy and beam-search CTC decoding on one GPU."""

    def __init__(self, labels, blank_index=0, device=0):
        L = lib()
        self.labels = labels
        self.device = device
        arr = (C.c_char_p * len(labels))(*[c.encode("utf-8") for c in labels])
        h = _vp()
        _handles[0] += 1
        rc = L.dsmi_decoder_create(device, arr, len(labels), int(blank_index), C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (L.dsmi_decoder_last_error(None) or b"").decode())
        self._h = h

    def _check(self, rc):
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_decoder_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().dsmi_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lm(self, lm_path, alpha, beta):
        self._check(lib().dsmi_decoder_set_lm(self._h, lm_path.encode() if lm_path else None, float(alpha), float(beta)))

    def greedy(self, probs, sizes=None):
        """probs: CUDA [B,T,C] -> list of (ids, offsets) int32 arrays per utterance."""
        B, T = probs.shape[0], probs.shape[1]
        ids = np.empty((B, T), dtype=np.int32)
        offs = np.empty((B, T), dtype=np.int32)
        n = np.empty(B, dtype=np.int32)
        sz = None if sizes is None else np.ascontiguousarray(sizes, dtype=np.int32)
        self._check(lib().dsmi_greedy(self._h, probs.data_ptr(), None if sz is None else _np_ptr(sz), B, T,
                                      _np_ptr(ids), _np_ptr(offs), _np_ptr(n), _stream(self.device)))
        return [(ids[b, :n[b]].copy(), offs[b, :n[b]].copy()) for b in range(B)]

    def greedy_enqueue(self, probs, sizes=None):
        """Launch the greedy decode and the copies of its results on the current stream and return at once."""
        B, T = probs.shape[0], probs.shape[1]
        sz = None if sizes is None else np.ascontiguousarray(sizes, dtype=np.int32)
        self._check(lib().dsmi_greedy_enqueue(self._h, probs.data_ptr(), None if sz is None else _np_ptr(sz), B, T, _stream(self.device)))
        self._greedy_pending = (probs, B, T)                     # keeps the probabilities alive until the collect

    def greedy_collect(self):
        _, B, T = self._greedy_pending
        self._greedy_pending = None
        ids = np.empty((B, T), dtype=np.int32)
        offs = np.empty((B, T), dtype=np.int32)
        n = np.empty(B, dtype=np.int32)
        self._check(lib().dsmi_greedy_collect(self._h, _np_ptr(ids), _np_ptr(offs), _np_ptr(n)))
        return [(ids[b, :n[b]].copy(), offs[b, :n[b]].copy()) for b in range(B)]

    def beam(self, probs, sizes=None, beam_width=64, cutoff_top_n=40, cutoff_prob=1.0):
        """probs: CUDA [B,T,C] -> (tokens [B,beam,T], timesteps [B,beam,T], lens [B,beam], scores [B,beam])."""
        self.beam_enqueue(probs, sizes, beam_width, cutoff_top_n, cutoff_prob)
        return self.beam_collect()

    def beam_enqueue(self, probs, sizes=None, beam_width=64, cutoff_top_n=40, cutoff_prob=1.0):
        """Launch the search on the current stream and return at once; ``beam_collect`` waits and returns the arrays."""
        B, T = probs.shape[0], probs.shape[1]
        sz = None if sizes is None else np.ascontiguousarray(sizes, dtype=np.int32)
        self._check(lib().dsmi_beam_enqueue(self._h, probs.data_ptr(), None if sz is None else _np_ptr(sz), B, T, int(beam_width),
                                            int(cutoff_top_n), float(cutoff_prob), _stream(self.device)))
        self._beam_pending = (probs, B, T, int(beam_width))        # keeps the probabilities alive until the collect

    def beam_collect(self):
        _, B, T, beam_width = self._beam_pending
        self._beam_pending = None
        tok = np.zeros((B, beam_width, T), dtype=np.int32)
        ts = np.zeros((B, beam_width, T), dtype=np.int32)
        ln = np.zeros((B, beam_width), dtype=np.int32)
        sc = np.zeros((B, beam_width), dtype=np.float32)
        self._check(lib().dsmi_beam_collect(self._h, _np_ptr(tok), _np_ptr(ts), _np_ptr(ln), _np_ptr(sc)))
        return tok, ts, ln, sc

    def beam_stamps(self):
        """[64, 8] uint64: 100 MHz phase-boundary stamps of the last collected search (dsmi_debug_beam_stamps)."""
        st = np.zeros((64, 8), dtype=np.uint64)
        self._check(lib().dsmi_debug_beam_stamps(self._h, _np_ptr(st), st.size))
        return st

    def beam_stats(self):
        """Of the last collected search: {revivals, walk_hops, list_rankings, full_rankings} (dsmi_decoder_beam_stats)."""
        c = np.zeros(4, dtype=np.int32)
        self._check(lib().dsmi_decoder_beam_stats(self._h, _np_ptr(c)))
        return dict(zip(("revivals", "walk_hops", "list_rankings", "full_rankings"), (int(v) for v in c)))


class NativeLM:
    """Host-only view of a language-model file through libdsmi.so's reader (``dsmi_lm_*``): no GPU needed."""
    KINDS = {0: "arpa", 1: "klm-probing", 2: "klm-trie"}

    def __init__(self, path):
        h = _vp()
        rc = lib().dsmi_lm_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_lm_last_error(None) or b"").decode())
        self._h = h
        o = C.c_int(); v = C.c_int64(); k = C.c_int()
        lib().dsmi_lm_info(self._h, C.byref(o), C.byref(v), C.byref(k))
        self.order, self.vocab_size, self.kind = o.value, v.value, self.KINDS[k.value]

    def close(self):
        if getattr(self, "_h", None):
            lib().dsmi_lm_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def word_index(self, word):
        return int(lib().dsmi_lm_word_index(self._h, word.encode("utf-8")))

    def lookup(self, ids):
        """(log10 prob, log10 backoff) of the n-gram, or None."""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        lp = C.c_float(); bo = C.c_float()
        rc = lib().dsmi_lm_lookup(self._h, _np_ptr(a), len(a), C.byref(lp), C.byref(bo))
        if rc < 0:
            raise DsmiError(rc, "bad n-gram")
        return (lp.value, bo.value) if rc == 1 else None

    def cond_log10(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        return float(lib().dsmi_lm_cond_log10(self._h, _np_ptr(a), len(a)))


_PCM_CODE = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def plan_shards(n_samples, world):
    """dsmi_plan_shards: (rank_of, slot_of) int32 arrays."""
    n = np.ascontiguousarray(n_samples, dtype=np.int64)
    rank_of = np.zeros(len(n), dtype=np.int32)
    slot_of = np.zeros(len(n), dtype=np.int32)
    rc = lib().dsmi_plan_shards(_np_ptr(n) if len(n) else None, len(n), int(world),
                                _np_ptr(rank_of) if len(n) else None, _np_ptr(slot_of) if len(n) else None)
    if rc != 0:
        raise DsmiError(rc, "dsmi_plan_shards")
    return rank_of, slot_of


class NativeSession:
    """dsmi_session_*: the fused recognise call a host without the Python layer uses (tests drive it through ctypes).
    ``frontend`` / ``model`` / ``decoder``: NativeFrontend / NativeModel / NativeDecoder; they must outlive the session."""

    def __init__(self, frontend, model, decoder):
        self._keep = (frontend, model, decoder)
        h = _vp()
        rc = lib().dsmi_session_create(frontend._h, model._h, decoder._h, C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_session_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if self._h:
            lib().dsmi_session_destroy(self._h)
            self._h = None

    def _err(self, rc):
        return DsmiError(rc, (lib().dsmi_session_last_error(self._h) or b"").decode())

    def enqueue(self, clips):
        clips = [np.ascontiguousarray(c) for c in clips]
        kinds = {c.dtype for c in clips}
        if len(kinds) != 1 or next(iter(kinds)) not in _PCM_CODE:
            raise ValueError("clips of one sample type: int16, float32 or float64")
        n = np.array([len(c) for c in clips], dtype=np.int64)
        ptrs = (_vp * len(clips))(*[c.ctypes.data for c in clips])
        rc = lib().dsmi_recognize_enqueue(self._h, ptrs, _np_ptr(n), _PCM_CODE[clips[0].dtype], len(clips))
        if rc != 0:
            raise self._err(rc)
        self._count = len(clips)

    def enqueue_device(self, pcm_dev, n_samples, dtype_code):
        n = np.ascontiguousarray(n_samples, dtype=np.int64)
        ptr = pcm_dev.data_ptr() if hasattr(pcm_dev, "data_ptr") else int(pcm_dev)
        rc = lib().dsmi_recognize_enqueue_device(self._h, ptr, _np_ptr(n), int(dtype_code), len(n))
        if rc != 0:
            raise self._err(rc)
        self._count = len(n)
        self._keep_pcm = pcm_dev

    def collect(self, beam_width=0, cutoff_top_n=40, cutoff_prob=1.0, text_stride=4096, raw=False):
        B = self._count
        text = np.zeros((B, text_stride), dtype=np.uint8)
        nbytes = np.zeros(B, dtype=np.int32)
        scores = np.zeros(B, dtype=np.float32)
        rc = lib().dsmi_recognize_collect(self._h, int(beam_width), int(cutoff_top_n), float(cutoff_prob), text.ctypes.data, text_stride,
                                          _np_ptr(nbytes), _np_ptr(scores))
        if rc < 0:
            raise self._err(rc)
        self.last_status = rc
        if raw:
            return text, nbytes, scores
        return [bytes(text[b]).split(b"\0", 1)[0].decode("utf-8") for b in range(B)], nbytes, scores

    def recognize_batch(self, clips, **kw):
        self.enqueue(clips)
        return self.collect(**kw)


class NativeComm:
    """dsmi_comm_*: scatter / gather over RCCL for hosts without torch.distributed."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        rc = lib().dsmi_comm_unique_id(buf)
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_comm_last_error(None) or b"").decode())
        return bytes(buf)

    def __init__(self, unique_id, rank, world, device):
        h = _vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        rc = lib().dsmi_comm_init(buf, int(rank), int(world), int(device), C.byref(h))
        if rc != 0:
            raise DsmiError(rc, (lib().dsmi_comm_last_error(None) or b"").decode())
        self._h, self.rank, self.world, self.device = h, rank, world, device

    def close(self):
        if self._h:
            lib().dsmi_comm_destroy(self._h)
            self._h = None

    def _err(self, rc):
        return DsmiError(rc, (lib().dsmi_comm_last_error(self._h) or b"").decode())

    def scatter(self, clips, root=0, cap=4096):
        """-> (device address of the shard, n_samples int64[count], positions int32[count], sample-type code, total)."""
        ptrs, n, code, count = None, None, 0, 0
        if self.rank == root:
            clips = [np.ascontiguousarray(c) for c in clips]
            code = _PCM_CODE[clips[0].dtype] if clips else 0
            n = np.array([len(c) for c in clips], dtype=np.int64)
            ptrs = (_vp * max(len(clips), 1))(*[c.ctypes.data for c in clips])
            count = len(clips)
        dev = _vp()
        sn = np.zeros(cap, dtype=np.int64)
        si = np.zeros(cap, dtype=np.int32)
        cnt, dt, tot = C.c_int(), C.c_int(), C.c_int()
        rc = lib().dsmi_comm_scatter(self._h, root, ptrs, _np_ptr(n) if n is not None and len(n) else None, code, count, C.byref(dev),
                                     _np_ptr(sn), _np_ptr(si), cap, C.byref(cnt), C.byref(dt), C.byref(tot), _stream(self.device))
        if rc != 0:
            raise self._err(rc)
        return dev.value, sn[:cnt.value].copy(), si[:cnt.value].copy(), dt.value, tot.value

    def gather_text(self, text, positions, total, root=0):
        """text: uint8 [count][stride] (NativeSession.collect(raw=True)); -> list of str on the root, None elsewhere."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        stride = text.shape[1]
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        out = np.zeros((max(total, 1), stride), dtype=np.uint8)
        rc = lib().dsmi_comm_gather_text(self._h, root, text.ctypes.data if len(pos) else None, stride, _np_ptr(pos) if len(pos) else None,
                                         len(pos), total, out.ctypes.data, _stream(self.device))
        if rc != 0:
            raise self._err(rc)
        if self.rank != root:
            return None
        return [bytes(out[i]).split(b"\0", 1)[0].decode("utf-8") for i in range(total)]
