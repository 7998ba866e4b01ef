"""``DeepSpeech``: the acoustic-model object of the drop-in surface.

Mirrors the contract the reference's engine relies on (reference
danspeech/deepspeech/model.py:287-666, used at danspeech/DanSpeechRecognizer.py:48-56,218-224):
attributes ``audio_conf, labels, model_name, context`` (+ the package fields), ``.to(device)``,
``.eval()``, and ``model(x[B,1,F,T], lengths[B]) -> (probs[B,T',C], output_lengths[B])`` with
eval-mode softmax probabilities.  The arithmetic runs in libdsmi.so (HIP, gfx950) through
``danspeech_amd._native.NativeModel``; this class only holds the state dict and the handle.
There is no CPU path: ``.to('cpu')`` keeps the weights on the host and any forward call then
fails loudly.
"""
import math
from collections import OrderedDict

import numpy as np

from ..errors.model_errors import ConvError
from .utils import get_default_audio_config, DANSPEECH_LABELS

# reference model.py:14-19 maps names to torch.nn classes; here names map to kernel kinds.
supported_rnns = {"lstm": "lstm", "rnn": "rnn", "gru": "gru"}
supported_rnns_inv = dict((v, k) for k, v in supported_rnns.items())


def _rnn_kind(rnn_type):
    """Accept 'gru'/'lstm'/'rnn' or the torch.nn.GRU/LSTM/RNN classes the reference passes."""
    if isinstance(rnn_type, str):
        k = rnn_type.lower()
    else:
        k = getattr(rnn_type, "__name__", str(rnn_type)).lower()
    if k not in supported_rnns:
        raise ValueError("unsupported rnn_type %r (supported: gru, lstm, rnn)" % (rnn_type,))
    return k


class DeepSpeech(object):
    def __init__(self, model_name, rnn_type="gru", labels=None, rnn_hidden_size=768, rnn_layers=5, audio_conf=None,
                 bidirectional=True, context=20, conv_layers=2, streaming_inference_model=False):
        if not labels:                      # model.py:318-322: default DanSpeech labels
            labels = DANSPEECH_LABELS
        if audio_conf is None:              # model.py:325-326
            audio_conf = get_default_audio_config()
        self.model_name = model_name
        self.rnn_hidden_size = rnn_hidden_size
        self.rnn_layers = rnn_layers
        self.rnn_type = _rnn_kind(rnn_type)
        self.audio_conf = audio_conf or {}
        self.labels = labels
        self.bidirectional = bidirectional
        self.conv_layers = conv_layers
        self.streaming_model = streaming_inference_model
        self.context = context
        if conv_layers == 0:                # model.py:344-348
            raise ConvError("0 convolutional layers configuration not supported by DanSpeech")
        if conv_layers > 3:
            raise ConvError("Maximum amount of convolutional layers supported by DanSpeech is 3")
        if self.streaming_model:
            # streaming_init (model.py:427-494) always builds unidirectional layers, and only its 2-conv shape
            # can run: the first RNN layer is sized for two conv layers whatever conv_layers says (:476-484)
            # and the 1-conv branch builds a plain MaskConv that streaming_forward cannot call
            if conv_layers != 2:
                raise ConvError("streaming models exist only with 2 convolutional layers (reference streaming_init)")
            self.bidirectional = False
            self.forward = self.streaming_forward      # model.py:424-425
        self._state = None
        self._native = None
        self._stream = None
        self.device = "cpu"
        self.training = False

    # ---- parameters -------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        sd = OrderedDict()
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            sd[k] = a
        self._state = sd
        if self._stream is not None:
            self._stream.close()
            self._stream = None
        if self._native is not None:
            self._native.close()
            self._native = None
        return self

    def state_dict(self):
        return self._state

    def _cfg(self):
        return dict(conv_layers=self.conv_layers, rnn_type=self.rnn_type, rnn_hidden_size=self.rnn_hidden_size,
                    rnn_layers=self.rnn_layers, bidirectional=self.bidirectional, context=self.context)

    # ---- torch.nn.Module look-alikes the engine calls ------------------------------------------
    def to(self, device):
        import torch
        dev = torch.device(device)
        if dev.type == "cuda":
            if self._state is None:
                raise RuntimeError("DeepSpeech has no weights: call load_state_dict()/load_model() first")
            index = dev.index if dev.index is not None else torch.cuda.current_device()
            if self._native is None or self._native.device != index:
                from .. import _native
                if self._stream is not None:
                    self._stream.close()
                    self._stream = None
                if self._native is not None:
                    self._native.close()
                self._native = _native.NativeModel(self._cfg(), self._state, device=index,
                                                   audio_conf=self.audio_conf, n_labels=len(self.labels))
            self.device = "cuda:%d" % index
        else:
            self.device = "cpu"
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else "cuda:%d" % device)

    def eval(self):
        self.training = False
        return self

    def get_seq_lens(self, input_length):
        """model.py:540-551."""
        import torch
        L = torch.as_tensor(input_length).clone().long()
        from ..synthetic import CONV_SPECS
        for (_, _, _, kt, _, st, _, pt) in CONV_SPECS[:self.conv_layers]:
            L = (L + 2 * pt - (kt - 1) - 1) // st + 1
        return L.int()

    def forward(self, x, lengths):
        """model.py:496-515. x: [B,1,F,T] float tensor, lengths: [B] sorted descending."""
        import torch
        if self._native is None:
            raise RuntimeError("this DeepSpeech runs only on an MI355X: call model.to('cuda') first (no CPU path)")
        lengths = torch.as_tensor(lengths).cpu().int()
        dev = torch.device(self.device)
        x = torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()
        probs, out_lens = self._native.forward(x, lengths.numpy())
        return probs, torch.from_numpy(out_lens.copy()).int()

    def enqueue(self, x, lengths):
        """``forward`` without waiting for the GPU: the kernels are enqueued on the current stream and the
        probabilities may be consumed only after ``collect()`` (once per ``enqueue``, in order).  For callers that
        keep several batches in flight (``DanSpeechRecognizer.transcribe_batches``)."""
        import torch
        if self._native is None:
            raise RuntimeError("this DeepSpeech runs only on an MI355X: call model.to('cuda') first (no CPU path)")
        lengths = torch.as_tensor(lengths).cpu().int()
        x = torch.as_tensor(x, dtype=torch.float32).to(torch.device(self.device)).contiguous()
        probs, out_lens = self._native.forward(x, lengths.numpy(), check=False)
        return probs, torch.from_numpy(out_lens.copy()).int()

    def collect(self):
        """Wait for the oldest ``enqueue`` and validate it (``dsmi_forward_status``)."""
        return self._native.status()

    def ready(self):
        """Would ``collect()`` return without waiting?"""
        return self._native.ready()

    def set_inflight(self, batches):
        """Tell the kernels how many batches the caller keeps in flight on this device (``dsmi_model_set_inflight``)."""
        if self._native is not None:
            self._native.set_inflight(batches)

    def set_ring_windows(self, windows):
        """``dsmi_model_set_ring_windows``: 2 when only two forwards will share the chip (0: by ``set_inflight``)."""
        if self._native is not None:
            self._native.set_ring_windows(windows)

    def replica(self):
        """A second handle on the same weights and device (own workspaces): a caller that keeps two batches in flight
        alternates between the model and its replica, so that the recurrent layers of the two batches share the CUs
        (half-CU persistent workgroups, one gate lane per handle: csrc/rnn_persist16.hip, csrc/api.hip)."""
        other = DeepSpeech(self.model_name, rnn_type=self.rnn_type, labels=self.labels, rnn_hidden_size=self.rnn_hidden_size,
                           rnn_layers=self.rnn_layers, audio_conf=self.audio_conf, bidirectional=self.bidirectional,
                           context=self.context, conv_layers=self.conv_layers)
        other._state = self._state
        return other.to(self.device)

    def streaming_forward(self, x, is_first, is_last):
        """model.py:517-537: one chunk [1,1,F,T] of the streaming parser's output -> probs [1,T_out,C], or None
        on the first pass (the lookahead is still buffering).  Conv context, recurrent state and lookahead
        buffer live in a ``dsmi_stream`` on the GPU between calls; ``is_last`` clears them."""
        import torch
        if self._native is None:
            raise RuntimeError("this DeepSpeech runs only on an MI355X: call model.to('cuda') first (no CPU path)")
        if self._stream is None:
            from .. import _native
            self._stream = _native.NativeStream(self._native)
        x = torch.as_tensor(x, dtype=torch.float32).to(torch.device(self.device))
        if x.dim() == 4 and x.shape[0] != 1:
            raise ValueError("streaming handles a single sequence (MaskConvStream, model.py:159)")
        return self._stream.forward(x, is_first, is_last)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def freeze_layers(self, number_to_freeze=0):
        raise NotImplementedError("training helpers live in the separate danspeech_training repository (reference README.md:19-21)")

    # ---- packages (reference model.py:599-650) --------------------------------------------------
    @classmethod
    def load_model(cls, path):
        import pickle
        import torch
        try:
            # plain dicts / lists / tensors (what the reference's own packages hold, model.py:607-619): nothing to unpickle
            package = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            # The reference loads with a full unpickle (model.py:607), which runs whatever code the file names.  A package that
            # needs that is not loaded silently: the caller decides, and hands the unpickled dict to load_model_package().
            raise RuntimeError(
                "%s holds objects beyond tensors, dicts, lists and scalars (%s). danspeech_amd does not unpickle arbitrary objects "
                "from a model file; if you trust this file, load it yourself -- package = torch.load(path, map_location='cpu', "
                "weights_only=False) -- and pass the dict to DeepSpeech.load_model_package(package)" % (path, str(e).splitlines()[0])) from e
        return cls.load_model_package(package)

    @classmethod
    def load_model_package(cls, package):
        model = cls(model_name=package["model_name"],
                    rnn_hidden_size=package["rnn_hidden_size"],
                    rnn_layers=package["rnn_layers"],
                    labels=package["labels"],
                    audio_conf=package["audio_conf"],
                    rnn_type=supported_rnns[package["rnn_type"]],
                    bidirectional=package["bidirectional"],
                    conv_layers=package["conv_layers"],
                    context=package["context"],
                    streaming_inference_model=package.get("streaming_model", False))
        model.load_state_dict(package["state_dict"])
        return model

    def serialize(self):
        """The ``.pth`` package layout load_model expects (model.py:607-619)."""
        import torch
        return {"model_name": self.model_name, "rnn_hidden_size": self.rnn_hidden_size, "rnn_layers": self.rnn_layers,
                "labels": self.labels, "audio_conf": self.audio_conf, "rnn_type": self.rnn_type,
                "bidirectional": self.bidirectional, "conv_layers": self.conv_layers, "context": self.context,
                "streaming_model": self.streaming_model,
                "state_dict": OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in self._state.items())}

    @staticmethod
    def get_param_size(model):
        """model.py:652-666: number of trainable parameters (BatchNorm running stats are buffers)."""
        params = 0
        for k, v in model.state_dict().items():
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                continue
            params += int(np.prod(np.asarray(v).shape)) if np.asarray(v).shape else 1
        return params
