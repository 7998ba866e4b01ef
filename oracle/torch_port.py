"""torch-CPU restatement of ``DeepSpeech.forward`` (eval mode, non-streaming).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The same algorithm as ``oracle/model.py``
(which spells every contraction out in numpy), built from the third-party operators the
reference's CPU path itself runs on -- ``F.conv2d``, ``F.batch_norm``, ``F.hardtanh``,
``pack_padded_sequence`` + ``torch.nn.{GRU,LSTM,RNN}``, ``F.linear``, ``F.softmax`` -- so that

* the big workloads (cfgA at B = 32, cfgB at B = 64, 30 s clips) have an oracle that finishes
  in seconds, and
* ``bench.py``'s ``cpu_baseline`` is timed on what the reference's CPU path costs (SURVEY 8(d)(ii):
  oneDNN convolutions, ``aten::gru`` with its per-step ``addmm``), with all cores.

It contains no reference code: the module tree is flattened into functional calls on the state
dict, following reference danspeech/deepspeech/model.py by line:

* ``get_seq_lens``        model.py:540-551
* ``MaskConv.forward``    model.py:65-81 (mask after every module)
* conv stacks             model.py:357-396
* reshape T x N x H       model.py:501-503
* ``BatchRNN.forward``    model.py:114-122
* ``Lookahead``           model.py:125-148, 407-411
* FC head + eval softmax  model.py:414-420, 84-93, 511-514

PINNED: tests/test_oracle_golden.py holds it to the golden vectors produced by the reference
itself (g2 conv stack, g3 BatchRNN, g4 forward small and full size).
"""
import numpy as np

from danspeech_amd.synthetic import CONV_SPECS

from .model import get_seq_lens  # integer arithmetic, shared


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _mask(x, lens):
    for i, L in enumerate(lens):
        if x.shape[3] > int(L):
            x[i, :, :, int(L):] = 0
    return x


def conv_stack(sd, x, out_lens, conv_layers):
    """x: torch [B,1,F,T] -> [B,C,F',T'] (model.py:65-81, 357-396)."""
    import torch.nn.functional as F
    for li, (_, _, _, _, sf, st, pf, pt) in enumerate(CONV_SPECS[:conv_layers]):
        p = "conv.seq_module.%d" % (3 * li)
        q = "conv.seq_module.%d" % (3 * li + 1)
        x = _mask(F.conv2d(x, _t(sd[p + ".weight"]), _t(sd[p + ".bias"]), stride=(sf, st), padding=(pf, pt)), out_lens)
        x = _mask(F.batch_norm(x, _t(sd[q + ".running_mean"]), _t(sd[q + ".running_var"]), _t(sd[q + ".weight"]),
                               _t(sd[q + ".bias"]), training=False, eps=1e-5), out_lens)
        x = _mask(F.hardtanh(x, 0.0, 20.0), out_lens)
    return x


def _rnn_module(sd, l, kind, I, H, bidirectional):
    import torch
    cls = {"gru": torch.nn.GRU, "lstm": torch.nn.LSTM, "rnn": torch.nn.RNN}[kind]
    rnn = cls(input_size=I, hidden_size=H, bidirectional=bidirectional, bias=True)
    with torch.no_grad():
        for name, prm in rnn.named_parameters():
            prm.copy_(_t(sd["rnns.%d.rnn.%s" % (l, name)]))
    rnn.eval()
    return rnn


def batch_rnn(sd, l, kind, x, lens, bidirectional, batch_norm):
    """x: torch [T,B,I] -> [T,B,H] (model.py:114-122)."""
    import torch
    import torch.nn.functional as F
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    T, B, I = x.shape
    if batch_norm:
        q = "rnns.%d.batch_norm.module" % l
        x = F.batch_norm(x.reshape(T * B, I), _t(sd[q + ".running_mean"]), _t(sd[q + ".running_var"]),
                         _t(sd[q + ".weight"]), _t(sd[q + ".bias"]), training=False, eps=1e-5).reshape(T, B, I)
    H = sd["rnns.%d.rnn.weight_hh_l0" % l].shape[1]
    rnn = _rnn_module(sd, l, kind, I, H, bidirectional)
    with torch.no_grad():
        y, _ = rnn(pack_padded_sequence(x, torch.as_tensor(np.asarray(lens), dtype=torch.int64)))
        y, _ = pad_packed_sequence(y, total_length=T)
    if bidirectional:
        y = y.view(T, B, 2, H).sum(2)
    return y


def forward(sd, cfg, x, lengths, threads=None):
    """DeepSpeech.forward (model.py:496-515) -> (probs [B,T',C] float32 numpy, out_lens int32[B])."""
    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(int(threads))
    lengths = np.asarray(lengths)
    if np.any(np.diff(lengths) > 0):
        raise RuntimeError("`lengths` array must be sorted in decreasing order")
    out_lens = get_seq_lens(lengths, cfg["conv_layers"])
    with torch.no_grad():
        y = conv_stack(sd, _t(x), out_lens, cfg["conv_layers"])
        B, C, Fq, T = y.shape
        y = y.reshape(B, C * Fq, T).permute(2, 0, 1).contiguous()
        for l in range(cfg["rnn_layers"]):
            y = batch_rnn(sd, l, cfg["rnn_type"], y, out_lens, cfg["bidirectional"], batch_norm=(l > 0))
        if not cfg["bidirectional"]:
            w = _t(sd["lookahead.0.conv.weight"])             # [H,1,context], depthwise over time
            ctx = w.shape[2]
            z = F.pad(y.permute(1, 2, 0), (0, ctx - 1))       # [B,H,T + ctx - 1]
            y = F.hardtanh(F.conv1d(z, w, groups=w.shape[0]).permute(2, 0, 1), 0.0, 20.0)
        q = "fc.0.module.0"
        Tn, Bn, H = y.shape
        y = F.batch_norm(y.reshape(Tn * Bn, H), _t(sd[q + ".running_mean"]), _t(sd[q + ".running_var"]),
                         _t(sd[q + ".weight"]), _t(sd[q + ".bias"]), training=False, eps=1e-5)
        logits = F.linear(y, _t(sd["fc.0.module.1.weight"])).reshape(Tn, Bn, -1).transpose(0, 1)
        probs = F.softmax(logits, dim=-1)
    return probs.contiguous().numpy(), out_lens


def spectrogram_batch(clips, pad_mode="reflect"):
    """The batch's features [B,1,161,Tmax] (zero past each clip's frames) + frame counts, from
    ``oracle.features.spectrogram`` per clip (parsers.py:50-72)."""
    from .features import spectrogram
    feats = [spectrogram(c, pad_mode=pad_mode) for c in clips]
    frames = np.array([f.shape[1] for f in feats], dtype=np.int32)
    x = np.zeros((len(clips), 1, feats[0].shape[0], int(frames.max())), dtype=np.float32)
    for b, f in enumerate(feats):
        x[b, 0, :, :f.shape[1]] = f
    return x, frames
