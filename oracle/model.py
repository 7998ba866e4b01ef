"""numpy fp32 restatement of ``DeepSpeech.forward`` (eval mode, non-streaming).

TEST INFRASTRUCTURE (see oracle/__init__.py). Follows, line by line:

* ``get_seq_lens``          reference danspeech/deepspeech/model.py:540-551
* ``MaskConv.forward``      model.py:65-81  (module -> zero t >= out_len after EVERY module)
* conv stacks               model.py:357-396
* reshape to T x N x H      model.py:501-503 (feature index = c * F' + f)
* ``BatchRNN.forward``      model.py:114-122 (SequenceWise BN1d, packed bidirectional RNN,
                            pad with zeros, sum the two directions)
* ``Lookahead`` + Hardtanh  model.py:125-148, 407-411, 508-509 (unidirectional only)
* FC head                   model.py:414-420, 511-512
* eval softmax              model.py:84-93, 514

Weights come in as a dict ``state_dict name -> np.ndarray`` with the reference's
names.  All arithmetic is float32 like the reference's; the summation order inside
the contractions is BLAS's, so agreement with the reference is ~1e-6, not bitwise.
"""
import numpy as np

from danspeech_amd.synthetic import CONV_SPECS, GATES

BN_EPS = np.float32(1e-5)  # torch.nn.BatchNorm{1,2}d default


def get_seq_lens(lengths, conv_layers):
    """model.py:540-551: L = (L + 2p - d(k-1) - 1)//s + 1 on the time axis per Conv2d."""
    L = np.asarray(lengths, dtype=np.int64).copy()
    for (_, _, _, kt, _, st, _, pt) in CONV_SPECS[:conv_layers]:
        L = (L + 2 * pt - (kt - 1) - 1) // st + 1
    return L.astype(np.int32)


def _bn_affine(sd, prefix):
    """Eval-mode BatchNorm as y = x * a + b with a = w / sqrt(var + eps), b = bias - mean * a."""
    a = sd[prefix + ".weight"].astype(np.float32) / np.sqrt(
        sd[prefix + ".running_var"].astype(np.float32) + BN_EPS)
    b = sd[prefix + ".bias"].astype(np.float32) - sd[prefix + ".running_mean"].astype(np.float32) * a
    return a.astype(np.float32), b.astype(np.float32)


def conv2d(x, w, bias, stride, pad):
    """Cross-correlation, NCHW, zero padding.

    Unfolds the time taps only (U[(ci,kt), r, t] = xp[ci, r, kt + st*t]) and then runs one BLAS
    matmul per kernel row kf on the input rows r = sf*f + kf, which are a contiguous slice of the
    rows with residue kf % sf -- no full im2col copy.
    """
    B, Ci, Fi, Ti = x.shape
    Co, _, kf, kt = w.shape
    sf, st = stride
    pf, pt = pad
    Fo = (Fi + 2 * pf - kf) // sf + 1
    To = (Ti + 2 * pt - kt) // st + 1
    R = Fi + 2 * pf
    out = np.empty((B, Co, Fo, To), dtype=np.float32)
    wk = [np.ascontiguousarray(w[:, :, a, :].reshape(Co, Ci * kt)) for a in range(kf)]
    for b in range(B):
        xp = np.zeros((Ci, R, Ti + 2 * pt), dtype=np.float32)
        xp[:, pf:pf + Fi, pt:pt + Ti] = x[b]
        ures = []
        for res in range(sf):
            rows = xp[:, res::sf, :]
            u = np.empty((Ci, kt, rows.shape[1], To), dtype=np.float32)
            for c in range(kt):
                u[:, c] = rows[:, :, c:c + st * (To - 1) + 1:st]
            ures.append(u.reshape(Ci * kt, rows.shape[1], To))
        acc = np.zeros((Co, Fo * To), dtype=np.float32)
        for a in range(kf):
            u = ures[a % sf][:, a // sf:a // sf + Fo, :].reshape(Ci * kt, Fo * To)
            acc += wk[a] @ u
        out[b] = acc.reshape(Co, Fo, To)
    out += bias.reshape(1, Co, 1, 1)
    return out


def conv_stack(sd, x, out_lens, conv_layers):
    """MaskConv over (Conv2d, BatchNorm2d, Hardtanh(0,20)) triples; model.py:65-81.

    The reference masks after every one of the three modules; the mask only zeroes
    and Conv2d does not see later time steps' outputs of the same layer, but BN maps
    0 -> b != 0, so the mask after BN/Hardtanh matters and is reproduced.
    """
    B = x.shape[0]
    x = x.astype(np.float32)
    for li, (_, _, _, _, sf, st, pf, pt) in enumerate(CONV_SPECS[:conv_layers]):
        w = sd["conv.seq_module.%d.weight" % (3 * li)]
        bias = sd["conv.seq_module.%d.bias" % (3 * li)]
        x = conv2d(x, w, bias, (sf, st), (pf, pt))
        _mask(x, out_lens)
        a, b = _bn_affine(sd, "conv.seq_module.%d" % (3 * li + 1))
        x = x * a.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)
        _mask(x, out_lens)
        x = np.clip(x, np.float32(0), np.float32(20))
        _mask(x, out_lens)
    return x


def _mask(x, lens):
    for i, L in enumerate(lens):
        if x.shape[3] - int(L) > 0:
            x[i, :, :, int(L):] = 0


def _sigmoid(v):
    return (np.float32(1) / (np.float32(1) + np.exp(-v))).astype(np.float32)


def rnn_direction(kind, xs, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of torch.nn.{GRU,LSTM,RNN} on a packed batch, zero initial state.

    xs: [T, B, I]; returns [T, B, H] with zeros at t >= len (pad_packed_sequence).
    Gate order is torch's: GRU [r; z; n], LSTM [i; f; g; o].
    """
    T, B, _ = xs.shape
    H = w_hh.shape[1]
    out = np.zeros((T, B, H), dtype=np.float32)
    h = np.zeros((B, H), dtype=np.float32)
    c = np.zeros((B, H), dtype=np.float32)
    lens = np.asarray(lens)
    gi_all = (xs.reshape(T * B, -1) @ w_ih.T + b_ih).reshape(T, B, -1).astype(np.float32)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        act = (t < lens)
        if not act.any():
            continue
        gi = gi_all[t]
        gh = (h @ w_hh.T + b_hh).astype(np.float32)
        if kind == "gru":
            r = _sigmoid(gi[:, :H] + gh[:, :H])
            z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]).astype(np.float32)
            hn = ((np.float32(1) - z) * n + z * h).astype(np.float32)
        elif kind == "lstm":
            g = gi + gh
            i_ = _sigmoid(g[:, :H]); f_ = _sigmoid(g[:, H:2 * H])
            g_ = np.tanh(g[:, 2 * H:3 * H]).astype(np.float32); o_ = _sigmoid(g[:, 3 * H:])
            cn = (f_ * c + i_ * g_).astype(np.float32)
            hn = (o_ * np.tanh(cn)).astype(np.float32)
            c = np.where(act[:, None], cn, c)
        else:
            hn = np.tanh(gi + gh).astype(np.float32)
        h = np.where(act[:, None], hn, h)
        out[t] = np.where(act[:, None], hn, np.float32(0))
    return out


def batch_rnn(sd, l, kind, x, lens, bidirectional, batch_norm):
    """BatchRNN.forward, model.py:114-122."""
    if batch_norm:
        a, b = _bn_affine(sd, "rnns.%d.batch_norm.module" % l)
        x = (x * a + b).astype(np.float32)  # applied to every (t, n) row incl. padding
    p = "rnns.%d.rnn." % l
    out = rnn_direction(kind, x, lens, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"],
                        sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"], reverse=False)
    if bidirectional:
        out = out + rnn_direction(kind, x, lens, sd[p + "weight_ih_l0_reverse"],
                                  sd[p + "weight_hh_l0_reverse"], sd[p + "bias_ih_l0_reverse"],
                                  sd[p + "bias_hh_l0_reverse"], reverse=True)
    return out.astype(np.float32)


def lookahead(sd, x, context):
    """Lookahead (model.py:143-148) + Hardtanh(0,20) (model.py:407-411)."""
    w = sd["lookahead.0.conv.weight"][:, 0, :]  # [H, context]
    T = x.shape[0]
    xp = np.concatenate([x, np.zeros((context - 1,) + x.shape[1:], np.float32)], axis=0)
    out = np.zeros_like(x)
    for k in range(context):
        out += xp[k:k + T] * w[:, k]
    return np.clip(out, np.float32(0), np.float32(20)).astype(np.float32)


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def forward(sd, cfg, x, lengths, return_logits=False):
    """DeepSpeech.forward (model.py:496-515).

    cfg: dict(conv_layers, rnn_type in {'gru','lstm','rnn'}, rnn_hidden_size, rnn_layers,
              bidirectional, context).  x: [B,1,F,T] float32, lengths: [B] ints sorted
    descending (pack_padded_sequence's requirement, model.py:117).
    Returns (probs [B,T',C] float32, out_lens [B] int32).
    """
    lengths = np.asarray(lengths)
    if np.any(np.diff(lengths) > 0):
        raise RuntimeError("`lengths` array must be sorted in decreasing order")
    out_lens = get_seq_lens(lengths, cfg["conv_layers"])
    y = conv_stack(sd, x, out_lens, cfg["conv_layers"])
    B, C, F, T = y.shape
    y = y.reshape(B, C * F, T).transpose(2, 0, 1).copy()  # T x N x H
    for l in range(cfg["rnn_layers"]):
        y = batch_rnn(sd, l, cfg["rnn_type"], y, out_lens, cfg["bidirectional"], batch_norm=(l > 0))
    if not cfg["bidirectional"]:
        y = lookahead(sd, y, cfg["context"])
    a, b = _bn_affine(sd, "fc.0.module.0")
    y = (y * a + b).astype(np.float32)
    logits = (y.reshape(T * B, -1) @ sd["fc.0.module.1.weight"].T).reshape(T, B, -1)
    logits = logits.transpose(1, 0, 2).astype(np.float32)
    if return_logits:
        return logits, out_lens
    return softmax(logits), out_lens


def flops_per_clip(cfg, t_out, n_labels=33):
    """SURVEY 8(d) algorithmic FLOPs for one clip whose conv output length is t_out."""
    from danspeech_amd.synthetic import conv_out_freq
    f = 161
    total = 0
    t = None
    for li, (ci, co, kf, kt, sf, st, pf, pt) in enumerate(CONV_SPECS[:cfg["conv_layers"]]):
        f = (f + 2 * pf - kf) // sf + 1
        total += co * f * t_out * ci * kf * kt
    H = cfg["rnn_hidden_size"]; G = GATES[cfg["rnn_type"]]
    D = 2 if cfg["bidirectional"] else 1
    I = CONV_SPECS[cfg["conv_layers"] - 1][1] * conv_out_freq(161, cfg["conv_layers"])
    for l in range(cfg["rnn_layers"]):
        total += D * t_out * G * H * ((I if l == 0 else H) + H)
    total += t_out * H * n_labels
    return 2 * total
