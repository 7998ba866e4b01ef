"""Seeded synthetic inputs for the hot path: model weights, audio clips, n-gram LMs.

Nothing can be downloaded on the build or GPU hosts (the reference's pretrained
``.pth`` / ``.klm`` artefacts live behind URLs, reference
``danspeech/pretrained_models/danspeech_primary.py:5,22-25``), so benchmarks and
parity tests run on weights of the reference's *shapes* drawn from a fixed PRNG.

The generators use ``numpy.random.default_rng`` (PCG64, stable across numpy
versions) so the GPU box regenerates bit-identical tensors from ``(shape, seed)``
and full-size weights never need to be committed.

State-dict key names and tensor shapes follow the reference module tree
(``danspeech/deepspeech/model.py:358-420``): ``conv.seq_module.{0,3,6}`` Conv2d,
``conv.seq_module.{1,4,7}`` BatchNorm2d, ``rnns.{l}.rnn.weight_ih_l0[_reverse]`` ...,
``rnns.{l}.batch_norm.module.*`` (layers >= 1 only), ``lookahead.0.conv.weight``
(unidirectional only), ``fc.0.module.0`` BatchNorm1d, ``fc.0.module.1.weight``.
"""
from collections import OrderedDict

import numpy as np

GATES = {"gru": 3, "lstm": 4, "rnn": 1}

# (c_in, c_out, k_f, k_t, stride_f, stride_t, pad_f, pad_t) -- model.py:359,372,389
CONV_SPECS = [
    (1, 32, 41, 11, 2, 2, 20, 5),
    (32, 32, 21, 11, 2, 1, 10, 5),
    (32, 96, 21, 11, 2, 1, 10, 5),
]

DANSPEECH_LABELS = "_abcdefghijklmnopqrstuvwxyzæøåéü "


def conv_out_freq(n_freq, conv_layers):
    """Frequency bins after the conv stack (model.py:354-355,378,394-395)."""
    f = n_freq
    for (_, _, kf, _, sf, _, pf, _) in CONV_SPECS[:conv_layers]:
        f = (f + 2 * pf - kf) // sf + 1
    return f


def rnn_input_size(conv_layers, sample_rate=16000, window_size=0.02):
    n_freq = int(sample_rate * window_size) // 2 + 1
    return CONV_SPECS[conv_layers - 1][1] * conv_out_freq(n_freq, conv_layers)


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _bn(rng, sd, prefix, n):
    # Non-trivial affine + running stats: the default init (1, 0, 0, 1) would hide
    # BatchNorm bugs.
    sd[prefix + ".weight"] = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    sd[prefix + ".bias"] = rng.normal(0.0, 0.1, size=n).astype(np.float32)
    sd[prefix + ".running_mean"] = rng.normal(0.0, 0.1, size=n).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.array(0, dtype=np.int64)


def make_state_dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5,
                    bidirectional=True, n_labels=33, context=20, seed=0, fc_gain=8.0,
                    sample_rate=16000, window_size=0.02, ih_gain=1.0, blank_boost=0.0):
    """Return an OrderedDict name -> np.ndarray mirroring DeepSpeech.state_dict().

    Distributions follow torch's default initialisers for the same modules
    (Conv2d: U(+-1/sqrt(fan_in)); RNN: U(+-1/sqrt(H)); Linear: U(+-1/sqrt(in)));
    ``fc_gain`` sharpens the logits so that the greedy argmax has a margin well
    above fp32 reassociation noise.

    With torch's default scales a deep random GRU stack forgets its input: every frame gets
    the same argmax and a 10 s clip decodes to one or two characters, which makes "identical
    transcripts" a vacuous check.  ``ih_gain`` multiplies every ``weight_ih`` (the input drives
    the state harder: ``ih_gain=6`` gives 130-170 tokens per 10 s clip with repeats) and
    ``blank_boost`` adds that many logits to the blank (label 0) through the FC BatchNorm's
    bias, so that blanks separate repeated characters as in a trained CTC model.  Both default
    to the neutral value: the golden vectors of tests/golden were generated without them.
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    H = rnn_hidden_size
    G = GATES[rnn_type]
    for li, (ci, co, kf, kt, *_r) in enumerate(CONV_SPECS[:conv_layers]):
        bound = 1.0 / np.sqrt(ci * kf * kt)
        sd["conv.seq_module.%d.weight" % (3 * li)] = _uniform(rng, (co, ci, kf, kt), bound)
        sd["conv.seq_module.%d.bias" % (3 * li)] = _uniform(rng, (co,), bound)
        _bn(rng, sd, "conv.seq_module.%d" % (3 * li + 1), co)
    in0 = rnn_input_size(conv_layers, sample_rate, window_size)
    bound = 1.0 / np.sqrt(H)
    for l in range(rnn_layers):
        I = in0 if l == 0 else H
        if l > 0:
            _bn(rng, sd, "rnns.%d.batch_norm.module" % l, H)
        for suffix in ([""] + (["_reverse"] if bidirectional else [])):
            sd["rnns.%d.rnn.weight_ih_l0%s" % (l, suffix)] = _uniform(rng, (G * H, I), bound) * np.float32(ih_gain)
            sd["rnns.%d.rnn.weight_hh_l0%s" % (l, suffix)] = _uniform(rng, (G * H, H), bound)
            sd["rnns.%d.rnn.bias_ih_l0%s" % (l, suffix)] = _uniform(rng, (G * H,), bound)
            sd["rnns.%d.rnn.bias_hh_l0%s" % (l, suffix)] = _uniform(rng, (G * H,), bound)
    if not bidirectional:
        sd["lookahead.0.conv.weight"] = _uniform(rng, (H, 1, context), 1.0 / np.sqrt(context))
    _bn(rng, sd, "fc.0.module.0", H)
    sd["fc.0.module.1.weight"] = (_uniform(rng, (n_labels, H), bound) * np.float32(fc_gain))
    if blank_boost:
        w0 = sd["fc.0.module.1.weight"][0].astype(np.float64)
        sd["fc.0.module.0.bias"] = (sd["fc.0.module.0.bias"] + blank_boost * w0 / np.dot(w0, w0)).astype(np.float32)
    return sd


# The weights of the benchmark and of the workload-level parity tests: transcripts of >= 100 tokens per 10 s clip.
TALKATIVE = dict(fc_gain=8.0, ih_gain=6.0, blank_boost=8.0)


def make_clip(index, n_samples=160000, seed=1234):
    """int16-scale mono PCM as float64, the dtype ``load_audio`` hands to recognize()
    (reference danspeech/audio/resources.py:640). Gaussian noise plus a few sinusoid
    bursts so the spectrum is not flat."""
    rng = np.random.default_rng(seed + index)
    x = 3000.0 * rng.standard_normal(n_samples)
    t = np.arange(n_samples) / 16000.0
    for _ in range(4):
        f = rng.uniform(100.0, 4000.0)
        a = rng.uniform(1000.0, 6000.0)
        s = int(rng.integers(0, max(1, n_samples - 1)))
        e = min(n_samples, s + int(rng.integers(1600, 32000)))
        x[s:e] += a * np.sin(2 * np.pi * f * t[s:e])
    return np.clip(np.rint(x), -32768, 32767).astype(np.float64)


def make_features(batch, n_frames, n_freq=161, seed=7):
    """Standard-normal features [B,1,F,T] (what a z-normalised spectrogram looks like)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, 1, n_freq, n_frames)).astype(np.float32)


def make_vocabulary(n_words=5000, seed=11, labels=DANSPEECH_LABELS):
    """Pseudo-Danish word list over the label alphabet (no blank, no space)."""
    rng = np.random.default_rng(seed)
    letters = [c for c in labels if c not in "_ "]
    # Zipf-ish letter weights so words share prefixes like a natural lexicon.
    w = 1.0 / np.arange(1, len(letters) + 1) ** 0.8
    w /= w.sum()
    words = set()
    while len(words) < n_words:
        n = int(rng.integers(1, 9))
        words.add("".join(rng.choice(letters, size=n, p=w)))
    return sorted(words)


def make_arpa(path, order=3, n_words=5000, seed=11, ngrams_per_order=20000,
              labels=DANSPEECH_LABELS):
    """Write a seeded synthetic ARPA n-gram LM (stand-in for dsl_3gram.klm, which
    cannot be fetched: reference danspeech/language_models/dsl_3gram.py:4,16-20).

    Every n-gram of order k>1 has all its (k-1)-gram prefixes and suffixes present,
    as a real back-off model does.
    """
    rng = np.random.default_rng(seed + 1000 * order)
    vocab = ["<unk>", "<s>", "</s>"] + make_vocabulary(n_words, seed, labels)
    V = len(vocab)
    grams = [None] * (order + 1)
    grams[1] = [(i,) for i in range(V)]
    have = {1: set(grams[1])}
    for k in range(2, order + 1):
        s = set()
        prev = list(have[k - 1])
        tries = 0
        while len(s) < ngrams_per_order and tries < ngrams_per_order * 20:
            tries += 1
            g = prev[int(rng.integers(0, len(prev)))]
            w = int(rng.integers(2, V))  # never predict <unk>/<s>
            cand = g + (w,)
            if cand[0] == 2 or 1 in cand[1:] or 2 in cand[:-1]:
                continue
            if cand[1:] in have[k - 1]:
                s.add(cand)
        have[k] = s
        grams[k] = sorted(s)
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        for k in range(1, order + 1):
            f.write("ngram %d=%d\n" % (k, len(grams[k])))
        for k in range(1, order + 1):
            f.write("\n\\%d-grams:\n" % k)
            for g in grams[k]:
                if k == 1 and g[0] == 1:
                    lp = -99.0
                else:
                    lp = -float(rng.uniform(0.3, 5.0 - 0.6 * k))
                words = " ".join(vocab[i] for i in g)
                if k < order and g[-1] != 2:
                    bo = -float(rng.uniform(0.0, 1.0))
                    f.write("%.6f\t%s\t%.6f\n" % (lp, words, bo))
                else:
                    f.write("%.6f\t%s\n" % (lp, words))
        f.write("\n\\end\\\n")
    return vocab


def gated_signal(plan, seed):
    """int16 signal for the long-form segmentation tests: ``plan`` = [[n_samples, rms], ...], each stretch white noise of
    that RMS (speech-like energy where rms is well above the script's threshold of 600, silence where it is well below)."""
    rng = np.random.default_rng(seed)
    parts = [np.clip(np.round(rng.standard_normal(int(n)) * float(rms)), -32768, 32767).astype(np.int16) for n, rms in plan]
    return np.concatenate(parts)


_C = 1024   # the simulated stream's chunk (example_scripts/video_transcribe_simulation.py:71)
# (name, plan, seed, --offset seconds): phrases of several lengths, pauses around the 9-chunk closing rule, blips around the
# 4-chunk keeping rule, speech in the very first chunks, speech running into the end, boundaries off the chunk grid,
# energies hovering around the threshold, an offset start
SEGMENT_CASES = [
    ("phrases", [[12 * _C, 60], [20 * _C, 3000], [10 * _C, 60], [9 * _C, 3000], [9 * _C, 60], [7 * _C, 2500], [30 * _C, 60],
                 [4 * _C, 3000], [12 * _C, 60], [5 * _C, 3000], [12 * _C, 60], [6 * _C, 3000], [11 * _C, 60]], 101, 0),
    ("starts_speaking", [[15 * _C, 2000], [14 * _C, 50], [_C, 50], [8 * _C, 2600], [10 * _C, 50], [20 * _C, 2600]], 102, 0),
    ("second_chunk", [[_C, 50], [9 * _C, 2000], [25 * _C, 50]], 103, 0),
    ("off_grid", [[7 * _C + 311, 80], [13 * _C + 97, 2800], [9 * _C + 512, 80], [6 * _C + 1000, 2800], [10 * _C + 3, 80],
                  [5 * _C + 700, 2800], [15 * _C + 5, 80]], 104, 0),
    ("around_threshold", [[6 * _C, 560], [6 * _C, 640], [3 * _C, 590], [8 * _C, 610], [12 * _C, 580], [7 * _C, 620], [2 * _C, 600],
                          [9 * _C, 601], [14 * _C, 599], [10 * _C, 1200], [20 * _C, 300]], 105, 0),
    ("offset_start", [[20 * _C, 3000], [12 * _C, 60], [18 * _C, 3000], [14 * _C, 60], [9 * _C, 3000], [20 * _C, 60]], 106, 1),
]
