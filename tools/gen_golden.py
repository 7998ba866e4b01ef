#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports danspeech from /root/reference with the stub recipe of SURVEY.md App. B
(``Levenshtein``/``librosa``/``wget`` are absent and carry no arithmetic on the paths
captured here; ``scipy.signal.hamming`` moved to ``scipy.signal.windows``), feeds it
seeded inputs/weights from ``danspeech_amd.synthetic`` and stores inputs + the
reference's outputs.  Only data is written: no reference source or bytecode leaves
this container (``sys.dont_write_bytecode``).

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _n in ("Levenshtein", "librosa", "wget"):
    sys.modules[_n] = types.ModuleType(_n)
import scipy.signal  # noqa: E402
import scipy.signal.windows as _W  # noqa: E402
for _w in ("hamming", "hann", "blackman", "bartlett"):
    setattr(scipy.signal, _w, getattr(_W, _w))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from danspeech.deepspeech.model import DeepSpeech, BatchRNN, supported_rnns  # noqa: E402
from danspeech.deepspeech.decoder import GreedyDecoder  # noqa: E402
from danspeech.audio import load_audio  # noqa: E402

from danspeech_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def ref_model(cfg, sd, labels=None):
    m = DeepSpeech("golden", rnn_type=supported_rnns[cfg["rnn_type"]], labels=labels,
                   rnn_hidden_size=cfg["rnn_hidden_size"], rnn_layers=cfg["rnn_layers"],
                   bidirectional=cfg["bidirectional"], context=cfg["context"],
                   conv_layers=cfg["conv_layers"])
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.eval()
    return m


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def g1_seq_lens():
    T = np.arange(1, 3002, dtype=np.int64)
    out = {}
    for cl in (1, 2, 3):
        cfg = dict(conv_layers=cl, rnn_type="gru", rnn_hidden_size=8, rnn_layers=1,
                   bidirectional=True, context=20)
        m = ref_model(cfg, syn.make_state_dict(cl, "gru", 8, 1, seed=1))
        out["conv%d" % cl] = m.get_seq_lens(torch.from_numpy(T).int()).numpy().astype(np.int32)
    save("g1_seq_lens", T=T.astype(np.int32), **out)


def g2_conv():
    for cl in (1, 2, 3):
        cfg = dict(conv_layers=cl, rnn_type="gru", rnn_hidden_size=8, rnn_layers=1,
                   bidirectional=True, context=20)
        sd = syn.make_state_dict(cl, "gru", 8, 1, seed=20 + cl)
        m = ref_model(cfg, sd)
        lens = np.array([41, 30, 13], dtype=np.int32) if cl < 3 else np.array([37, 22], dtype=np.int32)
        x = syn.make_features(len(lens), int(lens[0]), seed=30 + cl)
        for i, L in enumerate(lens):
            x[i, :, :, L:] = 0
        with torch.no_grad():
            ol = m.get_seq_lens(torch.from_numpy(lens))
            y, _ = m.conv(torch.from_numpy(x), ol)
        save("g2_conv%d" % cl, seed=20 + cl, x_seed=30 + cl, lens=lens, out_lens=ol.numpy().astype(np.int32),
             y=y.numpy(), sd_sha=sd_hash(sd))


def g3_batch_rnn():
    """BatchRNN alone.  Shapes are chosen so that the layer also exists inside a DeepSpeech
    model (needed to drive it through the C ABI): bn=0 is a layer 0 with I = 32 (1 conv layer at
    n_freq = 2 gives rnn_input_size 32), bn=1 is a layer >= 1 with I = H = 16."""
    rng = np.random.default_rng(5)
    T, B, H = 12, 3, 16
    lens = np.array([12, 9, 4], dtype=np.int32)
    xs = {}
    for I in (32, 16):
        x = rng.standard_normal((T, B, I)).astype(np.float32)
        for b, L in enumerate(lens):
            x[L:, b] = 0
        xs[I] = x
    out = {"x_bn0": xs[32], "x_bn1": xs[16], "lens": lens}
    for kind in ("gru", "lstm", "rnn"):
        for bn in (False, True):
            for bidir in (True, False):
                torch.manual_seed(3)
                I = 16 if bn else 32
                r = BatchRNN(I, H, rnn_type=supported_rnns[kind], bidirectional=bidir, batch_norm=bn)
                if bn:
                    bnm = r.batch_norm.module
                    with torch.no_grad():
                        bnm.weight.uniform_(0.5, 1.5); bnm.bias.normal_(0, 0.1)
                        bnm.running_mean.normal_(0, 0.1); bnm.running_var.uniform_(0.5, 1.5)
                r.eval()
                with torch.no_grad():
                    y = r(torch.from_numpy(xs[I]), torch.from_numpy(lens))
                tag = "%s_bn%d_bi%d" % (kind, bn, bidir)
                out["y_" + tag] = y.numpy()
                for k, v in r.state_dict().items():
                    out["w_%s__%s" % (tag, k)] = v.numpy()
    save("g3_batch_rnn", **out)


def g4_forward_small():
    out = {}
    i = 0
    for kind in ("gru", "lstm", "rnn"):
        for bidir in (True, False):
            for cl in (1, 2, 3):
                if cl != 2 and not (kind == "gru" and bidir):
                    continue
                i += 1
                cfg = dict(conv_layers=cl, rnn_type=kind, rnn_hidden_size=32, rnn_layers=3,
                           bidirectional=bidir, context=6)
                sd = syn.make_state_dict(cl, kind, 32, 3, bidirectional=bidir, context=6, seed=100 + i)
                m = ref_model(cfg, sd)
                lens = np.array([120, 97, 40], dtype=np.int32)
                x = syn.make_features(3, 120, seed=200 + i)
                for b, L in enumerate(lens):
                    x[b, :, :, L:] = 0
                with torch.no_grad():
                    p, ol = m(torch.from_numpy(x), torch.from_numpy(lens))
                tag = "%s_bi%d_c%d" % (kind, bidir, cl)
                out["probs_" + tag] = p.numpy()
                out["outlens_" + tag] = ol.numpy().astype(np.int32)
                out["seeds_" + tag] = np.array([100 + i, 200 + i])
                out["sha_" + tag] = sd_hash(sd)
    out["lens"] = np.array([120, 97, 40], dtype=np.int32)
    save("g4_forward_small", **out)


def g4_forward_full():
    """cfgA (2 conv, 5 x BiGRU 800), fc_gain=8 (sharpened, G7), B=2 ragged 10 s clips."""
    cfg = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5,
               bidirectional=True, context=20)
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, fc_gain=8.0)
    m = ref_model(cfg, sd)
    lens = np.array([1001, 777], dtype=np.int32)
    x = syn.make_features(2, 1001, seed=7)
    x[1, :, :, 777:] = 0
    with torch.no_grad():
        p, ol = m(torch.from_numpy(x), torch.from_numpy(lens))
        dec = GreedyDecoder(m.labels, blank_index=m.labels.index("_"))
        strings, offsets = dec.decode(p, ol)
    top2 = np.sort(p.numpy(), axis=2)[:, :, -2:]
    margin = (top2[:, :, 1] - top2[:, :, 0])
    print("full-size: min top-2 margin over valid frames:",
          min(margin[0, :501].min(), margin[1, :int(ol[1])].min()))
    save("g4_forward_full", probs=p.numpy(), out_lens=ol.numpy().astype(np.int32), lens=lens,
         strings=np.array([s[0] for s in strings]), off0=offsets[0][0].numpy(), off1=offsets[1][0].numpy(),
         sha=sd_hash(sd), margin_min=margin.min())


def g5_greedy():
    labels = syn.DANSPEECH_LABELS
    dec = GreedyDecoder(labels, blank_index=labels.index("_"))
    rng = np.random.default_rng(9)
    C = len(labels)
    cases = []
    # crafted id sequences: repeats, blank-separated repeats, leading/trailing blanks, spaces
    seqs = [
        [0, 0, 1, 1, 1, 0, 1, 2, 2, 0, 0],
        [5, 5, 0, 5, 32, 32, 7, 0, 32, 0],
        [0, 0, 0, 0],
        [3],
        [3, 3, 3, 3, 3],
        [32, 0, 32, 1, 0, 1, 1, 2, 0],
        list(rng.integers(0, C, size=40)),
        list(rng.integers(0, 4, size=60)),
    ]
    Tm = max(len(s) for s in seqs)
    probs = np.full((len(seqs), Tm, C), 0.0, dtype=np.float32)
    sizes = np.array([len(s) for s in seqs], dtype=np.int32)
    for b, s in enumerate(seqs):
        r = rng.uniform(0.0, 0.01, size=(Tm, C)).astype(np.float32)
        probs[b] = r
        for t, c in enumerate(s):
            probs[b, t, c] = 0.9
        # frames past `size` get a loud non-blank so a decoder ignoring sizes is caught
        probs[b, len(s):, 9] = 0.95
    # exact ties: two classes share the maximum -> lowest index must win (torch.max)
    probs[0, 3, 1] = 0.9; probs[0, 3, 4] = 0.9
    probs[1, 0, 5] = 0.9; probs[1, 0, 20] = 0.9
    sizes[6] = 33  # sizes shorter than T'
    strings, offsets = dec.decode(torch.from_numpy(probs), torch.from_numpy(sizes))
    # also sizes=None
    strings_n, offsets_n = dec.decode(torch.from_numpy(probs), None)
    save("g5_greedy", probs=probs, sizes=sizes,
         strings=np.array([s[0] for s in strings]),
         offsets=np.array([np.pad(o[0].numpy(), (0, Tm - len(o[0])), constant_values=-1) for o in offsets]),
         strings_nosize=np.array([s[0] for s in strings_n]),
         offsets_nosize=np.array([np.pad(o[0].numpy(), (0, Tm - len(o[0])), constant_values=-1) for o in offsets_n]))


def g6_audio():
    path = "/root/reference/example_files/u0013002.wav"
    y = load_audio(path)
    save("g6_audio", n=len(y), vmin=y.min(), vmax=y.max(), dtype=str(y.dtype),
         sha256=hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest(),
         head=y[:64], tail=y[-64:], wav_sha256=hashlib.sha256(open(path, "rb").read()).hexdigest())
    print("load_audio:", len(y), y.min(), y.max(), y.dtype)


def stream_state_dict(sd):
    """synthetic (non-streaming unidirectional) names -> the streaming model's: LookaheadStream is a
    direct attribute (model.py:490), not the first module of a Sequential (model.py:407-411)."""
    return {("lookahead.conv.weight" if k == "lookahead.0.conv.weight" else k): v for k, v in sd.items()}


def g8_streaming():
    """DeepSpeech(streaming_inference_model=True).streaming_forward (model.py:517-537) over chunked
    feature streams: MaskConvStream / BatchRNNStream / LookaheadStream carried state, first pass without
    output, last pass with the right padding, and a second utterance on the same model (state reset)."""
    out = {}
    # only 2-conv streaming models work in the reference: streaming_init sizes the first RNN for two conv
    # layers whatever conv_layers says (model.py:476-484), and the 1-conv branch builds a plain MaskConv
    cases = [("gru", 2, [86, 39, 39, 52, 25]), ("lstm", 2, [70, 45, 40]), ("rnn", 2, [64, 41])]
    for kind, cl, chunks in cases:
        H, L, ctx = 32, 3, 6
        tag = "%s_c%d" % (kind, cl)
        sd = syn.make_state_dict(cl, kind, H, L, bidirectional=False, context=ctx, seed=81, fc_gain=4.0)
        m = DeepSpeech("golden-stream", rnn_type=supported_rnns[kind], rnn_hidden_size=H, rnn_layers=L,
                       bidirectional=False, context=ctx, conv_layers=cl, streaming_inference_model=True)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in stream_state_dict(sd).items()})
        m.eval()
        for utt in range(2):                      # the second utterance checks that is_last resets every state
            for ci, T in enumerate(chunks):
                x = syn.make_features(1, T, seed=8100 + 100 * utt + ci)
                with torch.no_grad():
                    y = m(torch.from_numpy(x), ci == 0, ci == len(chunks) - 1)
                key = "%s_u%d_k%d" % (tag, utt, ci)
                out["probs_" + key] = np.zeros((0,), np.float32) if y is None else y.numpy()[0]
        out["chunks_" + tag] = np.array(chunks, dtype=np.int32)
    save("g8_streaming", **out)
    for k in sorted(out):
        if k.startswith("probs_gru_c2_u0"):
            print(k, out[k].shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4s", "g4f", "g5", "g6", "g8"]
    fns = dict(g1=g1_seq_lens, g2=g2_conv, g3=g3_batch_rnn, g4s=g4_forward_small,
               g4f=g4_forward_full, g5=g5_greedy, g6=g6_audio, g8=g8_streaming)
    for w in which:
        fns[w]()
