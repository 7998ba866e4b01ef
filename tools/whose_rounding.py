#!/usr/bin/env python3
"""Whose rounding flips a greedy transcript?  The 256 clips of tests/test_gpu_workloads.py::test_timed_path_recognize_batches_full_size
(8 batches x 32 ragged 4..10 s clips of cfgA, TALKATIVE weights) through FOUR implementations of the same forward:

    gpu-split   the product path: split-fp16 operands (3 MFMA products, fp32 accumulation), ring recurrent kernel
    gpu-f32     DSMI_DENSE_MODE=f32 DSMI_RNN_MODE=steps: fp32 operands on the fp32 MFMA, one launch per recurrent step
    port-1      oracle/torch_port.py (the reference's own CPU operators) at 1 thread
    port-N      the same at 16 threads (what the tests compare with)

For every batch in which any two of them decode a clip differently: the frame(s) where the argmax differs, each implementation's
top-2 labels and probabilities there, and who agrees with whom.  Reference: danspeech/deepspeech/decoder.py:183-198 (argmax per frame).
    python tools/whose_rounding.py            (GPU box; about two minutes)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from danspeech_amd import synthetic as syn

LABELS = syn.DANSPEECH_LABELS
CFG = dict(conv_layers=2, rnn_type="gru", rnn_hidden_size=800, rnn_layers=5, bidirectional=True, context=20)


def ragged(B, lo, hi, seed):
    rng = np.random.default_rng(seed)
    n = np.sort(rng.integers(lo, hi + 1, size=B))[::-1].copy()
    n[0] = hi
    return [syn.make_clip(100 * seed + i, int(k)) for i, k in enumerate(n)]


def batch(k):
    return ragged(32, 64000, 160000, seed=20 + k)          # longest first (the test shuffles; the forward sorts back)


def gpu_probs(ks):
    """As the pipeline runs them: batches 2j and 2j + 1 merged into ONE 64-clip forward, longest first (the four-tile ring window)."""
    import torch
    from danspeech_amd import _native
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
    m = _native.NativeModel(CFG, sd)
    m.set_inflight(4)
    fe = _native.NativeFrontend()
    out = {}
    for k0 in sorted({k & ~1 for k in ks}):
        clips = batch(k0) + batch(k0 + 1)
        order = np.argsort([-len(c) for c in clips], kind="stable")
        n = np.array([len(clips[i]) for i in order], dtype=np.int64)
        feat, frames = fe.features(torch.from_numpy(np.concatenate([clips[i] for i in order])).cuda(), n)
        probs, ol = m.forward(feat, frames)
        probs, ol = probs.cpu().numpy(), np.asarray(ol)
        back = np.empty(64, dtype=np.int64)
        back[order] = np.arange(64)
        for h in (0, 1):
            rows = back[32 * h:32 * h + 32]
            out[k0 + h] = (probs[rows], ol[rows])
    assert m.recompute_count() == 0
    return {k: out[k] for k in ks}


def port_probs(ks, threads):
    import torch
    from oracle import torch_port as tp
    torch.set_num_threads(threads)
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **syn.TALKATIVE)
    out = {}
    for k in ks:
        x, fr = tp.spectrogram_batch(batch(k))
        p, ol = tp.forward(sd, CFG, x, fr)
        out[k] = (np.asarray(p), np.asarray(ol))
    return out


def greedy(p, ol):
    from oracle import decoder as od
    s, _ = od.greedy_decode(p, ol, LABELS, 0)
    return [x[0] for x in s]


if len(sys.argv) > 2 and sys.argv[1] == "--child":        # gpu-f32 / port-1 run in processes of their own (env / thread pool)
    kind, path, ks = sys.argv[2], sys.argv[3], [int(a) for a in sys.argv[4:]]
    res = gpu_probs(ks) if kind == "gpu" else port_probs(ks, int(kind))
    np.savez(path, **{"p%d" % k: v[0] for k, v in res.items()}, **{"l%d" % k: v[1] for k, v in res.items()})
    sys.exit(0)


def child(kind, ks, env=None):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "o.npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, path] + [str(k) for k in ks],
                       env=dict(os.environ, **(env or {})), check=True, stdout=subprocess.DEVNULL)
        z = np.load(path)
        return {k: (z["p%d" % k], z["l%d" % k]) for k in ks}


def main():
    ks = list(range(8))
    impl = {"gpu-split": gpu_probs(ks)}
    impl["port-16"] = port_probs(ks, 16)
    impl["gpu-f32"] = child("gpu", ks, {"DSMI_DENSE_MODE": "f32", "DSMI_RNN_MODE": "steps"})
    text = {name: {k: greedy(*res[k]) for k in ks} for name, res in impl.items()}
    flips = [(k, b) for k in ks for b in range(32) if len({text[n][k][b] for n in impl}) > 1]
    print("256 clips; transcripts equal to port-16's: " + ", ".join("%s %d" % (n, sum(text[n][k][b] == text["port-16"][k][b] for k in ks for b in range(32))) for n in impl))
    print("max |probs - port-16|: " + ", ".join("%s %.3g" % (n, max(np.abs(impl[n][k][0][b, :impl[n][k][1][b]] - impl["port-16"][k][0][b, :impl[n][k][1][b]]).max()
                                                                        for k in ks for b in range(32))) for n in impl if n != "port-16"))
    if not flips:
        print("no clip decodes differently in any implementation")
        return
    impl["port-1"] = child("1", sorted({k for k, _ in flips}))
    names = ["gpu-split", "gpu-f32", "port-1", "port-16"]
    for k, b in flips:
        T = int(impl["port-16"][k][1][b])
        am = {n: impl[n][k][0][b, :T].argmax(-1) for n in names}
        frames = [t for t in range(T) if len({int(am[n][t]) for n in names}) > 1]
        print("\nbatch %d, clip %d (%d frames): transcripts %s" % (k, b, T, {n: text[n][k][b] == text["port-16"][k][b] for n in names if n in text}))
        for t in frames:
            print("  frame %d:" % t)
            for n in names:
                row = impl[n][k][0][b, t]
                top = np.argsort(row)[::-1][:2]
                print("    %-10s argmax %r  top-2 %r %.9f | %r %.9f  margin %+.3e" % (n, LABELS[top[0]], LABELS[top[0]], row[top[0]], LABELS[top[1]], row[top[1]], row[top[0]] - row[top[1]]))
            groups = {}
            for n in names:
                groups.setdefault(int(am[n][t]), []).append(n)
            print("    -> " + "  vs  ".join("%r: %s" % (LABELS[c], "+".join(v)) for c, v in groups.items()))


if __name__ == "__main__":
    main()
