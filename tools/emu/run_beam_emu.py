#!/usr/bin/env python3
"""Runs the beam-search kernel's source on the CPU SIMT emulation (tools/emu/beam_emu.cpp) and compares the beams with
oracle/beam.py -- a debugging aid for the kernel's phase structure (a hang shows up here, not on a GPU box).

    python tools/emu/run_beam_emu.py [--threads 1024|512|64] [--case small|revive|lm|ties]
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from danspeech_amd import synthetic as syn  # noqa: E402
from oracle import beam as ob  # noqa: E402

EXE = os.path.join(ROOT, "tools", "emu", "beam_emu")


def build():
    src = os.path.join(ROOT, "tools", "emu", "beam_emu.cpp")
    deps = [src, os.path.join(ROOT, "tools", "emu", "simt.h"), os.path.join(ROOT, "danspeech_amd", "csrc", "beam_kernel.inc")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-DSIMT_EMU", "-O1", "-g", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "danspeech_amd", "csrc"),
                               "-I", os.path.join(ROOT, "tools", "emu"), src, "-o", EXE])


def run(probs, sizes, labels, beam, lm_path=None, alpha=0.0, beta=0.0, top_n=40, cutoff_prob=1.0, threads=1024, timeout=600):
    B, T, C = probs.shape
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(pin, "wb") as f:
            f.write(struct.pack("<8i", B, T, C, beam, 0, top_n, 1 if lm_path else 0, len(labels)))
            f.write(struct.pack("<3d", cutoff_prob, alpha, beta))
            f.write(np.ascontiguousarray(probs, dtype=np.float32).tobytes())
            f.write(np.asarray(sizes if sizes is not None else [T] * B, dtype=np.int32).tobytes())
            for c in labels:
                e = c.encode("utf-8")
                f.write(struct.pack("<i", len(e)) + e)
            e = (lm_path or "").encode()
            f.write(struct.pack("<i", len(e)) + e)
        t0 = time.time()
        subprocess.run([EXE, pin, pout, str(threads)], check=True, timeout=timeout)
        dt = time.time() - t0
        raw = open(pout, "rb").read()
    n = B * beam * T
    tok = np.frombuffer(raw, dtype=np.int32, count=n).reshape(B, beam, T)
    step = np.frombuffer(raw, dtype=np.int32, count=n, offset=4 * n).reshape(B, beam, T)
    ln = np.frombuffer(raw, dtype=np.int32, count=B * beam, offset=8 * n).reshape(B, beam)
    nout = np.frombuffer(raw, dtype=np.int32, count=B, offset=8 * n + 4 * B * beam)
    score = np.frombuffer(raw, dtype=np.float64, count=B * beam, offset=8 * n + 4 * B * beam + 4 * B).reshape(B, beam)
    return tok, step, ln, nout, score, dt


def compare(probs, sizes, labels, beam, threads, **kw):
    tok, step, ln, nout, score, dt = run(probs, sizes, labels, beam, threads=threads, **kw)
    scorer = ob.Scorer(kw.get("alpha", 0.0), kw.get("beta", 0.0), kw["lm_path"], labels) if kw.get("lm_path") else None
    bad = 0
    for b in range(probs.shape[0]):
        n = probs.shape[1] if sizes is None else int(sizes[b])
        ref = ob.ctc_beam_search(probs[b, :n].astype(np.float64), labels, beam, kw.get("cutoff_prob", 1.0), kw.get("top_n", 40), 0, scorer)
        assert nout[b] == len(ref), (b, nout[b], len(ref))
        for p, (s, t, o) in enumerate(ref):
            if np.isinf(s):
                continue
            total = score[b, p]          # the kernel reports the total; the host strips the LM terms: compare tokens and steps here
            if list(tok[b, p, :ln[b, p]]) != t or list(step[b, p, :ln[b, p]]) != o:
                bad += 1
                if bad < 5:
                    print("MISMATCH b=%d p=%d\n  got  %s %s\n  want %s %s" % (b, p, list(tok[b, p, :ln[b, p]]), list(step[b, p, :ln[b, p]]), t, o))
    print("%d utterances, beam %d, %d threads: %s in %.1f s" % (probs.shape[0], beam, threads, "OK" if not bad else "%d MISMATCHES" % bad, dt))
    return bad == 0


def peaky(rng, B, T, C, sharp):
    logits = rng.standard_normal((B, T, C)) * sharp
    logits[:, :, 0] += 1.5
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=1024)
    ap.add_argument("--case", default="small")
    a = ap.parse_args()
    build()
    rng = np.random.default_rng(0)
    ok = True
    if a.case in ("small", "all"):
        probs = rng.dirichlet(np.ones(4), size=(2, 6)).astype(np.float32)
        ok &= compare(probs, None, "_ab ", 64, a.threads)
    if a.case in ("revive", "all"):
        probs = np.random.default_rng(42).dirichlet(np.ones(4) * 0.6, size=(4, 40)).astype(np.float32)
        for beam in (2, 5):
            ok &= compare(probs, None, "_abc", beam, a.threads)
    if a.case in ("labels", "all"):
        ok &= compare(peaky(np.random.default_rng(1), 2, 30, 33, 3.0), np.array([30, 17]), syn.DANSPEECH_LABELS, 16, a.threads)
    if a.case in ("lm", "all"):
        path = os.path.join(tempfile.gettempdir(), "emu3.arpa")
        syn.make_arpa(path, order=3, n_words=200, seed=5, ngrams_per_order=600)
        ok &= compare(peaky(np.random.default_rng(2), 2, 40, 33, 2.0), np.array([40, 20]), syn.DANSPEECH_LABELS, 16, a.threads,
                      lm_path=path, alpha=1.3, beta=0.2)
    if a.case in ("walk", "all"):
        for seed, beam in ((80, 4), (61, 3), (112, 6), (119, 3)):
            probs = np.random.default_rng(seed).dirichlet(np.ones(4) * 0.5, size=(1, 60)).astype(np.float32)
            ok &= compare(probs, None, "_abc", beam, a.threads)
    if a.case in ("ties", "all"):
        # exact ties fall by the kernel's slot numbering: compare with oracle/beam_flat.py numbering slots for this thread count
        from oracle import beam_flat as bf
        bf.KERNEL_THREADS = a.threads
        labels = syn.DANSPEECH_LABELS
        C = len(labels)
        probs = np.full((2, 12, C), 1.0 / C, dtype=np.float32)
        probs[1, 3:6, 5:] = 0.0
        probs[1, 3:6, :5] = 0.2
        tok, step, ln, nout, score, dt = run(probs, None, labels, 24, threads=a.threads)
        for b in range(2):
            res = bf.ctc_beam_search(probs[b].astype(np.float64), labels, 24)
            # which of several exactly tied candidates stays depends on the last bit of exp / log (libm here, ocml on the GPU):
            # the scores of the beams are determined, their identities among tied ones are not
            want = np.sort([r[0] for r in res])
            got = np.sort(-score[b, :nout[b]])
            if len(want) != len(got) or np.abs(want - got).max() > 1e-6:
                print("TIES MISMATCH", b, want[:5], got[:5])
                ok = False
        print("ties: %s in %.1f s" % ("OK" if ok else "FAILED", dt))
    if a.case in ("cutoff", "all"):
        ok &= compare(peaky(np.random.default_rng(4), 1, 25, 33, 3.0), None, syn.DANSPEECH_LABELS, 12, a.threads, top_n=10, cutoff_prob=0.98)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
