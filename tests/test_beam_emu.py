"""The beam-search kernel's OWN source (danspeech_amd/csrc/beam_kernel.inc) compiled for the CPU on a small SIMT emulation
(tools/emu/simt.h: one host thread per GPU thread, real barriers, wave collectives through a rendezvous) and held to
oracle/beam.py -- the phase structure, the edge-tuple bookkeeping, revivals with pool walks, the scorer and vocabulary
pruning are exercised here without a GPU.  Test infrastructure, not a CPU path of the product: the emulation is thousands of
host threads for inputs of a few dozen frames.  (What it cannot see: memory ordering and timing on the real chip.)"""
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

from danspeech_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))


@pytest.fixture(scope="module")
def emu():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    import run_beam_emu
    try:
        run_beam_emu.build()
    except Exception as e:            # an older g++ without <barrier>
        pytest.skip("cannot build the emulation: %s" % e)
    return run_beam_emu


def test_kernel_source_on_the_emulation_small_alphabet(emu):
    probs = np.random.default_rng(0).dirichlet(np.ones(4), size=(2, 6)).astype(np.float32)
    assert emu.compare(probs, None, "_ab ", 64, 192)
    assert emu.compare(probs, None, "_ab ", 5, 192)


def test_kernel_source_on_the_emulation_revivals_and_walks(emu):
    for seed, beam in ((61, 3), (119, 3)):
        probs = np.random.default_rng(seed).dirichlet(np.ones(4) * 0.5, size=(1, 40)).astype(np.float32)
        assert emu.compare(probs, None, "_abc", beam, 192)


def test_kernel_source_on_the_emulation_scorer_and_pruning(emu):
    path = os.path.join(tempfile.gettempdir(), "emu_test3.arpa")
    syn.make_arpa(path, order=3, n_words=120, seed=5, ngrams_per_order=300)
    probs = emu.peaky(np.random.default_rng(2), 1, 24, 33, 2.0)
    assert emu.compare(probs, np.array([24]), syn.DANSPEECH_LABELS, 12, 192, lm_path=path, alpha=1.3, beta=0.2)
    assert emu.compare(emu.peaky(np.random.default_rng(4), 1, 16, 33, 3.0), None, syn.DANSPEECH_LABELS, 10, 192, top_n=10, cutoff_prob=0.98)
