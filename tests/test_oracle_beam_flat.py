"""oracle/beam_flat.py (the kernel's array formulation of the prefix beam search: implicit trie, edge tuples, no
child counts) must equal oracle/beam.py (ctcdecode's pointer trie restated) -- strings, timesteps, scores."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import beam as ob
from oracle import beam_flat as bf
from oracle.lm import Scorer


def _peaky(rng, T, C, sharp):
    logits = rng.standard_normal((T, C)) * sharp
    logits[:, 0] += 1.5
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32).astype(np.float64)


def _same(a, b, tol=1e-9):
    assert len(a) == len(b)
    for (sa, ta, oa), (sb, tb, ob_) in zip(a, b):
        if np.isinf(sa) and np.isinf(sb):
            continue
        assert ta == tb and oa == ob_, (ta, tb, oa, ob_)
        assert abs(sa - sb) <= tol * max(1.0, abs(sa)), (sa, sb)


@pytest.mark.parametrize("seed,T,beam,sharp", [(0, 40, 8, 3.0), (1, 80, 16, 2.0), (2, 120, 4, 1.0), (3, 60, 64, 3.0), (4, 200, 12, 0.5),
                                               (5, 150, 3, 2.0)])
def test_flat_equals_trie_without_lm(seed, T, beam, sharp):
    rng = np.random.default_rng(seed)
    labels = syn.DANSPEECH_LABELS
    probs = _peaky(rng, T, len(labels), sharp)
    _same(bf.ctc_beam_search(probs, labels, beam), ob.ctc_beam_search(probs, labels, beam))


def test_flat_small_alphabet_exhaustive():
    rng = np.random.default_rng(0)
    for b in range(3):
        probs = rng.dirichlet(np.ones(4), size=6).astype(np.float32).astype(np.float64)
        _same(bf.ctc_beam_search(probs, "_ab ", 64), ob.ctc_beam_search(probs, "_ab ", 64))
        _same(bf.ctc_beam_search(probs, "_ab ", 5), ob.ctc_beam_search(probs, "_ab ", 5))


@pytest.mark.parametrize("seed,order,T,beam,sharp,top_n,cp", [(10, 3, 80, 16, 2.0, 40, 1.0), (11, 5, 60, 32, 1.5, 40, 1.0), (12, 3, 160, 6, 1.0, 40, 1.0),
                                                              (13, 3, 120, 12, 2.5, 15, 0.98), (14, 2, 100, 4, 0.7, 40, 1.0)])
def test_flat_equals_trie_with_lm(tmp_path, seed, order, T, beam, sharp, top_n, cp):
    labels = syn.DANSPEECH_LABELS
    path = str(tmp_path / "lm.arpa")
    syn.make_arpa(path, order=order, n_words=200, seed=seed, ngrams_per_order=500)
    rng = np.random.default_rng(seed)
    probs = _peaky(rng, T, len(labels), sharp)
    sc = Scorer(1.3, 0.2, path, labels)
    bf.stats.update(revivals=0, walk_hops=0, frames=0, inherit_hops=0)
    got = bf.ctc_beam_search(probs, labels, beam, cp, top_n, 0, sc)
    want = ob.ctc_beam_search(probs, labels, beam, cp, top_n, 0, sc)
    _same(got, want)


def test_revival_and_walk_paths_are_exercised():
    """Flat, noisy distributions with a narrow beam make prefixes leave the beam and come back: the slow path (a dormant
    top re-entering, the entries below it re-hung by a walk through the node pool) must have run in this suite's inputs."""
    labels = "_abc"
    rng = np.random.default_rng(42)
    bf.stats.update(revivals=0, walk_hops=0, frames=0, inherit_hops=0)
    for k in range(30):
        probs = rng.dirichlet(np.ones(4) * 0.6, size=40)
        for beam in (2, 3, 5, 9):
            _same(bf.ctc_beam_search(probs, labels, beam), ob.ctc_beam_search(probs, labels, beam))
    assert bf.stats["revivals"] > 0 and bf.stats["inherit_hops"] > 0
    # ... and with the entries below the returning prefix more than one level down (the walk proper); the GPU test
    # tests/test_gpu_beam.py::test_beam_dormant_prefixes_come_back uses these inputs
    bf.stats.update(revivals=0, walk_hops=0, frames=0, inherit_hops=0)
    for seed, beam in ((80, 4), (61, 3), (112, 6), (119, 3), (165, 4), (192, 6)):
        probs = np.random.default_rng(seed).dirichlet(np.ones(4) * 0.5, size=60).astype(np.float32).astype(np.float64)
        _same(bf.ctc_beam_search(probs, labels, beam), ob.ctc_beam_search(probs, labels, beam))
    assert bf.stats["walk_hops"] > 0
