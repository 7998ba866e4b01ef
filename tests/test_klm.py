"""KenLM binary (.klm) reader of libdsmi.so (csrc/lm_klm.cpp.inc) on the CPU: header parsing, both supported data
structures, every n-gram and back-off score against the ARPA text the binary was written from, refusal of what is not
supported.  The binaries come from oracle/klm.py's writer (a restatement of KenLM's published layout): a self-consistency
check, PARITY WITH KenLM's OWN FILES IS UNPINNED (no KenLM, no .klm offline) -- see tests/test_gallery_optin.py."""
import struct

import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import klm


@pytest.fixture(scope="module")
def native():
    from danspeech_amd import _native
    _native.lib()
    return _native


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("klm")
    out = {}
    for order, words, per in ((3, 300, 800), (5, 200, 400)):
        arpa = str(d / ("syn%d.arpa" % order))
        syn.make_arpa(arpa, order=order, n_words=words, seed=5 + order, ngrams_per_order=per)
        out[order] = dict(arpa=arpa)
        for mt, name in ((klm.PROBING, "probing"), (klm.TRIE, "trie")):
            p = str(d / ("syn%d_%s.klm" % (order, name)))
            klm.write_klm(arpa, p, mt)
            out[order][name] = p
    return out


def test_known_hashes():
    # MurmurHash64A, seed 0: the empty string hashes to 0 by construction (h = 0 ^ 0, all mixing steps keep 0)
    assert klm.murmur64a(b"") == 0
    # one-byte key, by hand: h = len * m; h ^= byte; h *= m; then the final avalanche
    m = 0xc6a4a7935bd1e995
    h = ((m ^ 0x61) * m) & klm.M64
    h ^= h >> 47
    h = (h * m) & klm.M64
    h ^= h >> 47
    assert klm.murmur64a(b"a") == h
    assert klm.ngram_key([3, 7]) == klm.combine(7, 3) and klm.ngram_key([1, 2, 3]) == klm.combine(klm.combine(3, 2), 1)
    assert klm.required_bits(0) == 0 and klm.required_bits(1) == 1 and klm.required_bits(255) == 8 and klm.required_bits(256) == 9
    assert klm.buckets_for(10, 1.5) == 15 and klm.buckets_for(1, 1.5) == 2


@pytest.mark.parametrize("order", [3, 5])
@pytest.mark.parametrize("name", ["probing", "trie"])
def test_reader_equals_arpa(native, files, order, name):
    ref = native.NativeLM(files[order]["arpa"])
    lm = native.NativeLM(files[order][name])
    assert lm.kind == "klm-" + name and lm.order == order == ref.order and lm.vocab_size == ref.vocab_size
    _, grams = klm.read_arpa(files[order]["arpa"])
    assert lm.word_index("<unk>") == 0 and lm.word_index("no-such-word") == -1
    rng = np.random.default_rng(1)
    for n in range(1, order + 1):
        for g, lp, bo in grams[n]:
            ids = [lm.word_index(w) for w in g]
            assert min(ids) >= 0
            got = lm.lookup(ids)
            assert got is not None, g
            assert got[0] == np.float32(lp) and got[1] == np.float32(bo if n < order else 0.0), (g, got, lp, bo)
    # back-off scores of random word sequences (mostly unseen n-grams): identical floats through both readers
    vocab = [g[0][0] for g in grams[1]]
    for _ in range(400):
        n = int(rng.integers(1, order + 1))
        ws = [vocab[int(i)] for i in rng.integers(0, len(vocab), size=n)]
        a = lm.cond_log10([lm.word_index(w) for w in ws])
        b = ref.cond_log10([ref.word_index(w) for w in ws])
        assert a == b, (ws, a, b)
    lm.close(); ref.close()


def test_python_reader_agrees(files):
    for name in ("probing", "trie"):
        r = klm.KlmReader(files[3][name])
        _, grams = klm.read_arpa(files[3]["arpa"])
        for w, i in r.ids.items():
            assert r.index(w) == i
        for g, lp, bo in grams[3][:200]:
            assert r.lookup([r.ids[w] for w in g])[0] == np.float32(lp)


def test_unsupported_and_damaged_files_are_refused(native, files, tmp_path):
    good = open(files[3]["probing"], "rb").read()

    def refused(blob, needle):
        p = tmp_path / "x.klm"
        p.write_bytes(blob)
        with pytest.raises(native.DsmiError) as e:
            native.NativeLM(str(p))
        assert needle in str(e.value), str(e.value)

    refused(good[:60], "truncated")
    refused(good.replace(b"version 5", b"version 4", 1), "format version")
    refused(good[:96] + struct.pack("<i", 3) + good[100:], "quantised trie")
    refused(good[:96] + struct.pack("<i", 4) + good[100:], "array-compressed")
    refused(good[:100] + b"\x00" + good[101:], "vocabulary strings")
    refused(good[:64] + struct.pack("<f", 0.5) + good[68:], "sanity")
    refused(good[:-7], "layout mismatch")
    refused(good[:88] + bytes([9]) + good[89:], "order 9")
    bad_vocab = bytearray(good)
    off = 108 + 8 * 3
    off += -off % 8
    bad_vocab[off + 8 + 4] ^= 0xFF                      # corrupt one vocabulary hash entry (or an empty slot)
    p = tmp_path / "y.klm"
    p.write_bytes(bytes(bad_vocab))
    try:
        native.NativeLM(str(p))
    except native.DsmiError as e:
        assert "layout mismatch" in str(e)
    with pytest.raises(native.DsmiError):
        native.NativeLM(str(tmp_path / "missing.klm"))
