"""KenLM binary (.klm) reader of libdsmi.so (csrc/lm_klm.cpp.inc) on the CPU: header parsing, probing tables and the four
trie variants (plain, quantised, array-compressed pointers, both), every n-gram and back-off score against the ARPA text the binary was written from, refusal of what is not
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
        # build_binary -q / -b / -a: quantised values, array-compressed pointers, both (the common DeepSpeech-era recipe is
        # `trie -q 8 -a 255`); small -a values force several offset-array entries on files this small
        for mt, name, kw in ((klm.QUANT_TRIE, "quant", dict(quant_bits=(8, 8))), (klm.ARRAY_TRIE, "array", dict(array_bits=3)),
                             (klm.QUANT_ARRAY_TRIE, "quant_array", dict(quant_bits=(6, 4), array_bits=255)),
                             (klm.ARRAY_TRIE, "array255", dict(array_bits=255))):
            p = str(d / ("syn%d_%s.klm" % (order, name)))
            _, stored = klm.write_klm(arpa, p, mt, **kw)
            out[order][name] = p
            out[order][name + "_stored"] = stored
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


@pytest.mark.parametrize("order", [3, 5])
@pytest.mark.parametrize("name", ["quant", "array", "quant_array", "array255"])
def test_quantised_and_array_compressed_tries(native, files, order, name):
    """KenLM model types 3, 4, 5.  Every n-gram is found with the value the file stores for it: the exact floats where only
    the pointers are compressed, the bin centres where the values are quantised (unigrams are never quantised)."""
    lm = native.NativeLM(files[order][name])
    plain = native.NativeLM(files[order]["trie"])
    assert lm.kind == "klm-trie" and lm.order == order and lm.vocab_size == plain.vocab_size
    stored = files[order][name + "_stored"]
    quantised = name.startswith("quant")
    n_changed = 0
    for n in range(1, order + 1):
        for g, (lp, bo) in stored[n].items():
            ids = [lm.word_index(w) for w in g]
            assert ids == [plain.word_index(w) for w in g]
            got = lm.lookup(ids)
            assert got is not None, g
            assert got[0] == lp and got[1] == bo, (g, got, lp, bo)
            n_changed += got != plain.lookup(ids)
    assert (n_changed > 0) == quantised            # quantisation moves values; pointer compression must not
    if not quantised:
        rng = np.random.default_rng(2)
        vocab = [g[0] for g in stored[1]]
        for _ in range(300):
            k = int(rng.integers(1, order + 1))
            ws = [vocab[int(i)] for i in rng.integers(0, len(vocab), size=k)]
            assert lm.cond_log10([lm.word_index(w) for w in ws]) == plain.cond_log10([plain.word_index(w) for w in ws])
    lm.close(); plain.close()


def test_bin_encoding_rules():
    c = np.array([-3.0, -2.0, -1.0, -0.5], dtype=np.float32)
    assert [klm.encode_bin(c, v) for v in (-9.0, -3.0, -2.6, -2.5, -2.4, -0.7, 0.0)] == [0, 0, 0, 1, 1, 3, 3]
    assert klm.encode_bin(np.array([-0.0, 0.0, -1.0, -0.2], dtype=np.float32), -0.9, 2) == 2     # never the reserved bins
    assert list(klm.make_bins([1.0, 2.0, 3.0, 4.0], 2)) == [1.5, 3.5]
    assert klm.chop_bits(1000, 5000, 0) == 0 and klm.chop_bits(1000, 5000, 64) >= 1


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
    refused(good[:96] + struct.pack("<i", 1) + good[100:], "rest-cost probing")
    refused(good[:96] + struct.pack("<i", 7) + good[100:], "unknown")
    for mt in (3, 4, 5):                     # a probing image under a trie model type: refused somewhere in the layout checks
        p = tmp_path / "z.klm"
        p.write_bytes(good[:96] + struct.pack("<i", mt) + good[100:])
        with pytest.raises(native.DsmiError):
            native.NativeLM(str(p))
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
