"""GPU parity: dsmi_beam (HIP prefix beam search + n-gram scorer) against oracle/beam.py.
Parity with ctcdecode itself is unpinned (third-party, absent); both sides restate the same
published algorithm and carry float64, so strings/timesteps must be identical and scores agree
to 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest

from danspeech_amd import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def native():
    from danspeech_amd import _native
    assert torch.cuda.is_available()
    _native.lib()
    return _native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _peaky_probs(rng, B, T, C, sharp=3.0):
    logits = rng.standard_normal((B, T, C)) * sharp
    logits[:, :, 0] += 1.5        # blanks dominate like a trained CTC model
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def _compare(native, probs, sizes, labels, beam, lm_path=None, alpha=0.0, beta=0.0, cutoff_top_n=40, cutoff_prob=1.0,
             n_check=None):
    from oracle import beam as ob
    dec = native.NativeDecoder(labels, blank_index=0)
    dec.set_lm(lm_path, alpha, beta)
    tok, ts, ln, sc = dec.beam(_dev(probs), sizes, beam_width=beam, cutoff_top_n=cutoff_top_n, cutoff_prob=cutoff_prob)
    strings, offsets, scores = ob.beam_decode(probs.astype(np.float64), sizes, labels, beam, lm_path=lm_path, alpha=alpha,
                                              beta=beta, cutoff_top_n=cutoff_top_n, cutoff_prob=cutoff_prob)
    B = probs.shape[0]
    for b in range(B):
        nref = len([s for s, o in zip(strings[b], offsets[b]) if True])
        k = n_check or beam
        for p in range(min(k, nref)):
            if p >= len(scores[b]) or (strings[b][p] == "" and scores[b][p] == 0.0 and ln[b, p] == 0 and p > 0):
                continue
            got = "".join(labels[i] for i in tok[b, p, :ln[b, p]])
            if np.isinf(scores[b][p]):
                continue
            assert got == strings[b][p], (b, p, got, strings[b][p])
            assert list(ts[b, p, :ln[b, p]]) == offsets[b][p], (b, p)
            # 1e-4 (north star) + half a float32 ulp of the score: the ABI returns float32, like ctcdecode's FloatTensor
            assert abs(float(sc[b, p]) - scores[b][p]) < 1e-4 + 0.5 * float(np.spacing(np.float32(abs(scores[b][p])))), (b, p, sc[b, p], scores[b][p])
    dec.close()


def test_beam_no_lm_small_alphabet_exhaustive(native):
    rng = np.random.default_rng(0)
    probs = rng.dirichlet(np.ones(4), size=(3, 6)).astype(np.float32)
    _compare(native, probs, None, "_ab ", beam=64)


def test_beam_dormant_prefixes_come_back(native):
    """Flat noisy rows and narrow beams: prefixes leave the beam while their extensions stay and come back later.  The
    kernel then re-hangs the entries below them by a walk through its node pool -- the one slow path of its frame loop
    (oracle/beam_flat.py is the same formulation on the CPU; tests/test_oracle_beam_flat.py shows these inputs take it)."""
    probs = np.stack([np.random.default_rng(seed).dirichlet(np.ones(4) * 0.5, size=60) for seed in (80, 61, 112, 119, 165, 192)]).astype(np.float32)
    revivals = hops = 0
    for beam in (3, 4, 6):
        dec = native.NativeDecoder("_abc", blank_index=0)
        dec.beam(_dev(probs), None, beam_width=beam)
        st = dec.beam_stats()
        revivals += st["revivals"]; hops += st["walk_hops"]
        dec.close()
        _compare(native, probs, None, "_abc", beam=beam)
    assert revivals > 0 and hops > 0


def test_beam_exact_ties_and_flat_rows(native):
    """Uniform rows: every candidate of a frame ties in the selection histogram's threshold bin (the exact ranking of a bin
    that is larger than the on-chip list) and scores tie exactly; zero rows (log(FLT_MIN)) pile up in the last bin."""
    labels = syn.DANSPEECH_LABELS
    C = len(labels)
    probs = np.full((2, 12, C), 1.0 / C, dtype=np.float32)
    probs[1, 3:6, 5:] = 0.0
    probs[1, 3:6, :5] = 0.2
    dec = native.NativeDecoder(labels, blank_index=0)
    tok, ts, ln, sc = dec.beam(_dev(probs), None, beam_width=24)
    st = dec.beam_stats()
    dec.close()
    assert st["list_rankings"] + st["full_rankings"] > 0
    # which of several exactly tied candidates stays depends on the last bit of exp / log (libm in the oracle, ocml on the GPU):
    # the SCORES of the beams are determined, the identities among tied beams are not
    from oracle import beam_flat as bf
    for b in range(2):
        res = bf.ctc_beam_search(probs[b].astype(np.float64), labels, 24)
        want = np.sort(np.array([r[0] for r in res], dtype=np.float64))
        got = np.sort(sc[b].astype(np.float64))
        assert np.abs(want - got).max() < 1e-4
        for p in range(24):
            assert 0 <= ln[b, p] <= 12 and (np.diff(ts[b, p, :ln[b, p]]) > 0).all()


def test_beam_no_lm_danspeech_labels(native):
    rng = np.random.default_rng(1)
    labels = syn.DANSPEECH_LABELS
    probs = _peaky_probs(rng, 4, 60, len(labels))
    _compare(native, probs, np.array([60, 45, 33, 7], dtype=np.int32), labels, beam=16)


def test_beam_with_lm(native, tmp_path):
    labels = syn.DANSPEECH_LABELS
    path = str(tmp_path / "toy3.arpa")
    syn.make_arpa(path, order=3, n_words=200, seed=5, ngrams_per_order=600)
    rng = np.random.default_rng(2)
    probs = _peaky_probs(rng, 3, 80, len(labels), sharp=2.0)
    # DanSpeechRecognizer defaults: alpha=1.3, beta=0.2, cutoff_top_n=40, cutoff_prob=1.0 (DanSpeechRecognizer.py:16-17,89-92)
    _compare(native, probs, np.array([80, 64, 20], dtype=np.int32), labels, beam=32, lm_path=path, alpha=1.3, beta=0.2)


def test_beam_with_lm_wide_beam_5gram(native, tmp_path):
    labels = syn.DANSPEECH_LABELS
    path = str(tmp_path / "toy5.arpa")
    syn.make_arpa(path, order=5, n_words=300, seed=6, ngrams_per_order=500)
    rng = np.random.default_rng(3)
    probs = _peaky_probs(rng, 2, 50, len(labels), sharp=1.5)
    _compare(native, probs, None, labels, beam=128, lm_path=path, alpha=1.2, beta=0.15, n_check=40)


def test_beam_wide_beams_and_capacity(native, tmp_path):
    """Beams beyond 128 (more pairs per thread: the kernel's largest instantiation) against the oracle, and the documented
    refusal where the on-chip buffers end."""
    labels = syn.DANSPEECH_LABELS
    path = str(tmp_path / "toy3.arpa")
    syn.make_arpa(path, order=3, n_words=200, seed=5, ngrams_per_order=600)
    rng = np.random.default_rng(8)
    probs = _peaky_probs(rng, 2, 30, len(labels), sharp=1.0)
    _compare(native, probs, np.array([30, 19], dtype=np.int32), labels, beam=200, lm_path=path, alpha=1.3, beta=0.2, n_check=30)
    _compare(native, probs, None, labels, beam=160, n_check=30)
    dec = native.NativeDecoder(labels, blank_index=0)
    with pytest.raises(native.DsmiError) as e:
        dec.beam(_dev(probs), None, beam_width=400)
    assert e.value.code == native.DSMI_ERR_CAPACITY
    dec.close()


def test_beam_cutoff_top_n(native):
    labels = syn.DANSPEECH_LABELS
    rng = np.random.default_rng(4)
    probs = _peaky_probs(rng, 2, 40, len(labels))
    _compare(native, probs, None, labels, beam=20, cutoff_top_n=10)


def test_beam_cutoff_prob_with_lm_and_long_utterance(native, tmp_path):
    """cutoff_prob < 1 (cumulative-probability vocabulary pruning, decoder_utils get_pruned_log_probs) together
    with the scorer, on 300 frames: many radix-select passes, node reuse and dictionary pruning in one run."""
    labels = syn.DANSPEECH_LABELS
    path = str(tmp_path / "lm3.arpa")
    syn.make_arpa(path, order=3, n_words=400, seed=21, ngrams_per_order=1500)
    rng = np.random.default_rng(7)
    probs = _peaky_probs(rng, 2, 300, len(labels))
    _compare(native, probs, np.array([300, 171], dtype=np.int32), labels, beam=48, lm_path=path, alpha=1.1, beta=0.3,
             cutoff_top_n=15, cutoff_prob=0.98, n_check=24)


@pytest.mark.parametrize("order,beam", [(3, 64), (5, 128)])
def test_beam_with_kenlm_binaries_equals_arpa(native, tmp_path, order, beam):
    """The same model as ARPA text, as a KenLM probing binary (looked up in place on the GPU with KenLM's own hash
    chain) and as a KenLM trie binary (converted at load time): identical beams, timesteps and scores.  Binaries from
    oracle/klm.py's writer (parity with KenLM's own files is unpinned, tests/test_klm.py)."""
    from oracle import klm
    labels = syn.DANSPEECH_LABELS
    arpa = str(tmp_path / "lm.arpa")
    syn.make_arpa(arpa, order=order, n_words=400, seed=30 + order, ngrams_per_order=1500)
    rng = np.random.default_rng(11 + order)
    probs = _dev(_peaky_probs(rng, 3, 120, len(labels), sharp=2.0))
    sizes = np.array([120, 97, 40], dtype=np.int32)
    outs = {}
    for name, mt in (("arpa", None), ("probing", klm.PROBING), ("trie", klm.TRIE)):
        path = arpa
        if mt is not None:
            path = str(tmp_path / ("lm_%s.klm" % name))
            klm.write_klm(arpa, path, mt)
        dec = native.NativeDecoder(labels, blank_index=0)
        dec.set_lm(path, 1.3, 0.2)
        outs[name] = dec.beam(probs, sizes, beam_width=beam)
        dec.close()
    # and the oracle on the ARPA text
    _compare(native, probs.cpu().numpy(), sizes, labels, beam=beam, lm_path=arpa, alpha=1.3, beta=0.2, n_check=20)
    for name in ("probing", "trie"):
        for a, b in zip(outs["arpa"], outs[name]):
            assert np.array_equal(a, b), name
    assert (outs["arpa"][2][:, 0] > 3).all()


def test_lm_file_errors(native, tmp_path):
    dec = native.NativeDecoder(syn.DANSPEECH_LABELS, blank_index=0)
    with pytest.raises(native.DsmiError):
        dec.set_lm(str(tmp_path / "missing.arpa"), 1.0, 0.1)
    bad = tmp_path / "bad.klm"
    bad.write_bytes(b"mmap lm http://kheafield.com/code format version 5\\n\\x00\\x01")
    with pytest.raises(native.DsmiError):
        dec.set_lm(str(bad), 1.0, 0.1)
    dec.close()


def test_beam_enqueue_collect_halves(native):
    """dsmi_beam_enqueue / dsmi_beam_collect: the split the pipelined recogniser uses (two decoder handles, two searches in
    flight, results collected later) equals dsmi_beam; one search per handle at a time."""
    rng = np.random.default_rng(5)
    labels = syn.DANSPEECH_LABELS
    pa, pb = _peaky_probs(rng, 3, 60, len(labels)), _peaky_probs(rng, 2, 45, len(labels))
    da, db = native.NativeDecoder(labels, blank_index=0), native.NativeDecoder(labels, blank_index=0)
    want_a, want_b = da.beam(_dev(pa), None, beam_width=16), db.beam(_dev(pb), None, beam_width=16)
    xa, xb = _dev(pa), _dev(pb)
    da.beam_enqueue(xa, None, beam_width=16)
    db.beam_enqueue(xb, None, beam_width=16)
    with pytest.raises(native.DsmiError) as e:
        da.beam_enqueue(xa, None, beam_width=16)
    assert "not been collected" in e.value.msg
    got_b, got_a = db.beam_collect(), da.beam_collect()
    for got, want in ((got_a, want_a), (got_b, want_b)):
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    with pytest.raises(native.DsmiError):
        lib = native.lib()
        import ctypes
        z = np.zeros(4, dtype=np.int32)
        da._check(lib.dsmi_beam_collect(da._h, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data))     # nothing enqueued
    da.close(); db.close()
