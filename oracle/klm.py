"""KenLM binary (``.klm``) files, restated in Python: a writer (ARPA -> probing / trie binary) and a reader.

TEST INFRASTRUCTURE (see oracle/__init__.py).  KenLM is a third-party dependency of ctcdecode and is absent from
the reference tree and from this image, and no ``.klm`` file is available offline (the reference only names URLs:
danspeech/language_models/dsl_3gram.py:4,16-20): **PARITY WITH KenLM IS UNPINNED**.  What is restated here is the
published on-disk layout of KenLM's ``build_binary`` (format version 5; lm/binary_format.{hh,cc}, lm/vocab.{hh,cc},
lm/search_hashed.hh, lm/search_trie.hh, lm/trie.{hh,cc}, lm/bhiksha.hh, lm/quantize.hh, util/probing_hash_table.hh,
util/bit_packing.hh, util/murmur_hash.cc), from knowledge of that code:

    [Sanity 88 B][FixedWidthParameters 20 B][counts: order x u64]  -> padded to 8           (header)
    [vocabulary lookup][search tables]                                                      (one mapped block)
    [vocabulary strings, NUL-terminated, in word-index order]                               (if has_vocabulary)

  probing (model_type 0): vocabulary = {u32 version, u32 bound} + open-addressing table of {u64 hash, u32 id} (12 B,
      ``max(V + 1, multiplier * V)`` buckets, slot = hash % buckets, linear probing, key 0 = empty);
      unigrams = (V + 1) x {f32 prob, f32 backoff} indexed by word id;
      orders 2..N-1: tables of {u64 key, f32 prob, f32 backoff}; order N: {u64 key, f32 prob} (12 B);
      key(w1..wn) = fold over wn, wn-1, .., w1 of  h = (h * 8978948897894561157) ^ ((1 + w) * 17894857484156487943),
      starting from h = wn;  the SIGN BIT of a stored prob is a flag ("independent left"): log10 p = -|stored|.
  trie (model_type 2, no quantisation, pointers inline): vocabulary = {u64 n} + n sorted u64 hashes (word id = rank + 1,
      <unk> = 0); unigrams = (V + 2) x {f32 prob, f32 backoff, u64 next}; orders 2..N-1: bit-packed records
      [word : bits(V)] [prob : 31, sign implied] [backoff : 32] [next : bits(count of the next order)], sorted by word
      inside a parent's range, one extra record at the end for the last ``next``; order N: [word][prob : 31];
      the trie is keyed by the n-gram REVERSED (last word first).
  quantised tries (model_type 3 and 5, ``build_binary -q P -b B``; lm/quantize.hh SeparatelyQuantize): the search memory
      starts with {u8 version = 2, u8 P, u8 B, 5 pad}, then for every middle order a table of 2^P float bin centres for the
      probabilities and one of 2^B for the back-offs (its first two bins are the reserved -0.0 "does not extend" and +0.0),
      then the longest order's 2^P table; unigrams stay floats; a record's [prob 31][backoff 32] becomes one field of B + P
      bits, back-off index in the LOW bits; centres = means of equal-population slices of the sorted values (MakeBins), a
      value is stored as the nearer of its two neighbouring centres (Bins::Encode);
  array-compressed pointers (model_type 4 and 5, ``-a A``; lm/bhiksha.hh ArrayBhiksha): every middle order is preceded by
      {u8 version = 0, u8 A}, padding to the next 8-byte boundary, an 8-byte header word and an array of u64 offsets:
      entry e = index of the first record whose ``next >> inline_bits`` is >= e (entry 0 = 0); the records keep only the low
      ``inline_bits = bits(max_next) - chop`` bits of ``next``, with chop = argmin over 0..min(bits, A) of
      (max_next >> (bits - chop)) * 64 - records * chop; the block is sized 8 * (1 + entries) + 7 bytes.
  word hash = MurmurHash64A(bytes, seed 0).

The reader below and the C++ reader (danspeech_amd/csrc/lm_klm.cpp.inc) are held to this writer (round trip against
the ARPA text) -- a self-consistency check of the restatement, not evidence about real KenLM output.
"""
import struct

import numpy as np

MAGIC = b"mmap lm http://kheafield.com/code format version 5\n\x00"
PROBING, TRIE = 0, 2
QUANT_TRIE, ARRAY_TRIE, QUANT_ARRAY_TRIE = 3, 4, 5
M64 = (1 << 64) - 1


def murmur64a(data, seed=0):
    m, r = 0xc6a4a7935bd1e995, 47
    n = len(data)
    h = (seed ^ (n * m)) & M64
    for i in range(0, n - n % 8, 8):
        k = int.from_bytes(data[i:i + 8], "little")
        k = (k * m) & M64
        k ^= k >> r
        k = (k * m) & M64
        h ^= k
        h = (h * m) & M64
    tail = data[n - n % 8:]
    if tail:
        h ^= int.from_bytes(tail, "little")
        h = (h * m) & M64
    h ^= h >> r
    h = (h * m) & M64
    h ^= h >> r
    return h


def combine(h, w):
    return ((h * 8978948897894561157) ^ ((1 + w) * 17894857484156487943)) & M64


def ngram_key(ids):
    """Probing key of the n-gram with word ids ``ids`` (natural order), n >= 2."""
    h = ids[-1]
    for w in reversed(ids[:-1]):
        h = combine(h, w)
    return h


def required_bits(x):
    return int(x).bit_length()


def buckets_for(entries, multiplier):
    return max(entries + 1, int(np.float32(multiplier) * np.float32(entries)))


def read_arpa(path):
    """-> (order, grams) with grams[n] = list of (words tuple, log10 prob, log10 backoff)."""
    grams, section, order = {}, 0, 0
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line or line == "\\data\\" or line.startswith("ngram "):
                continue
            if line == "\\end\\":
                break
            if line.startswith("\\"):
                section = int(line[1:line.index("-")])
                order = max(order, section)
                grams[section] = []
                continue
            tok = line.split()
            words = tuple(tok[1:1 + section])
            bo = float(tok[1 + section]) if len(tok) > 1 + section else 0.0
            grams[section].append((words, float(tok[0]), bo))
    return order, grams


def _header(order, counts, model_type, multiplier=1.5, has_vocab=True):
    sanity = MAGIC + b"\x00" * (56 - len(MAGIC) - 0)
    sanity = sanity[:56]
    sanity += struct.pack("<fff", 0.0, 1.0, -0.5)
    sanity += struct.pack("<III", 1, 0xFFFFFFFF, 0)
    sanity += struct.pack("<Q", 1)
    assert len(sanity) == 88
    fixed = struct.pack("<B3xfIB3xI", order, multiplier, model_type, 1 if has_vocab else 0, 0 if model_type == PROBING else 1)
    assert len(fixed) == 20
    head = sanity + fixed + b"".join(struct.pack("<Q", c) for c in counts)
    return head + b"\x00" * (-len(head) % 8)


def _f32(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def make_bins(values, bins):
    """quantize.cc MakeBins: centres of ``bins`` equal-population slices of the sorted float32 values."""
    v = np.sort(np.asarray(values, dtype=np.float32))
    out = []
    start = 0
    for i in range(bins):
        finish = (len(v) * (i + 1)) // bins
        if finish == start:
            out.append(out[-1] if i else np.float32(-np.inf))
        else:
            out.append(np.float32(np.sum(v[start:finish], dtype=np.float64) / np.float32(finish - start)))
        start = finish
    return np.asarray(out, dtype=np.float32)


def encode_bin(centres, value, reserved=0):
    """quantize.hh Bins::Encode: index of the nearer neighbouring centre (lower_bound, ties go up)."""
    value = np.float32(value)
    above = reserved + int(np.searchsorted(centres[reserved:], value, side="left"))
    if above == reserved:
        return reserved
    if above == len(centres):
        return len(centres) - 1
    return above - int(value - centres[above - 1] < centres[above] - value)


def chop_bits(max_offset, max_next, configured):
    required = required_bits(max_next)
    best, lowest = 0, None
    for chop in range(0, min(required, configured) + 1):
        change = (max_next >> (required - chop)) * 64 - max_offset * chop
        if lowest is None or change < lowest:
            lowest, best = change, chop
    return best


def write_klm(arpa_path, out_path, model_type=PROBING, multiplier=1.5, flag_some_signs=True, quant_bits=(8, 8), array_bits=64):
    """ARPA text -> KenLM binary (probing, or a trie: plain, quantised, array-compressed or both).  ``flag_some_signs`` clears
    the sign bit of every third stored prob in the probing tables, as KenLM does for n-grams that extend left: a reader must
    take -|prob|.  For the quantised variants the function returns, beside the word ids, what a reader must find:
    ``stored[n][words] = (prob, backoff)`` after quantisation."""
    quantised, arrayed = model_type in (QUANT_TRIE, QUANT_ARRAY_TRIE), model_type in (ARRAY_TRIE, QUANT_ARRAY_TRIE)
    order, grams = read_arpa(arpa_path)
    words = [g[0][0] for g in grams[1]]
    if "<unk>" not in words:
        grams[1].insert(0, (("<unk>",), -100.0, 0.0))
        words.insert(0, "<unk>")
    V = len(words)
    counts = [len(grams[n]) for n in range(1, order + 1)]
    hashes = {w: murmur64a(w.encode("utf-8")) for w in words}
    if model_type == PROBING:
        ids = {"<unk>": 0}
        for w in words:
            if w != "<unk>":
                ids[w] = len(ids)
        vb = buckets_for(V, multiplier)
        table = [(0, 0)] * vb
        for w, i in ids.items():
            if w == "<unk>":
                continue                      # <unk> is id 0 and not in the lookup table (a miss returns 0)
            s = hashes[w] % vb
            while table[s][0]:
                s = (s + 1) % vb
            table[s] = (hashes[w], i)
        vocab = struct.pack("<II", 0, V) + b"".join(struct.pack("<QI", k, v) for k, v in table)
    else:
        rest = sorted((hashes[w], w) for w in words if w != "<unk>")
        ids = {"<unk>": 0}
        for rank, (_, w) in enumerate(rest):
            ids[w] = rank + 1
        vocab = struct.pack("<Q", len(rest)) + b"".join(struct.pack("<Q", h) for h, _ in rest)
        vocab += b"\x00" * (8 + 8 * V - len(vocab))       # sized for V entries; <unk> has no slot
    by_id = sorted(ids, key=ids.get)
    uni = {ids[g[0][0]]: g for g in grams[1]}

    if model_type == PROBING:
        search = b""
        for i in range(V):
            _, lp, bo = uni[i]
            search += struct.pack("<ff", lp, bo)
        search += struct.pack("<ff", 0.0, 0.0)
        n_flag = 0
        for n in range(2, order + 1):
            nb = buckets_for(counts[n - 1], multiplier)
            tab = [None] * nb
            for g, lp, bo in grams[n]:
                key = ngram_key([ids[w] for w in g])
                n_flag += 1
                stored = abs(lp) if (flag_some_signs and n_flag % 3 == 0) else lp
                s = key % nb
                while tab[s] is not None:
                    s = (s + 1) % nb
                tab[s] = (key, stored, bo)
            for e in tab:
                k, lp, bo = e if e is not None else (0, 0.0, 0.0)
                search += struct.pack("<Qff", k, lp, bo) if n < order else struct.pack("<Qf", k, lp)
    else:
        # reversed trie: children of a context node are the words that PRECEDE it
        rev = [dict() for _ in range(order + 1)]       # rev[n][reversed id tuple] = (lp, bo)
        for n in range(1, order + 1):
            for g, lp, bo in grams[n]:
                rev[n][tuple(reversed([ids[w] for w in g]))] = (lp, bo)
        level = [None, [(i,) for i in range(V)]]        # level[n] = records of order n in trie order
        for n in range(2, order + 1):
            children = {}
            for key in rev[n]:
                children.setdefault(key[:-1], []).append(key)
            recs = []
            for parent in level[n - 1]:
                recs.extend(sorted(children.get(parent, [])))
            assert len(recs) == len(rev[n]), "an n-gram's suffix context is missing from the ARPA file"
            level.append(recs)
        first_child = [None] * (order + 1)              # first_child[n][i] = index of the first order-(n+1) record under record i
        for n in range(1, order):
            pos, fc, nxt = 0, [], level[n + 1]
            for parent in level[n]:
                fc.append(pos)
                while pos < len(nxt) and nxt[pos][:-1] == parent:
                    pos += 1
            fc.append(pos)
            first_child[n] = fc
        pbits, bbits = quant_bits
        tables = {}                      # order -> (prob centres, backoff centres or None)
        search = b""
        if quantised:
            search += struct.pack("<BBB5x", 2, pbits, bbits)
            for n in range(2, order + 1):
                pc = make_bins([lp for lp, _ in rev[n].values()], 1 << pbits)
                bc = None
                if n < order:
                    bc = np.concatenate((np.array([-0.0, 0.0], dtype=np.float32),
                                         make_bins([bo for _, bo in rev[n].values() if bo != 0.0], (1 << bbits) - 2)))
                tables[n] = (pc, bc)
                search += pc.astype("<f4").tobytes() + (bc.astype("<f4").tobytes() if bc is not None else b"")
        for i in range(V):
            lp, bo = rev[1][(i,)]
            search += struct.pack("<ffQ", lp, bo, first_child[1][i] if order > 1 else 0)
        search += struct.pack("<ffQ", 0.0, 0.0, first_child[1][V] if order > 1 else 0)
        search += struct.pack("<ffQ", 0.0, 0.0, 0)
        wbits = required_bits(V)
        stored = {1: {tuple(g): (np.float32(lp), np.float32(bo)) for g, lp, bo in grams[1]}}
        back = {i: w for w, i in ids.items()}
        for n in range(2, order + 1):
            last = n == order
            recs = level[n]
            qbits = (pbits if last else pbits + bbits) if quantised else (31 if last else 63)
            nbits = 0 if last else required_bits(counts[n])
            prefix = b""
            if arrayed and not last:
                base = len(_header(order, counts, model_type, multiplier)) + len(vocab) + len(search)      # absolute file offset
                chop = chop_bits(len(recs) + 1, counts[n], array_bits)
                nbits = required_bits(counts[n]) - chop
                n_arr = (counts[n] >> nbits) + 1
                offsets, e = [0] * n_arr, 1
                for idx in range(len(recs) + 1):
                    hi = first_child[n][idx] >> nbits
                    while e <= hi:
                        offsets[e] = idx
                        e += 1
                assert e == n_arr, "did not get all the array entries that were expected"
                blob = bytearray(8 * (1 + n_arr) + 7)
                blob[0], blob[1] = 0, min(array_bits, 255)
                at = (-base) % 8 + 8
                blob[at:at + 8 * n_arr] = b"".join(struct.pack("<Q", o) for o in offsets)
                prefix = bytes(blob)
            total = wbits + qbits + nbits
            nbytes = ((1 + len(recs)) * total + 7) // 8 + 8
            buf = bytearray(nbytes)
            stored[n] = {}
            for idx in range(len(recs) + 1):
                rec = 0
                if idx < len(recs):
                    lp, bo = rev[n][recs[idx]]
                    words = tuple(back[i] for i in reversed(recs[idx]))
                    if quantised:
                        pc, bc = tables[n]
                        qp = encode_bin(pc, lp)
                        field, sp, sb = qp, pc[qp], np.float32(0.0)
                        if not last:
                            qb = (1 if bo == 0.0 else encode_bin(bc, bo, 2))     # 0.0 -> the "extends" reserved bin
                            field = (qp << bbits) | qb
                            sb = bc[qb]
                        stored[n][words] = (np.float32(sp), np.float32(sb) + np.float32(0.0))
                        rec = recs[idx][-1] | (field << wbits)
                    else:
                        stored[n][words] = (np.float32(lp), np.float32(bo if not last else 0.0))
                        rec = recs[idx][-1] | ((_f32(lp) & 0x7FFFFFFF) << wbits)
                        if not last:
                            rec |= _f32(bo) << (wbits + 31)
                if not last:
                    rec |= (first_child[n][idx] & ((1 << nbits) - 1)) << (wbits + qbits)
                pos = idx * total
                byte, shift = pos >> 3, pos & 7
                nb = (total + shift + 7) // 8
                chunk = int.from_bytes(buf[byte:byte + nb], "little") | (rec << shift)
                buf[byte:byte + nb] = chunk.to_bytes(nb, "little")
            search += prefix + bytes(buf)
    strings = b"".join(w.encode("utf-8") + b"\x00" for w in by_id)
    with open(out_path, "wb") as f:
        f.write(_header(order, counts, model_type, multiplier) + vocab + search + strings)
    if model_type in (QUANT_TRIE, ARRAY_TRIE, QUANT_ARRAY_TRIE):
        return ids, stored
    return ids


class KlmReader(object):
    """Reads what ``write_klm`` (and, if the restatement is right, KenLM's build_binary) writes."""

    def __init__(self, path):
        d = open(path, "rb").read()
        if d[:len(MAGIC)] != MAGIC:
            raise ValueError("not a KenLM binary of format version 5")
        zero, one, mhalf = struct.unpack_from("<fff", d, 56)
        w1, wmax, _pad = struct.unpack_from("<III", d, 68)
        (u1,) = struct.unpack_from("<Q", d, 80)
        if (zero, one, mhalf, w1, wmax, u1) != (0.0, 1.0, -0.5, 1, 0xFFFFFFFF, 1):
            raise ValueError("sanity block mismatch (other endianness or type sizes)")
        self.order, self.multiplier, self.model_type, has_vocab, _ver = struct.unpack_from("<B3xfIB3xI", d, 88)
        self.counts = list(struct.unpack_from("<%dQ" % self.order, d, 108))
        off = 108 + 8 * self.order
        off += -off % 8
        V = self.counts[0]
        self.d = d
        if self.model_type == PROBING:
            self.vb = buckets_for(V, self.multiplier)
            self.vocab_off = off + 8
            off += 8 + 12 * self.vb
            self.uni_off = off
            off += 8 * (V + 1)
            self.tabs = []
            for n in range(2, self.order + 1):
                nb = buckets_for(self.counts[n - 1], self.multiplier)
                es = 16 if n < self.order else 12
                self.tabs.append((off, nb, es))
                off += nb * es
        elif self.model_type == TRIE:
            (n_sorted,) = struct.unpack_from("<Q", d, off)
            self.sorted_hashes = np.frombuffer(d, dtype="<u8", count=n_sorted, offset=off + 8)
            off += 8 + 8 * V
            self.uni_off = off
            off += 16 * (V + 2)
            self.wbits = required_bits(V)
            self.levels = []
            for n in range(2, self.order + 1):
                last = n == self.order
                nbits = 0 if last else required_bits(self.counts[n])
                total = self.wbits + (31 if last else 63) + nbits
                size = ((1 + self.counts[n - 1]) * total + 7) // 8 + 8
                self.levels.append((off, total, nbits, last))
                off += size
        else:
            raise ValueError("unsupported KenLM model type %d" % self.model_type)
        self.strings_off = off
        if not has_vocab:
            raise ValueError("binary without vocabulary strings")
        self.words = d[off:].split(b"\x00")[:-1]
        if len(self.words) != V:
            raise ValueError("vocabulary strings do not match the unigram count")
        self.words = [w.decode("utf-8") for w in self.words]
        self.ids = {w: i for i, w in enumerate(self.words)}

    def index(self, word):
        """Word id through the file's own lookup structure (not the string list)."""
        h = murmur64a(word.encode("utf-8"))
        if self.model_type == PROBING:
            s = h % self.vb
            while True:
                k, v = struct.unpack_from("<QI", self.d, self.vocab_off + 12 * s)
                if k == h:
                    return v
                if k == 0:
                    return 0
                s = (s + 1) % self.vb
        i = int(np.searchsorted(self.sorted_hashes, np.uint64(h)))
        return i + 1 if i < len(self.sorted_hashes) and int(self.sorted_hashes[i]) == h else 0

    def _bits(self, base, bit, length):
        byte = base + (bit >> 3)
        return (int.from_bytes(self.d[byte:byte + 16], "little") >> (bit & 7)) & ((1 << length) - 1)

    def lookup(self, ids):
        """(log10 prob, log10 backoff) of the n-gram with word ids ``ids`` (natural order), or None."""
        n = len(ids)
        if self.model_type == PROBING:
            if n == 1:
                lp, bo = struct.unpack_from("<ff", self.d, self.uni_off + 8 * ids[0])
                return -abs(lp), bo
            off, nb, es = self.tabs[n - 2]
            key = ngram_key(list(ids))
            s = key % nb
            while True:
                (k,) = struct.unpack_from("<Q", self.d, off + es * s)
                if k == key:
                    lp = struct.unpack_from("<f", self.d, off + es * s + 8)[0]
                    bo = struct.unpack_from("<f", self.d, off + es * s + 12)[0] if es == 16 else 0.0
                    return -abs(lp), bo
                if k == 0:
                    return None
                s = (s + 1) % nb
        rev = list(reversed(ids))
        lp, bo, begin = struct.unpack_from("<ffQ", self.d, self.uni_off + 16 * rev[0])
        (end,) = struct.unpack_from("<Q", self.d, self.uni_off + 16 * (rev[0] + 1) + 8)
        for depth, w in enumerate(rev[1:]):
            off, total, nbits, last = self.levels[depth]
            lo, hi, found = begin, end, -1
            while lo < hi:
                mid = (lo + hi) // 2
                v = self._bits(off, mid * total, self.wbits)
                if v < w:
                    lo = mid + 1
                elif v > w:
                    hi = mid
                else:
                    found = mid
                    break
            if found < 0:
                return None
            bit = found * total + self.wbits
            lp = -abs(struct.unpack("<f", struct.pack("<I", self._bits(off, bit, 31) | 0x80000000))[0])
            bo = 0.0
            if not last:
                bo = struct.unpack("<f", struct.pack("<I", self._bits(off, bit + 31, 32)))[0]
                begin = self._bits(off, bit + 63, nbits)
                end = self._bits(off, bit + 63 + total, nbits)
        return lp, bo
