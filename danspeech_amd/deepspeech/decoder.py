"""Decoders of the drop-in surface (reference danspeech/deepspeech/decoder.py).

``GreedyDecoder`` and ``BeamCTCDecoder`` keep the reference's constructor arguments and the
``decode(probs, sizes=None) -> (strings[B][K], offsets[B][K])`` contract; the work is done by
the HIP kernels behind ``dsmi_greedy`` / ``dsmi_beam`` (danspeech_amd/csrc/decoder.hip).
"""
import numpy as np


def _edit_distance(a, b):
    """Levenshtein distance (the reference imports the ``Levenshtein`` package for this)."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class Decoder(object):
    """decoder.py:24-88."""

    def __init__(self, labels, blank_index=0):
        self.labels = labels
        self.int_to_char = dict([(i, c) for (i, c) in enumerate(labels)])
        self.blank_index = blank_index
        space_index = len(labels)          # out-of-range marker when there is no space label
        if ' ' in labels:
            space_index = labels.index(' ')
        self.space_index = space_index
        self._native = None
        # id -> code point table for the common case of one-character labels
        self._code_points = (np.array([ord(c) for c in labels], dtype="<u4")
                             if all(len(c) == 1 for c in labels) else None)

    def _to_string(self, ids):
        """Label ids (int array) -> text."""
        if self._code_points is not None:
            return self._code_points[ids].tobytes().decode("utf-32-le")
        return "".join(self.int_to_char[int(i)] for i in ids)

    def _dec(self, device_index, slot=0):
        """The native handle; ``slot`` > 0: a further handle with its own workspaces, for a caller that keeps several
        decodes in flight (one search per handle at a time)."""
        from .. import _native
        if self._native is None or self._native_device != device_index:
            if self._native is not None:
                self._native.close()
            for extra in getattr(self, "_native_slots", {}).values():
                extra.close()
            self._native_slots = {}
            self._native = _native.NativeDecoder(self.labels, blank_index=self.blank_index, device=device_index)
            self._native_device = device_index
            self._configure(self._native)
        if slot == 0:
            return self._native
        slots = self.__dict__.setdefault("_native_slots", {})
        if slot not in slots:
            slots[slot] = _native.NativeDecoder(self.labels, blank_index=self.blank_index, device=device_index)
            self._configure(slots[slot])
        return slots[slot]

    def _configure(self, native):
        pass

    @staticmethod
    def _on_gpu(probs):
        import torch
        probs = torch.as_tensor(probs, dtype=torch.float32)
        if not probs.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("decoding runs on the MI355X only (no CPU path) and no GPU is visible")
            probs = probs.cuda()
        return probs.contiguous()

    def wer(self, s1, s2):
        b = set(s1.split() + s2.split())
        word2char = dict(zip(b, range(len(b))))
        return _edit_distance([word2char[w] for w in s1.split()], [word2char[w] for w in s2.split()])

    def cer(self, s1, s2):
        return _edit_distance(s1.replace(' ', ''), s2.replace(' ', ''))

    def decode(self, probs, sizes=None):
        raise NotImplementedError


class GreedyDecoder(Decoder):
    """decoder.py:147-198: argmax per frame, collapse repeats, drop blanks; one path per utterance."""

    def __init__(self, labels, blank_index=0):
        super(GreedyDecoder, self).__init__(labels, blank_index)

    def decode(self, probs, sizes=None):
        import torch
        probs = self._on_gpu(probs)
        dec = self._dec(probs.device.index or 0)
        sz = None if sizes is None else np.asarray(torch.as_tensor(sizes).cpu()).astype(np.int32)
        res = dec.greedy(probs, sz)
        strings = [[self._to_string(ids)] for ids, _ in res]
        offsets = [[torch.from_numpy(off.astype(np.int32))] for _, off in res]
        return strings, offsets

    # the same in two halves for the batch pipeline (DanSpeechRecognizer.transcribe_batches): the kernel and the copies of its
    # results are queued right behind the forward, on the forward's own stream (on_lane: no side stream), `slot` = one native
    # handle per forward in flight
    on_lane = True

    def decode_enqueue(self, probs, sizes=None, slot=0):
        import torch
        probs = self._on_gpu(probs)
        dec = self._dec(probs.device.index or 0, slot)
        sz = None if sizes is None else np.asarray(torch.as_tensor(sizes).cpu()).astype(np.int32)
        if not hasattr(dec, "greedy_enqueue"):           # (a native handle without the split entry points: decode now)
            return ("decoded", dec.greedy(probs, sz))
        dec.greedy_enqueue(probs, sz)
        return dec

    def decode_collect(self, ticket):
        import torch
        res = ticket[1] if isinstance(ticket, tuple) else ticket.greedy_collect()
        strings = [[self._to_string(ids)] for ids, _ in res]
        offsets = [[torch.from_numpy(off.astype(np.int32))] for _, off in res]
        return strings, offsets


class BeamCTCDecoder(Decoder):
    """decoder.py:91-144.  ``lm_path`` may be an ARPA text model; ``num_processes`` is accepted for
    signature compatibility (the batch is decoded in parallel on the GPU, one workgroup per utterance)."""

    def __init__(self, labels, lm_path=None, alpha=0, beta=0, cutoff_top_n=40, cutoff_prob=1.0, beam_width=100,
                 num_processes=4, blank_index=0):
        super(BeamCTCDecoder, self).__init__(labels, blank_index)
        self.lm_path = lm_path
        self.alpha = alpha
        self.beta = beta
        self.cutoff_top_n = cutoff_top_n
        self.cutoff_prob = cutoff_prob
        self.beam_width = beam_width
        self.num_processes = num_processes
        self.last_scores = None

    def _configure(self, native):
        native.set_lm(self.lm_path, self.alpha, self.beta)

    def decode(self, probs, sizes=None):
        return self.decode_collect(self.decode_enqueue(probs, sizes))

    def decode_enqueue(self, probs, sizes=None, slot=0):
        """Launch the search on the current stream and return a ticket at once (the search is a kernel; ctcdecode's
        thread pool has no counterpart).  ``decode_collect(ticket)`` waits and builds the strings."""
        import torch
        probs = self._on_gpu(probs)
        dec = self._dec(probs.device.index or 0, slot)
        sz = None if sizes is None else np.asarray(torch.as_tensor(sizes).cpu()).astype(np.int32)
        dec.beam_enqueue(probs, sz, beam_width=self.beam_width, cutoff_top_n=self.cutoff_top_n, cutoff_prob=self.cutoff_prob)
        return dec

    def decode_collect(self, ticket):
        import torch
        tok, ts, ln, sc = ticket.beam_collect()
        self.last_scores = sc      # the reference drops ctcdecode's scores (decoder.py:140); kept for inspection
        # label strings and emission offsets of every beam, built in bulk: the valid prefixes of all beams are gathered
        # once (a few hundred thousand ids instead of B x beam x T), turned into ONE string, and cut by length
        B, W = tok.shape[0], tok.shape[1]
        lmax = int(ln.max()) if ln.size else 0
        valid = np.arange(lmax, dtype=np.int32)[None, None, :] < ln[:, :, None]
        flat_len = ln.reshape(-1).tolist()
        flat_ts = torch.split(torch.from_numpy(ts[:, :, :lmax][valid]), flat_len)
        ids = tok[:, :, :lmax][valid]
        cuts = np.concatenate(([0], np.cumsum(ln.reshape(-1), dtype=np.int64))).tolist()
        if self._code_points is not None:
            big = self._code_points[ids].tobytes().decode("utf-32-le")
            flat_str = [big[cuts[k]:cuts[k + 1]] for k in range(B * W)]
        else:
            flat_str = ["".join(self.int_to_char[int(i)] for i in ids[cuts[k]:cuts[k + 1]]) for k in range(B * W)]
        strings = [flat_str[b * W:(b + 1) * W] for b in range(B)]
        offsets = [list(flat_ts[b * W:(b + 1) * W]) for b in range(B)]
        return strings, offsets
