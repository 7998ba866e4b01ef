"""A deterministic stand-in for ``ctcdecode.CTCBeamDecoder.decode`` (third-party, absent): the SAME function runs under
the reference when tools/gen_golden_surface.py records G11 and under this package when the CPU test replays it, so every
difference in what comes back is plumbing (argument order, [0][0] vs [0], string building), not search arithmetic."""
import numpy as np


def fake_beams(probs, sizes, beam_width, blank):
    """probs [B,T,C] array-like, sizes [B] or None -> (tokens [B,beam,T], steps [B,beam,T], lens [B,beam], scores [B,beam]).
    Beam p of an utterance is its greedy path with the first p tokens dropped."""
    probs = np.asarray(probs, dtype=np.float32)
    B, T, _ = probs.shape
    tok = np.zeros((B, beam_width, T), dtype=np.int32)
    steps = np.zeros((B, beam_width, T), dtype=np.int32)
    lens = np.zeros((B, beam_width), dtype=np.int32)
    scores = np.zeros((B, beam_width), dtype=np.float32)
    for b in range(B):
        n = T if sizes is None else int(np.asarray(sizes).reshape(-1)[b])
        path = probs[b, :n].argmax(-1)
        ids, at, prev = [], [], -1
        for t, c in enumerate(path):
            if c != blank and c != prev:
                ids.append(int(c)); at.append(t)
            prev = c
        for p in range(beam_width):
            k = max(len(ids) - p, 0)
            tok[b, p, :k] = ids[p:]
            steps[b, p, :k] = at[p:]
            lens[b, p] = k
            scores[b, p] = b + 0.5 * p
    return tok, steps, lens, scores
