"""Restatement of ``GreedyDecoder`` (reference danspeech/deepspeech/decoder.py:147-198).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import numpy as np


def space_index(labels):
    """Decoder.__init__, decoder.py:39-42."""
    return labels.index(" ") if " " in labels else len(labels)


def greedy_decode(probs, sizes, labels, blank_index=0):
    """``GreedyDecoder.decode`` -> (strings[B][1], offsets[B][1]).

    decoder.py:183-198: argmax over classes (first maximum wins, as torch.max does),
    then ``process_string`` with remove_repetitions=True (decoder.py:166-181): frame i is
    skipped when it is blank, or when i != 0 and it equals frame i-1's argmax; otherwise
    its character is appended and i recorded as the offset.
    """
    ids = np.argmax(probs, axis=2)
    strings, offsets = [], []
    for b in range(ids.shape[0]):
        n = int(sizes[b]) if sizes is not None else ids.shape[1]
        s, off = [], []
        for i in range(n):
            c = int(ids[b, i])
            if c == blank_index:
                continue
            if i != 0 and c == int(ids[b, i - 1]):
                continue
            s.append(labels[c])
            off.append(i)
        strings.append(["".join(s)])
        offsets.append([np.asarray(off, dtype=np.int32)])
    return strings, offsets
