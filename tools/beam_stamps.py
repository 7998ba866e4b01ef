#!/usr/bin/env python3
"""Anatomy of a beam-search frame: the kernel stamps the phase boundaries of utterance 0 (100 MHz clock) for 64 frames from
the middle of the clip (dsmi_debug_beam_stamps); this prints the mean time per phase for the BASELINE decoder geometries.

    python tools/beam_stamps.py            # on the GPU box
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from danspeech_amd import _native, synthetic as syn  # noqa: E402

PHASES = ["F1 candidates (thread 0's own work)", "   ... wait for the slowest wave", "F2 wave scans + next row", "F3 threshold bin", "F4 rank + number",
          "F5 slots", "F6 commit"]


def probs_for(B, T, talkative=True):
    from danspeech_amd.deepspeech.model import DeepSpeech
    kw = syn.TALKATIVE if talkative else dict(fc_gain=8.0)
    sd = syn.make_state_dict(2, "gru", 800, 5, seed=0, **kw)
    m = DeepSpeech("cfg", rnn_hidden_size=800, rnn_layers=5).load_state_dict(sd).to("cuda")
    n = np.full(B, (T - 1) * 160, dtype=np.int64)
    pcm = torch.from_numpy(np.concatenate([syn.make_clip(i, int(n[0])) for i in range(B)])).cuda()
    fe = _native.NativeFrontend()
    feat, frames = fe.features(pcm, n)
    probs, sizes = m(feat, torch.from_numpy(frames.astype(np.int32)))
    return probs, np.asarray(sizes).astype(np.int32)


def main():
    lm3 = os.path.join(tempfile.gettempdir(), "stamps3.arpa")
    syn.make_arpa(lm3, order=3, n_words=5000, seed=11, ngrams_per_order=20000)
    probs, sizes = probs_for(32, 1001)
    for beam, lm in ((64, None), (64, lm3), (128, lm3)):
        dec = _native.NativeDecoder(syn.DANSPEECH_LABELS, blank_index=0)
        dec.set_lm(lm, 1.3, 0.2)
        dec.beam(probs, sizes, beam_width=beam)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        dec.beam_enqueue(probs, sizes, beam_width=beam)
        ev1.record()
        out = dec.beam_collect()
        torch.cuda.synchronize()
        st = dec.beam_stamps().astype(np.int64)
        ok = st[:, 0] > 0
        st = st[ok]
        d = np.stack([st[:, 7] - st[:, 0], st[:, 1] - st[:, 7], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3],
                      st[:, 5] - st[:, 4], st[:, 6] - st[:, 5]], axis=1) * 10.0     # ns
        frame = (st[1:, 0] - st[:-1, 0]) * 10.0
        print("beam %d, %s: kernel %.2f ms for %d frames (%.2f us/frame); stamped frames: %.2f us/frame; stats %s" %
              (beam, "3-gram" if lm else "no LM", ev0.elapsed_time(ev1), int(sizes.max()), ev0.elapsed_time(ev1) * 1e3 / int(sizes.max()),
               frame.mean() / 1e3, dec.beam_stats()))
        for name, v in zip(PHASES, d.mean(axis=0)):
            print("    %-42s %7.0f ns" % (name, v))
        print("    best beam of clip 0: %d tokens" % out[2][0, 0])
        dec.close()


if __name__ == "__main__":
    main()
