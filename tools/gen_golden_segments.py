#!/usr/bin/env python3
"""Generate tests/golden/g9_segments.json by running the REFERENCE's own long-form example (build container only).

``example_scripts/video_transcribe_simulation.py`` is a command-line script: its energy gate lives in the
``if __name__ == '__main__'`` block and cannot be imported as a function.  Here the script itself is executed
(``runpy.run_path(..., run_name='__main__')``) on seeded WAV files, with ``danspeech`` imported from /root/reference by
the stub recipe of tools/gen_golden.py and three names replaced: ``Recognizer`` by a recorder that notes which slice of
the loaded audio every ``recognize()`` call receives, and the two model factories (downloads) by no-ops.  Everything
that decides the slices -- chunking, energy, thresholds, counters, ``load_audio`` -- is the reference's code.

Only data is written (the recipe of each signal, a sha256 of its samples, the slices and the lines the script printed).

    python tools/gen_golden_segments.py            # rewrites tests/golden/g9_segments.json
"""
import contextlib
import hashlib
import io
import json
import os
import runpy
import sys
import tempfile
import types
import wave

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _n in ("Levenshtein", "librosa", "wget"):
    sys.modules[_n] = types.ModuleType(_n)
import scipy.signal  # noqa: E402
import scipy.signal.windows as _W  # noqa: E402
for _w in ("hamming", "hann", "blackman", "bartlett"):
    setattr(scipy.signal, _w, getattr(_W, _w))

import numpy as np  # noqa: E402

import danspeech  # noqa: E402
import danspeech.language_models  # noqa: E402
import danspeech.pretrained_models  # noqa: E402

from danspeech_amd import synthetic as syn  # noqa: E402

SCRIPT = "/root/reference/example_scripts/video_transcribe_simulation.py"
CALLS = []


class RecordingRecognizer(object):
    """Stands where danspeech.Recognizer stands in the script; notes the sample range of every slice it is handed."""

    def __init__(self, *a, **kw):
        pass

    def recognize(self, audio, show_all=False):
        base = audio.base if audio.base is not None else audio
        start = (audio.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // audio.itemsize
        CALLS.append((int(start), int(start + len(audio))))
        return "phrase%d" % len(CALLS)


def run_case(name, plan, seed, offset=0):
    pcm = syn.gated_signal(plan, seed)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, name + ".wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(pcm.tobytes())
        del CALLS[:]
        argv, sys.argv = sys.argv, [SCRIPT, "--wav-path", path, "--offset", str(offset), "--outfile", os.path.join(tmp, "out.txt")]
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                runpy.run_path(SCRIPT, run_name="__main__")
        finally:
            sys.argv = argv
    return dict(name=name, plan=plan, seed=seed, offset_seconds=offset, n_samples=int(len(pcm)),
                sha256=hashlib.sha256(pcm.tobytes()).hexdigest(),
                segments=[[a - offset * 16000, b - offset * 16000] for a, b in CALLS],       # relative to audio[offset:], like the script's indices
                printed=buf.getvalue().splitlines())


def main():
    danspeech.Recognizer = RecordingRecognizer
    danspeech.pretrained_models.Folketinget = lambda *a, **kw: None
    danspeech.language_models.Folketinget3gram = lambda *a, **kw: None
    cases = [run_case(name, plan, seed, off) for name, plan, seed, off in syn.SEGMENT_CASES]
    out = os.path.join(ROOT, "tests", "golden", "g9_segments.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(dict(script="example_scripts/video_transcribe_simulation.py", cases=cases), f, indent=1)
    for c in cases:
        print(c["name"], len(c["segments"]), "segments", c["segments"][:3])


if __name__ == "__main__":
    main()
