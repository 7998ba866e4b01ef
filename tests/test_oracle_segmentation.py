"""Known-answer checks of oracle/segmentation.py (restatement of the reference example's energy gate,
example_scripts/video_transcribe_simulation.py:68-143): hand-traced state machine cases."""
import numpy as np

from oracle import segmentation as oseg

STEP = 1024


def _audio(nhops, loud):
    x = np.zeros(nhops * STEP + 1)
    for h in loud:
        x[h * STEP:(h + 1) * STEP] = 1000.0
    return x


def test_phrase_with_lead_in_and_pause_accounting():
    # loud hops 5..10; pause_buffer_count = ceil(0.55/0.064) = 9, so the phrase closes after the 10th quiet
    # hop (hop 20, iterator = 21*STEP); it held 16 hops, 6 besides the pause > ceil(0.2/0.064) = 4 -> kept,
    # starting two hops before hop 5.
    segs, e = oseg.segment(_audio(40, range(5, 11)))
    assert segs == [(3 * STEP, 21 * STEP)]
    assert e.shape == (40,) and e[5] == 1000.0 and e[4] == 0.0


def test_short_burst_is_dropped_and_start_is_clamped():
    # three loud hops: 13 hops at close, 3 besides the pause, not > 4 -> dropped
    assert oseg.segment(_audio(40, range(30, 33)))[0] == []
    # a phrase in the very first hop cannot take the lead-in: start stays at 0 (script :111-113)
    assert oseg.segment(_audio(40, range(0, 8)))[0] == [(0, 18 * STEP)]
    # second hop: iterator - 2*step < 0 -> start = iterator
    assert oseg.segment(_audio(40, range(1, 9)))[0] == [(1 * STEP, 19 * STEP)]


def test_short_pause_does_not_split_and_open_phrase_at_the_end_is_dropped():
    loud = list(range(5, 10)) + list(range(15, 20))      # 5-hop pause < 9
    assert oseg.segment(_audio(60, loud))[0] == [(3 * STEP, 30 * STEP)]
    assert oseg.segment(_audio(25, range(18, 25)))[0] == []          # still speaking when the audio ends
    # the loop runs while iterator + step < len(audio): a recording of exactly k*step samples has k-1 hops
    assert len(oseg.segment(np.zeros(10 * STEP))[1]) == 9


def _golden_cases():
    import hashlib
    import json
    import os
    from danspeech_amd import synthetic as syn
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_segments.json")
    for c in json.load(open(path, encoding="utf-8"))["cases"]:
        pcm = syn.gated_signal(c["plan"], c["seed"])
        assert hashlib.sha256(pcm.tobytes()).hexdigest() == c["sha256"], "the seeded signal is not the one the fixture was made from"
        yield c, pcm.astype(np.float64)[c["offset_seconds"] * 16000:]          # what the script's loop sees after load_audio + offset


def test_restatement_equals_the_reference_script_itself():
    """G9: tools/gen_golden_segments.py executed example_scripts/video_transcribe_simulation.py (the script's own loop,
    a recording Recognizer in place of the real one) on these seeded signals; the slices it handed to recognize() are the
    fixture.  Pins oracle/segmentation.py to the reference's behaviour, not just to its text."""
    n = 0
    for c, audio in _golden_cases():
        segs, _ = oseg.segment(audio)
        assert [list(s) for s in segs] == c["segments"], c["name"]
        n += len(segs)
    assert n >= 12
