"""Opt-in: the reference's recorded example output (docs_source/auto_examples/execute_recognize.rst:34-50 -- TestModel +
DSL3gram, alpha 1.2, beta 0.15, beam_width 10 on example_files/u0013002.wav) reproduced with the REAL artefacts.

The pretrained ``TestModel.pth`` and ``dsl_3gram.klm`` cannot be fetched here (no network); when both sit in
``~/.danspeech/{models,lms}/`` with the md5 sums the reference's factories pin (test_model.py:26-27, dsl_3gram.py:17-18)
this test runs the gallery script's calls through the drop-in surface and compares with the listing.  It is the only
anchor to real KenLM / ctcdecode / librosa output available for this path; without the files it is skipped.
"""
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "golden", "u0013002.wav")
MODEL = os.path.join(os.path.expanduser("~"), ".danspeech", "models", "TestModel.pth")
LM = os.path.join(os.path.expanduser("~"), ".danspeech", "lms", "dsl_3gram.klm")
MODEL_MD5, LM_MD5 = "c21438a33f847a9c8d4e08779e98bf31", "33ca3e2a8db3a036af6d7ad85972dbb0"

GREEDY = "tester en to tre fire sem seks syv otte"
BEAMS = ["tester en to tre fire fem seks syv otte", "tester en to tre fire fem seks syv ofte",
         "tester en to tre fire fem seks syv otter", "tester en to tre fire fem seks syv tte",
         "tester en to tre fire fem seks syv ottey", "tester en to tre fire fem seks syv ote",
         "tester en to tre fire fem seks syv ottet", "tester en to tre fire fem seks syv ottek",
         "tester en to tre fire fem seks syv ottes", "tester en to tre fire fem seks syv otteo"]


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def _have_artefacts():
    return (os.path.exists(MODEL) and os.path.exists(LM) and _md5(MODEL) == MODEL_MD5 and _md5(LM) == LM_MD5)


@pytest.mark.skipif(not _have_artefacts(), reason="TestModel.pth / dsl_3gram.klm (reference md5s) are not in ~/.danspeech")
def test_gallery_listing_is_reproduced():
    from danspeech_amd import Recognizer
    from danspeech_amd.pretrained_models import TestModel
    from danspeech_amd.language_models import DSL3gram
    from danspeech_amd.audio import load_audio
    recognizer = Recognizer(model=TestModel())
    audio = load_audio(path=WAV)
    assert recognizer.recognize(audio) == GREEDY
    recognizer.update_decoder(lm=DSL3gram(), alpha=1.2, beta=0.15, beam_width=10)
    assert recognizer.recognize(audio, show_all=False) == BEAMS[0]
    assert recognizer.recognize(audio, show_all=True) == BEAMS
