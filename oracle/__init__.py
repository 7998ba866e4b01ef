"""CPU oracle for the recognize() hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy restatement of what the reference computes on the path
``Recognizer.recognize -> DanSpeechRecognizer.transcribe -> parse_audio ->
DeepSpeech.forward -> decoder.decode`` (reference danspeech/Recognizer.py:82-95,
danspeech/DanSpeechRecognizer.py:218-231). Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
the shipped package (``danspeech_amd``) never does and fails loudly when its HIP
library is missing.

Pinning status
--------------
* ``oracle.model`` (conv stack, BatchRNN, FC, softmax) and ``oracle.decoder.greedy``:
  PINNED -- checked against golden vectors produced by importing the reference
  itself in the build container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``).
* ``oracle.features`` (STFT/log1p/normalise): parity vs librosa UNPINNED (librosa is
  not installable here and the reference has no test for it); cross-checked against
  ``torch.stft`` as an independent implementation.
* ``oracle.beam`` / ``oracle.lm`` (CTC prefix beam search + n-gram scorer): PARITY
  UNPINNED -- the arithmetic lives in the un-vendored third-party ``ctcdecode``
  (parlance, unpinned master; reference danspeech/deepspeech/decoder.py:95-100) and
  KenLM; restated from the published algorithm and anchored on brute-force CTC
  known-answer tests.
* ``oracle.streaming`` model half (MaskConvStream / BatchRNNStream / LookaheadStream /
  streaming_forward): PINNED by ``tests/golden/g8_streaming.npz`` (the reference's own
  streaming model, ``tools/gen_golden.py g8``).  Parser half
  (InferenceSpectrogramAudioParser): UNPINNED like ``oracle.features`` (librosa).
* ``oracle.segmentation``: restates an example script that cannot be imported
  (argparse + downloaded models) and has no recorded output: UNPINNED, anchored on
  hand-traced known answers.
"""
