"""Public facade of the drop-in surface (reference danspeech/Recognizer.py:13-130).

``Recognizer(model=None, lm=None, with_gpu=False, **kwargs)``, ``recognize``, ``update_model``,
``update_decoder`` with the reference's signatures, prints and exceptions, plus
``recognize_batch``.  The microphone / streaming half of the reference class
(Recognizer.py:133-818) is live-audio control flow outside the hot path and is not provided.
"""
from .errors.recognizer_errors import ModelNotInitialized
from .DanSpeechRecognizer import DanSpeechRecognizer


class Recognizer(object):

    def __init__(self, model=None, lm=None, with_gpu=False, **kwargs):
        self.danspeech_recognizer = DanSpeechRecognizer(with_gpu=with_gpu, **kwargs)
        self.stream = False
        self.stream_thread_stopper = None
        if model:
            self.update_model(model)
        if lm:
            if not model:
                raise ModelNotInitialized("Trying to initialize language model without also choosing a DanSpeech "
                                          "acoustic model.")
            else:
                self.update_decoder(lm=lm)
        self.microphone = None

    def recognize(self, audio_data, show_all=False):
        """Most likely transcription of ``audio_data`` (numpy array as returned by ``load_audio``);
        all beams when ``show_all`` and a language model is set."""
        return self.danspeech_recognizer.transcribe(audio_data, show_all=show_all)

    def recognize_batch(self, audio_list, show_all=False):
        """``recognize`` for a list of clips in one batched pass over the GPU."""
        return self.danspeech_recognizer.transcribe_batch(audio_list, show_all=show_all)

    def recognize_long(self, audio_data, energy_threshold=600, step=1024, pause_threshold=0.55, phrase_threshold=0.2,
                       max_batch=32, show_all=False):
        """Segment a long recording with the reference example's energy gate
        (example_scripts/video_transcribe_simulation.py:68-143) and transcribe the phrases in batches:
        ``[(start_sample, end_sample, transcription), ...]`` in time order."""
        return self.danspeech_recognizer.transcribe_long(audio_data, energy_threshold=energy_threshold, step=step,
                                                         pause_threshold=pause_threshold, phrase_threshold=phrase_threshold,
                                                         max_batch=max_batch, show_all=show_all)

    def recognize_files(self, paths, show_all=False):
        """``[recognize(load_audio(p)) for p in paths]`` in batched passes, WAV decoding on the GPU."""
        return self.danspeech_recognizer.transcribe_files(paths, show_all=show_all)

    def update_model(self, model):
        self.danspeech_recognizer.update_model(model)
        print("DanSpeech model updated to: {0}".format(model.model_name))

    def update_decoder(self, lm=None, alpha=None, beta=None, beam_width=None):
        self.danspeech_recognizer.update_decoder(lm=lm, alpha=alpha, beta=beta, beam_width=beam_width)
        print("DanSpeech decoder updated ")  # ToDO: Include model name
