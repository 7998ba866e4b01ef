"""Public facade of the drop-in surface (reference danspeech/Recognizer.py:13-130).

``Recognizer(model=None, lm=None, with_gpu=False, **kwargs)``, ``recognize``, ``update_model``,
``update_decoder`` with the reference's signatures, prints and exceptions, plus
``recognize_batch`` / ``recognize_files`` / ``recognize_long``.  Of the streaming half of the reference
class (Recognizer.py:133-818) the real-time path is provided array-driven
(``enable_real_time_streaming`` + ``stream_recording``); the microphone threads are live-audio control
flow outside the hot path.
"""
import numpy as np

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

    def recognize_batches(self, batches, show_all=False):
        """Generator: ``recognize_batch`` over a sequence of batches, with the upload of the next batch and the
        decoding of the previous one overlapped with the GPU's work on the current one."""
        return self.danspeech_recognizer.transcribe_batches(batches, show_all=show_all)

    def recognize_batch_distributed(self, audio_list):
        """``recognize_batch`` sharded over the ranks of an initialised ``torch.distributed`` job (one process per
        GPU, backend "nccl" = RCCL): rank 0 passes the clips and receives the transcriptions in its order, the
        other ranks pass ``None`` and receive ``None``.  Clips are dealt longest-first round-robin
        (``parallel.plan_shards``), int16/float32/float64 PCM travels in its own sample type."""
        import torch
        import torch.distributed as dist
        from . import parallel
        eng = self.danspeech_recognizer
        dev = torch.device("cuda", eng._device_index())
        return parallel.recognize_sharded(eng, audio_list, dist.get_rank(), dist.get_world_size(), dev)

    # ---- real-time streaming (Recognizer.py:499-720) without the microphone -----------------------------
    def enable_real_time_streaming(self, streaming_model, secondary_model=None, string_parts=True):
        """Recognizer.py:499-533: switch to a unidirectional streaming model (e.g.
        ``pretrained_models.GPUStreamingRNN``), optionally with a secondary model for the final text."""
        self.update_model(streaming_model)
        self.danspeech_recognizer.enable_streaming(secondary_model, string_parts)
        self.stream = True

    def disable_real_time_streaming(self, keep_secondary_model_loaded=False):
        """Recognizer.py:535-558 (no microphone thread to stop here)."""
        if getattr(self, "stream", False):
            self.stream = False
            self.danspeech_recognizer.disable_streaming(keep_secondary_model=keep_secondary_model_loaded)
        else:
            print("No stream is running for the Recognizer")

    def stream_recording(self, audio_data, chunk_samples=None):
        """The body of ``real_time_streaming`` (Recognizer.py:560-710) driven by an array instead of the
        microphone thread: the utterance is cut with the reference's sample requirements (first pass
        ``general + 15 * samples_pr_10ms``, later passes ``general``, :598-612; ``chunk_samples`` is the size
        of the parts the source would deliver), every part goes through
        ``DanSpeechRecognizer.streaming_transcribe`` and ``(is_last, text)`` is yielded for every non-empty
        output.  Requires ``enable_real_time_streaming``."""
        if not getattr(self, "stream", False):
            raise RuntimeError("call enable_real_time_streaming(streaming_model) first")
        rec = self.danspeech_recognizer
        lookahead_context = rec.model.context
        required_spec_frames = (lookahead_context - 1) * 2
        samples_pr_10ms = int(rec.audio_parser.sampling_rate / 100)
        general_sample_requirement = samples_pr_10ms * 2 + (samples_pr_10ms * (required_spec_frames - 1))
        first_sample_requirement = general_sample_requirement + (samples_pr_10ms * 15)
        audio_data = np.asarray(audio_data, dtype=np.float64)
        step = int(chunk_samples) if chunk_samples else 1024
        pos, n = 0, len(audio_data)
        is_first_pass = True
        data_array = audio_data[:0]
        while pos < n:
            part = audio_data[pos:pos + step]
            pos += len(part)
            is_last = pos >= n
            data_array = np.concatenate((data_array, part))
            output = None
            if is_first_pass:
                if is_last:
                    output = None                      # too short for a first pass: discarded (:666-667)
                elif len(data_array) >= first_sample_requirement:
                    output = rec.streaming_transcribe(data_array, is_last=False, is_first=True)
                    is_first_pass = False
                    data_array = audio_data[:0]
            else:
                if is_last or len(data_array) >= general_sample_requirement:
                    output = rec.streaming_transcribe(data_array, is_last=is_last, is_first=False)
                    data_array = audio_data[:0]
            if output:
                yield is_last, output

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
