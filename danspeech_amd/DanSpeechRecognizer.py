"""The engine of the drop-in surface (reference danspeech/DanSpeechRecognizer.py:13-231,
non-streaming part): owns the device, the model, the audio parser and the decoder.

Same constructor, ``update_model`` / ``update_decoder`` state machine, prints and warnings as
the reference; ``transcribe`` is the reference's batch-1 path and ``transcribe_batch`` is the
batched extension the MI355X needs to be fed properly.  Everything numeric
runs on the GPU through libdsmi.so; there is no CPU path, so ``with_gpu=False`` is accepted
for signature compatibility but the device is still the MI355X.
"""
import warnings

import numpy as np

from .deepspeech.decoder import GreedyDecoder, BeamCTCDecoder
from .errors.recognizer_errors import ModelNotInitialized
from .audio.parsers import SpectrogramAudioParser, InferenceSpectrogramAudioParser


class NoLmInstantiatedWarning(Warning):
    pass


class DanSpeechRecognizer(object):

    def __init__(self, model_name=None, lm_name=None, alpha=1.3, beta=0.2, with_gpu=False, beam_width=64):
        import torch
        self.device = torch.device("cuda")
        print("Using device: {0}".format(self.device))
        # state first (the reference reads self.lm / self.decoder inside update_model before
        # they exist when a model is passed here, DanSpeechRecognizer.py:23-24 vs 43-46)
        self.lm = None
        self.decoder = None
        self.alpha = alpha
        self.beta = beta
        self.beam_width = beam_width
        if model_name:
            self.update_model(model_name)
        else:
            self.model = None
            self.model_name = None
            self.labels = None
            self.audio_config = None
            self.audio_parser = None
        if lm_name:
            if not self.model:
                raise ModelNotInitialized("Trying to initialize LM without also choosing a DanSpeech model.")
            else:
                self.update_decoder(lm_name)
                self.lm = lm_name

    def update_model(self, model):
        self.audio_config = model.audio_conf
        self.model = model.to(self.device)
        self.model.eval()
        index = int(str(self.model.device).split(":")[1]) if ":" in str(self.model.device) else 0
        self.audio_parser = SpectrogramAudioParser(self.audio_config, device=index)
        self.labels = self.model.labels
        # When updating model, always update decoder because of labels
        self.update_decoder(labels=self.labels)

    def update_decoder(self, lm=None, alpha=None, beta=None, labels=None, beam_width=None):
        """DanSpeechRecognizer.py:58-95, verbatim semantics: falsy arguments are ignored, the decoder
        is rebuilt only when something changed, the first call selects greedy decoding."""
        update = False
        if not self.lm and not self.decoder:
            update = True
            self.lm = "greedy"
        if lm and self.lm != lm:
            update = True
            self.lm = lm
        if alpha and self.alpha != alpha:
            update = True
            self.alpha = alpha
        if beta and self.beta != beta:
            update = True
            self.beta = beta
        if labels and labels != self.labels:
            update = True
            self.labels = labels
        if beam_width and beam_width != self.beam_width:
            update = True
            self.beam_width = beam_width
        if update:
            if self.lm != "greedy":
                self.decoder = BeamCTCDecoder(labels=self.labels, lm_path=self.lm,
                                              alpha=self.alpha, beta=self.beta,
                                              beam_width=self.beam_width, num_processes=6, cutoff_prob=1.0,
                                              cutoff_top_n=40, blank_index=self.labels.index('_'))
            else:
                self.decoder = GreedyDecoder(labels=self.labels, blank_index=self.labels.index('_'))

    # ---- streaming (DanSpeechRecognizer.py:97-216) -----------------------------------------------
    def _device_index(self):
        return int(str(self.model.device).split(":")[1]) if ":" in str(self.model.device) else 0

    def enable_streaming(self, secondary_model=None, return_string_parts=True):
        """DanSpeechRecognizer.py:97-127."""
        self.full_output = []
        self.iterating_transcript = ""
        if secondary_model:
            self.secondary_model = secondary_model.to(self.device)
            self.secondary_model.eval()
        else:
            self.secondary_model = None
        self.spectrograms = []
        self.greedy_decoder = GreedyDecoder(labels=self.labels, blank_index=self.labels.index('_'))
        self.audio_parser = InferenceSpectrogramAudioParser(audio_config=self.audio_config, device=self._device_index())
        self.string_parts = bool(return_string_parts)

    def disable_streaming(self, keep_secondary_model=False):
        """DanSpeechRecognizer.py:129-136."""
        self.audio_parser = SpectrogramAudioParser(self.audio_config, device=self._device_index())
        self.greedy_decoder = None
        self.reset_streaming_params()
        self.string_parts = False
        if not keep_secondary_model:
            self.secondary_model = None

    def reset_streaming_params(self):
        self.iterating_transcript = ""
        self.full_output = []
        self.spectrograms = []

    def streaming_transcribe(self, recording, is_last, is_first):
        """DanSpeechRecognizer.py:144-216: one part of an utterance through the streaming model; greedy text of
        the part (or the whole running text), and on ``is_last`` the final transcription (secondary model
        on the collected spectrograms, or the LM decoder on the collected outputs, or the running text)."""
        import torch
        recording = self.audio_parser.parse_audio(recording, is_last)
        out = ""
        if len(recording) != 0:
            if self.secondary_model:
                self.spectrograms.append(recording)
            recording = recording.view(1, 1, recording.size(0), recording.size(1))
            recording = recording.to(self.device)
            out = self.model(recording, is_first, is_last)
            if is_first:
                return ""
            self.full_output.append(out)
            decoded_out, _ = self.greedy_decoder.decode(out)
            transcript = decoded_out[0][0]
            # Collapsing characters hack
            if self.iterating_transcript and transcript and self.iterating_transcript[-1] == transcript[0]:
                self.iterating_transcript = self.iterating_transcript + transcript[1:]
                transcript = transcript[1:]
            else:
                self.iterating_transcript += transcript
            if self.string_parts:
                out = transcript
            else:
                out = self.iterating_transcript
        if is_last:
            if len(self.iterating_transcript) > 1:
                if self.secondary_model:
                    final = torch.cat(self.spectrograms, dim=1)
                    self.spectrograms = []
                    final = final.view(1, 1, final.size(0), final.size(1))
                    final = final.to(self.device)
                    input_sizes = torch.IntTensor([final.size(3)]).int()
                    out, _ = self.secondary_model(final, input_sizes)
                    decoded_out, _ = self.decoder.decode(out)
                    decoded_out = decoded_out[0][0]
                    self.reset_streaming_params()
                    return decoded_out
                else:
                    if self.lm != "greedy":
                        final_out = torch.cat(self.full_output, dim=1)
                        decoded_out, _ = self.decoder.decode(final_out)
                        decoded_out = decoded_out[0][0]
                        self.reset_streaming_params()
                        return decoded_out
                    else:
                        out = self.iterating_transcript
                        self.reset_streaming_params()
                        return out
            else:
                return ""
        return out

    def transcribe(self, recording, show_all=False):
        """DanSpeechRecognizer.py:218-231."""
        import torch
        recording = self.audio_parser.parse_audio(recording)
        recording = recording.view(1, 1, recording.size(0), recording.size(1))
        recording = recording.to(self.device)
        input_sizes = torch.IntTensor([recording.size(3)]).int()
        out, output_sizes = self.model(recording, input_sizes)
        decoded_output, _ = self.decoder.decode(out, output_sizes)
        if show_all:
            if self.lm == 'greedy':
                warnings.warn("You are trying to get all beams but no LM has been instantiated.",
                              NoLmInstantiatedWarning)
            return decoded_output[0]
        else:
            return decoded_output[0][0]

    def transcribe_long(self, recording, energy_threshold=600, step=1024, pause_threshold=0.55, phrase_threshold=0.2,
                        max_batch=32, show_all=False):
        """Long-form transcription: the energy gate of the reference's
        example_scripts/video_transcribe_simulation.py:68-143 cuts the recording into phrases
        (``dsmi_segment``: hop energies on the GPU), the phrases are transcribed in batches of at most
        ``max_batch`` (longest first), and ``[(start_sample, end_sample, transcription), ...]`` comes back
        in time order.  The recording is uploaded once; phrases are sliced on the device."""
        import torch
        parser = self.audio_parser
        pcm = torch.from_numpy(np.ascontiguousarray(recording, dtype=np.float64)).to("cuda:%d" % parser.device)
        hop_seconds = step / float(parser.sampling_rate)
        segs = parser._frontend().segment(pcm, energy_threshold=energy_threshold, step=step,
                                          pause_hops=int(np.ceil(pause_threshold / hop_seconds)),
                                          phrase_hops=int(np.ceil(phrase_threshold / hop_seconds)))
        res = [None] * len(segs)
        order = sorted(range(len(segs)), key=lambda i: -(int(segs[i][1]) - int(segs[i][0])))
        for k in range(0, len(order), max_batch):
            idxs = order[k:k + max_batch]
            n = np.array([int(segs[i][1] - segs[i][0]) for i in idxs], dtype=np.int64)
            cat = torch.cat([pcm[int(segs[i][0]):int(segs[i][1])] for i in idxs])
            feats, frames = parser._frontend().features(cat, n)
            out, output_sizes = self.model(feats, torch.from_numpy(frames.astype(np.int32)))
            decoded_output, _ = self.decoder.decode(out, output_sizes)
            for pos, i in enumerate(idxs):
                res[i] = (int(segs[i][0]), int(segs[i][1]), decoded_output[pos] if show_all else decoded_output[pos][0])
        return res

    def transcribe_files(self, paths, show_all=False):
        """``transcribe_batch([load_audio(p) for p in paths])`` without decoding the files on the host:
        the WAV frames go to the GPU as bytes and ``dsmi_features`` applies ``load_audio``'s sample-width
        conversion and saturating stereo fold (resources.py:302-303).  Files are grouped by
        (sample width, channels); every group is one batch."""
        import torch
        from .audio.resources import read_wav_frames
        if len(paths) == 0:
            return []
        loaded = [read_wav_frames(p) for p in paths]
        groups = {}
        for i, (raw, width, nch) in enumerate(loaded):
            groups.setdefault((width, nch), []).append(i)
        res = [None] * len(paths)
        for (width, nch), idxs in groups.items():
            idxs = sorted(idxs, key=lambda i: -len(loaded[i][0]))       # stable: longest first
            feats, frames = self.audio_parser.parse_wav_frames([loaded[i][0] for i in idxs], width, nch)
            out, output_sizes = self.model(feats, torch.from_numpy(frames.astype(np.int32)))
            decoded_output, _ = self.decoder.decode(out, output_sizes)
            for pos, i in enumerate(idxs):
                res[i] = decoded_output[pos] if show_all else decoded_output[pos][0]
        if show_all and self.lm == 'greedy':
            warnings.warn("You are trying to get all beams but no LM has been instantiated.", NoLmInstantiatedWarning)
        return res

    def transcribe_batch(self, recordings, show_all=False):
        """Batched ``transcribe``: clips are sorted by length (pack_padded_sequence's order,
        reference model.py:117), run as ONE batch, and results return in the caller's order."""
        import torch
        if len(recordings) == 0:
            return []
        order = np.argsort([-len(r) for r in recordings], kind="stable")
        feats, frames = self.audio_parser.parse_batch([recordings[i] for i in order])
        out, output_sizes = self.model(feats, torch.from_numpy(frames.astype(np.int32)))
        decoded_output, _ = self.decoder.decode(out, output_sizes)
        if show_all and self.lm == 'greedy':
            warnings.warn("You are trying to get all beams but no LM has been instantiated.", NoLmInstantiatedWarning)
        res = [None] * len(recordings)
        for pos, i in enumerate(order):
            res[i] = decoded_output[pos] if show_all else decoded_output[pos][0]
        return res
