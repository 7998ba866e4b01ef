"""The engine behind ``Recognizer`` (the role of reference danspeech/DanSpeechRecognizer.py): it owns the
device, the acoustic model, the audio parser and the CTC decoder, and turns recordings into text.

Behaviour kept from the reference, because callers and the parity tests rely on it:

* constructor arguments and defaults (``alpha=1.3, beta=0.2, beam_width=64``, DanSpeechRecognizer.py:16-17), the
  ``Using device:`` line, ``ModelNotInitialized`` when a language model is given without a model (:36-38);
* ``update_decoder``: only truthy arguments that differ from the current setting count as a change, the decoder object
  is rebuilt only on a change, the very first call selects greedy decoding (:58-95); the beam decoder is created with
  ``num_processes=6, cutoff_prob=1.0, cutoff_top_n=40`` and the blank at ``labels.index('_')`` (:89-94);
* ``transcribe``: one recording -> best transcription, or every beam with ``show_all`` (with a
  ``NoLmInstantiatedWarning`` when there is no language model, :218-231);
* the chunked real-time protocol ``enable_streaming`` / ``streaming_transcribe`` / ``disable_streaming`` (:97-216).

What is new is batching: ``transcribe_batch`` runs many recordings as ONE padded batch, and ``transcribe_batches``
keeps one batch in flight on the GPU while the next one is staged and uploaded and the previous one is decoded.
Everything numeric runs in libdsmi.so on the MI355X; ``with_gpu`` is accepted for signature compatibility only.
"""
import warnings

import os
import numpy as np

from .deepspeech.decoder import GreedyDecoder, BeamCTCDecoder
from .errors.recognizer_errors import ModelNotInitialized
from .audio.parsers import SpectrogramAudioParser, InferenceSpectrogramAudioParser, DeviceClips, StagedClips


class NoLmInstantiatedWarning(Warning):
    pass


# decoder settings update_decoder can change, in the order the reference examines them
_DECODER_SETTINGS = ("lm", "alpha", "beta", "labels", "beam_width")


class _StreamingSession(object):
    """Running state of one utterance that arrives in parts (reference DanSpeechRecognizer.py:97-216): the text so
    far, the model outputs of every part (for a final pass of the language-model decoder) and, when a secondary
    model gives the final text, the spectrograms of every part."""

    def __init__(self, secondary_model, string_parts):
        self.secondary_model = secondary_model
        self.string_parts = string_parts
        self.clear()

    def clear(self):
        self.text = ""
        self.outputs = []
        self.spectrograms = []

    def extend_text(self, piece):
        """Append a part's greedy text; a part that starts with the character the text ends with continues that
        character (CTC collapse across the part boundary).  Returns what this part contributed."""
        if self.text and piece and self.text[-1] == piece[0]:
            piece = piece[1:]
        self.text += piece
        return piece


class _UnmergedDeviceClips(object):
    """Device-resident batches on their way into one forward: merged on the stream that runs it."""

    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return sum(len(b) for b in self.batches)


class _BatchJob(object):
    """One batch between enqueue and decode."""
    __slots__ = ("order", "probs", "sizes", "count", "model", "ticket", "slot", "collected", "recomputed")

    def __init__(self, order, probs, sizes, count, model):
        self.order, self.probs, self.sizes, self.count, self.model = order, probs, sizes, count, model
        self.ticket = None                       # a beam search launched behind the forward (transcribe_batches)
        self.slot = 0                            # the decoder handle that search occupies
        self.collected = self.recomputed = False # the forward has been waited for (model.collect) / had to be redone

    def collect_forward(self):
        if not self.collected:
            self.recomputed = self.model.collect()
            self.collected = True
        return self.recomputed


_SIDE_STREAMS = {}        # (name, device index) -> torch.cuda.Stream, shared by every engine of the process: see _side_stream


class DanSpeechRecognizer(object):

    def __init__(self, model_name=None, lm_name=None, alpha=1.3, beta=0.2, with_gpu=False, beam_width=64):
        import torch
        from . import _native
        _native.want_hw_queues(8)          # one hardware queue per stream of the batch pipeline, if the runtime can still be told
        self.device = torch.device("cuda")
        print("Using device: {0}".format(self.device))
        self.lm = None
        self.decoder = None
        self.alpha, self.beta, self.beam_width = alpha, beta, beam_width
        self.model = self.model_name = self.labels = self.audio_config = self.audio_parser = None
        self._session = None
        self._side_streams = {}
        self._replicas = []
        if model_name:
            self.update_model(model_name)
        if lm_name:
            if not self.model:
                raise ModelNotInitialized("Trying to initialize LM without also choosing a DanSpeech model.")
            self.update_decoder(lm_name)

    # ---- model / decoder lifecycle ----------------------------------------------------------------------------------
    def _device_index(self):
        name = str(self.model.device)
        return int(name.split(":")[1]) if ":" in name else 0

    def update_model(self, model):
        self.audio_config = model.audio_conf
        self.model = model.to(self.device)
        self.model.eval()
        self._replicas = []
        self.audio_parser = SpectrogramAudioParser(self.audio_config, device=self._device_index())
        # a new model may bring a new alphabet: the decoder follows
        self.update_decoder(labels=self.model.labels)

    def _build_decoder(self):
        blank = self.labels.index("_")
        if self.lm == "greedy":
            return GreedyDecoder(labels=self.labels, blank_index=blank)
        return BeamCTCDecoder(labels=self.labels, lm_path=self.lm, alpha=self.alpha, beta=self.beta,
                              beam_width=self.beam_width, num_processes=6, cutoff_prob=1.0, cutoff_top_n=40,
                              blank_index=blank)

    def update_decoder(self, lm=None, alpha=None, beta=None, labels=None, beam_width=None):
        requested = dict(lm=lm, alpha=alpha, beta=beta, labels=labels, beam_width=beam_width)
        stale = not self.lm and not self.decoder
        if stale:
            self.lm = "greedy"
        for name in _DECODER_SETTINGS:
            value = requested[name]
            if value and value != getattr(self, name):
                setattr(self, name, value)
                stale = True
        if stale:
            self.decoder = self._build_decoder()

    # ---- batches ----------------------------------------------------------------------------------------------------
    def _side_stream(self, name):
        """A HIP stream beside the compute stream ("lane k": the k-th forward in flight; "decode": the decoder of batch i runs
        while batch i + 1 computes).  ONE set per device for every engine of the process: the ROCm runtime deals each stream a
        process creates onto GPU_MAX_HW_QUEUES hardware queues in turn, so a second engine with streams of its own would find
        two of them on one queue as often as not, and those run one after the other (measured: config 5's share 113 ms per
        batch as the fourth engine of a process, 100 ms alone).  Engines are used one at a time; two used at once share the lanes."""
        import torch
        key = (name, self._device_index())
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=key[1])
        self._side_streams[key] = _SIDE_STREAMS[key]
        return _SIDE_STREAMS[key]

    def _stage_batch(self, recordings, parser=None):
        """Host clips, longest first (pack_padded_sequence's order, reference model.py:117), copied to pinned memory and on
        their way to the device: (order, StagedClips).  Needs no model handle, so a pipeline calls it before it waits for one."""
        order = np.argsort([-len(r) for r in recordings], kind="stable")
        return order, (parser or self.audio_parser).stage([recordings[i] for i in order])

    def _enqueue_batch(self, recordings, model=None, parser=None, decode_slot=None, staged=None):
        """Stage + upload + spectrograms + forward of one batch, all asynchronous.  Clips run longest first
        (pack_padded_sequence's order, reference model.py:117)."""
        import torch
        model = model or self.model
        if isinstance(recordings, DeviceClips):             # already on the device, longest first
            order = getattr(recordings, "order", None)
            order = np.arange(len(recordings)) if order is None else order      # (a merged batch remembers where its clips came from)
            # the clips were produced on whatever stream was current when the batch was handed over (an RCCL scatter, a widening
            # copy, a slicing kernel): the stream this batch runs on waits for that point, and the allocator learns of the use
            here = torch.cuda.current_stream(self._device_index())
            if isinstance(staged, torch.cuda.Event):
                here.wait_event(staged)
            recordings.pcm.record_stream(here)
            feats, frames = (parser or self.audio_parser).parse_batch(recordings)
        else:
            if staged is not None:
                order, clips = staged
            else:
                order = np.argsort([-len(r) for r in recordings], kind="stable")
                clips = [recordings[i] for i in order]
            feats, frames = (parser or self.audio_parser).parse_batch(clips)
        probs, sizes = model.enqueue(feats, torch.from_numpy(frames.astype(np.int32)))
        job = _BatchJob(order, probs, sizes, len(recordings), model)
        if decode_slot is not None and hasattr(self.decoder, "decode_enqueue"):
            job.slot = decode_slot
            if getattr(self.decoder, "on_lane", False):
                # greedy decoding: a short kernel and three small copies, queued right behind the forward on its own stream -- the
                # host never stands in a busy device's copy queue while the stream it should be feeding runs dry
                job.ticket = self.decoder.decode_enqueue(probs, sizes, slot=decode_slot)
            else:
                # the beam search is a kernel: launched now, behind this forward, on a stream of its own, so that the host
                # never waits for it while it could be feeding the next batch
                side = self._side_stream("decode")           # one decode stream for all slots: see audio/parsers.py on hardware queues
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self._device_index()))
                side.wait_event(done)
                probs.record_stream(side)
                with torch.cuda.stream(side):
                    job.ticket = self.decoder.decode_enqueue(probs, sizes, slot=decode_slot)
        return job

    def _finish_batch(self, job, show_all, warn=True):
        import torch
        recomputed = job.collect_forward()       # waits for the forward; a timed-out batch has been recomputed by now
        side = self._side_stream("decode")
        job.probs.record_stream(side)
        decoded = None
        if getattr(job, "ticket", None) is not None:
            decoded, _ = self.decoder.decode_collect(job.ticket)      # the search launched behind the forward ...
            if recomputed:
                decoded = None                                         # ... read the probabilities of a forward that had to be redone
        if decoded is None:
            with torch.cuda.stream(side):
                if hasattr(self.decoder, "decode_enqueue"):
                    # on the job's OWN decoder handle: it is free again after the collect above, while handle 0 may hold the
                    # search of the batch enqueued after this one
                    decoded, _ = self.decoder.decode_collect(self.decoder.decode_enqueue(job.probs, job.sizes, slot=job.slot))
                else:
                    decoded, _ = self.decoder.decode(job.probs, job.sizes)
        if warn and show_all and self.lm == "greedy":
            warnings.warn("You are trying to get all beams but no LM has been instantiated.", NoLmInstantiatedWarning)
        if getattr(self, "keep_last_output", False):
            self.last_output = (job.probs, job.sizes)        # bench.py checks the timed batch's probabilities against the oracle
        results = [None] * job.count
        for pos, i in enumerate(job.order):
            results[i] = decoded[pos] if show_all else decoded[pos][0]
        return results

    def transcribe_batch(self, recordings, show_all=False):
        """``[transcribe(r) for r in recordings]`` as one batch; results in the caller's order."""
        if len(recordings) == 0:
            return []
        return self._finish_batch(self._enqueue_batch(recordings), show_all)

    # how `transcribe_batches` fills the GPU: forwards in flight (each on a model handle, stream and persistent-kernel gate slot of
    # its own) and clips per forward (consecutive batches are merged up to this many: the recurrent kernel walks up to four
    # 16-clip tiles per workgroup, and its cost per clip falls with the tiles it walks)
    pipeline_lanes = 4
    pipeline_merge_clips = 64
    pipeline_balance_tail = True         # a sized source's last round of forwards is dealt evenly over the lanes (tail_plan)

    def _lanes(self, count):
        """The model handles, parsers and streams of the pipeline's forwards in flight: the engine's own and `count - 1` replicas."""
        import torch
        while len(self._replicas) < count - 1 and hasattr(self.model, "replica"):
            # every forward in flight has its own model handle AND its own parser (frontend scratch, staging buffers)
            self._replicas.append((self.model.replica(), SpectrogramAudioParser(self.audio_config, device=self._device_index())))
        handles = [self.model] + [r[0] for r in self._replicas[:count - 1]]
        parsers = [self.audio_parser] + [r[1] for r in self._replicas[:count - 1]]
        streams = [torch.cuda.current_stream(self._device_index())] + [self._side_stream("lane %d" % k) for k in range(1, len(handles))]
        return handles, parsers, streams

    def _lanes_that_pay(self, most, clips):
        """Forwards in flight when the caller did not say.  Several forwards side by side pay where the recurrent kernel of
        each holds a fifth of the chip (the ring form: GRU / RNN up to 896 units, LSTM up to 512, one window of up to 64
        clips -- ``transcribe_batches`` cuts larger batches to that) and the dense kernels of the others fill the rest.  A model
        whose recurrent kernel takes the whole device runs two: their recurrent launches take turns (csrc/api.hip, the turn
        lock) and each forward's GEMM runs beside the other's launch; a third forward only slows those launches down (config 4:
        33.7 ms per batch with two, 35.0 with three, 37.1 with four; profiles/r06_config4.txt)."""
        hidden, kind = getattr(self.model, "rnn_hidden_size", 0), getattr(self.model, "rnn_type", "gru")
        ring = hidden % 16 == 0 and hidden <= (512 if kind == "lstm" else 896)
        return most if ring and clips <= 64 else min(most, 2)

    def transcribe_batches(self, batches, show_all=False, lanes=None, merge_clips=None):
        """Generator over ``transcribe_batch(b)`` for every ``b`` of ``batches`` through the pipeline of ``_transcribe_forwards``
        (read its docstring for ``lanes`` / ``merge_clips``, the read-ahead and the helper thread).  What this layer adds: a
        caller's batch of MORE than ``merge_clips`` clips is cut into forwards of at most that many -- longest clips first, so
        that a forward's clips are of a kind -- and its results are put back in the caller's order: the recurrent kernel's cost
        per clip is lowest at four 16-clip tiles per window, and four forwards of 64 clips side by side fill the chip where two
        of 128 leave it to one recurrent window at a time (config 5's share, 128 x 30 s: DESIGN.md 6)."""
        import collections
        merge = self.pipeline_merge_clips if merge_clips is None else int(merge_clips)
        shapes = collections.deque()          # per caller batch: None (handed through) or (clips, [caller positions of piece k])

        def pieces():
            for b in batches:
                n = len(b)
                if merge <= 0 or n <= merge:
                    shapes.append(None)
                    yield b
                    continue
                if isinstance(b, DeviceClips):       # longest first already: consecutive parts
                    cuts = [np.arange(lo, min(lo + merge, n)) for lo in range(0, n, merge)]
                    parts = [b.part(int(c[0]), int(c[-1]) + 1) for c in cuts]
                else:
                    order = np.argsort([-len(r) for r in b], kind="stable")
                    cuts = [order[lo:lo + merge] for lo in range(0, n, merge)]
                    parts = [[b[i] for i in c] for c in cuts]
                shapes.append((n, cuts))
                for part in parts:
                    yield part

        # a sized source (a list of lists): how many pieces the pipeline will see -- it then knows when it is enqueueing its last
        # forwards and gives a forward that has the chip to itself the kernels of a lone batch (short calls: _transcribe_forwards)
        total = None
        try:
            total = sum(1 if (merge <= 0 or len(b) <= merge) else -(-len(b) // merge) for b in batches) if hasattr(batches, "__len__") else None
        except TypeError:
            total = None
        inner = self._transcribe_forwards(pieces(), show_all=show_all, lanes=lanes, merge_clips=merge_clips, total=total)
        try:
            for res in inner:
                shape = shapes.popleft()
                if shape is None:
                    yield res
                    continue
                n, cuts = shape
                whole = [None] * n
                for k, cut in enumerate(cuts):
                    part = res if k == 0 else next(inner)
                    for pos, i in enumerate(cut):
                        whole[int(i)] = part[pos]
                yield whole
        finally:
            inner.close()

    def _transcribe_forwards(self, batches, show_all=False, lanes=None, merge_clips=None, total=None):
        """Generator over ``transcribe_batch(b)`` for every ``b`` of ``batches``, software-pipelined: consecutive batches are
        merged into forwards of up to ``merge_clips`` clips (per-clip results do not depend on the batch they run in), and up to
        ``lanes`` forwards are in flight (default: ``pipeline_lanes``, or two where more do not pay: ``_lanes_that_pay``), each
        on a model handle, stream and workspaces of its own -- the latency-bound recurrent layers of several forwards run side by side on disjoint compute units while the dense kernels of the others
        fill the rest of the chip; the decoder of a finished forward runs on a side stream.  Results come out in order, one
        list per batch.  ``batches`` is read AHEAD of the results: up to ``merge_clips`` clips for the forward being put
        together, plus one forward more (host clips are copied to pinned memory and uploaded before the loop waits for the
        GPU) -- a source that produces a batch only after it has seen an earlier batch's result must pass ``lanes=1,
        merge_clips=0``: then batch k + 1 is read only after result k has been yielded, on the caller's own thread.  In every
        other mode ``batches`` is advanced on a HELPER THREAD (one, the same for the whole call), concurrently with the
        consumer's loop body: a source with thread-affine state (a GUI toolkit's objects, a thread-local CUDA stream of its own)
        must be wrapped accordingly or use the sequential mode.  Latency: with the defaults (four forwards of up to 64 clips in
        flight and one staged) a result comes out eight to ten batches of 32 clips after its batch was read."""
        import torch
        import collections
        auto_lanes = lanes is None
        lanes = self.pipeline_lanes if auto_lanes else max(1, int(lanes))
        merge_clips = self.pipeline_merge_clips if merge_clips is None else int(merge_clips)
        searching = hasattr(self.decoder, "decode_enqueue") and not getattr(self.decoder, "on_lane", False)
        # (the lanes are set up once the first forward has been put together: how many pay depends on its size)
        handles, parsers, streams = self._lanes(1)

        def set_up(count):
            hs, ps, ss = self._lanes(count)
            for h in hs:
                if hasattr(h, "set_inflight"):
                    h.set_inflight(max(2, len(hs)) if len(hs) > 1 else 1)     # (re-stated per forward in the loop below)
            for p, st in zip(ps, ss):
                p.share_copy_stream = searching      # a search kernel on the decode stream: fewer streams
                p.upload_on_compute_stream = True    # no copy stream in the pipeline: see SpectrogramAudioParser.stage
                # ... and the upload is ISSUED where the clips are staged, on the helper thread, into the lane's own stream: the first
                # copy a process hands a DMA engine holds its caller for 6-12 ms (hipMemcpyAsync creating the engine's queue; which
                # engine a copy gets depends on which are busy, so new ones are met well into a process's second call:
                # profiles/r06_second_call_stall.txt) -- the thread that feeds the other lanes must not be the one held
                p.upload_stream = st
            return hs, ps, ss
        # Depth of the pipeline in forwards.  Greedy decoding is a short host-synchronous step.  A beam search is a kernel of its
        # own that starts when its forward ends: one more job in flight (the oldest forward's search) keeps every lane's forward
        # running while the host waits for that search.
        pending, turn, count, job, done = collections.deque(), 0, 0, None, None
        end = object()
        source = iter(batches)
        held = [end, False]                      # a batch read from the source that did not fit the group being put together
        taken = [0, False]                       # batches read from the source so far; the source has ended

        def next_batch():
            if held[1]:
                held[1] = False
                return held[0]
            b = next(source, end)
            if b is end:
                taken[1] = True
            else:
                taken[0] += 1
            return b

        plan = [None, 0]                         # the call's last round, once it is known: batches per forward; forwards put together so far

        def tail_plan(first_len):
            """A sized source's LAST ROUND of forwards is dealt evenly over the lanes: 20 batches of 32 clips on four lanes are
            eight forwards of 64 clips and then four of 32 -- not ten of 64, whose last two run on two lanes while the other two
            stand empty (a 20-batch call: 115 -> @@ ms, profiles/r06_short_calls.txt).  -> batches this forward may merge, or None."""
            index, plan[1] = plan[1], plan[1] + 1
            if total is None or merge_clips <= 0 or lanes < 2 or not self.pipeline_balance_tail:
                return None
            if plan[0] is None:
                if index % lanes:                     # rounds start on the first lane
                    return None
                full = max(1, merge_clips // max(first_len, 1))
                rem = total - taken[0] + (1 if held[1] else 0) + 1           # batches not yet in a forward, this one's first included
                if rem > lanes * full:
                    return None
                k = min(lanes, rem)
                plan[0] = [rem // k + (1 if i < rem % k else 0) for i in range(k)]
            return plan[0].pop(0) if plan[0] else None

        def forwards_to_come(per_forward):
            """Forwards that will follow the one being enqueued (which merged `per_forward` batches), as far as this call can know
            WITHOUT asking the source for anything (a live source must not be waited for here): from the batch count of a sized
            source (`total`), otherwise none once the source has ended and 'plenty' before."""
            if taken[1]:
                return 1 if held[1] else 0
            if total is not None:
                left = total - taken[0] + (1 if held[1] else 0)
                return min(-(-left // max(per_forward, 1)), lanes)
            return lanes

        def fetch(parser):
            """The next forward: consecutive batches of one kind (host clips / device-resident clips) up to merge_clips clips.
            -> (parts, merged recordings, what _enqueue_batch needs of the staging) or None at the end of the source."""
            first = next_batch()
            if first is end:
                return None
            parts, nclips, on_device = [first], len(first), isinstance(first, DeviceClips)
            most = tail_plan(len(first))
            while nclips and nclips < merge_clips and (most is None or len(parts) < most):
                nxt = next_batch()
                if nxt is end:
                    break
                if isinstance(nxt, DeviceClips) != on_device or nclips + len(nxt) > merge_clips or \
                        (on_device and nxt.pcm.dtype != first.pcm.dtype):          # (one sample type per device-resident forward)
                    held[0], held[1] = nxt, True
                    break
                parts.append(nxt)
                nclips += len(nxt)
            live = [b for b in parts if len(b)]
            if not live:
                return parts, [], None
            if on_device:
                # device-resident clips: ordered behind whatever produced them on the caller's stream (the lanes' streams are
                # ordered behind nothing else); merged into one longest-first buffer on the LANE's stream, behind that point
                ahead = torch.cuda.Event()
                ahead.record(streams[0])
                return parts, (live[0] if len(live) == 1 else _UnmergedDeviceClips(live)), ahead
            merged = live[0] if len(live) == 1 else [clip for b in live for clip in b]
            staged = self._stage_batch(merged, parser) if hasattr(parser, "stage") else None
            return parts, merged, staged

        def results_of(done):
            """A finished forward -> one result list per batch it was merged from."""
            parts, job = done
            res = self._finish_batch(job, show_all) if job is not None else []
            out, lo = [], 0
            for b in parts:
                out.append(res[lo:lo + len(b)])
                lo += len(b)
            return out

        # The next forward is put together by a helper thread (reading the source, the copy into pinned memory, the upload's start)
        # while this thread waits for the GPU, makes strings and runs the caller's loop body: on a busy host the staging of 80 MB
        # is milliseconds during which a lane that has just finished would otherwise stand empty.  (One forward in flight without
        # merging = the caller asked for strictly sequential reads: inline.)
        helper = None
        if lanes > 1 or merge_clips > 0:
            from concurrent.futures import ThreadPoolExecutor
            helper = ThreadPoolExecutor(max_workers=1)
        device_index = self._device_index()

        def fetch_ahead(parser):
            if helper is None:
                return fetch(parser)

            def work():
                torch.cuda.set_device(device_index)
                torch.cuda.set_stream(streams[0])        # a source that launches GPU work does so on the caller's stream, as inline
                return fetch(parser)
            return helper.submit(work)

        try:
            set_up(1)
            ahead = fetch_ahead(parsers[0])
            group = ahead.result() if helper is not None else ahead
            if auto_lanes and group is not None:
                lanes = self._lanes_that_pay(lanes, sum(len(b) for b in group[0]))
            handles, parsers, streams = set_up(lanes)
            lanes = len(handles)
            depth = lanes + 1 if searching else lanes
            while group is not None:
                parts, merged, staged = group
                job = None
                if len(merged):
                    for older in pending:            # this forward's model handle gives back its previous forward first
                        if older[1] is not None and older[1].model is handles[turn]:
                            older[1].collect_forward()
                    # Kernel forms follow what will BE on the chip, not what the call was set up for: a forward that is enqueued
                    # with nothing else running and nothing to come (a call of one or two batches) takes the forms of a lone batch
                    # (the whole-device recurrent kernel, or two ring windows side by side); two forwards that will share the chip
                    # between them (a call of three or four batches, the last forwards of a sized source) take two ring windows
                    # each; anything more, one window each.  profiles/r06_short_calls.txt
                    if lanes > 1 and hasattr(handles[turn], "set_ring_windows"):
                        busy = sum(1 for _p, j in pending if j is not None and not j.collected and not j.model.ready())
                        expect = busy + 1 + forwards_to_come(len(parts))
                        handles[turn].set_inflight(1 if expect <= 1 else max(2, lanes))
                        handles[turn].set_ring_windows(2 if expect == 2 else 0)
                    with torch.cuda.stream(streams[turn]):
                        if isinstance(merged, _UnmergedDeviceClips):
                            streams[turn].wait_event(staged)
                            merged = DeviceClips.merge(merged.batches)
                        job = self._enqueue_batch(merged, handles[turn], parsers[turn], decode_slot=count % (depth + 1), staged=staged)
                    turn = (turn + 1) % lanes
                    count += 1
                pending.append((parts, job))
                job = None
                if helper is None:
                    # strictly sequential (lanes=1, merge_clips=0): every result is out before the source is asked for its next
                    # batch -- a source may wait for result k before it produces batch k + 1
                    while pending:
                        done = pending.popleft()
                        res = results_of(done)
                        done = None
                        for r in res:
                            yield r
                ahead = fetch_ahead(parsers[turn])   # the next forward: staged now, beside the waits below
                # (one more than `depth` may be pending for a moment: the oldest forward has been waited for above -- it ran on the
                # lane that was just refilled -- and only its strings are still to be made, while every lane is busy again)
                while len(pending) > depth:
                    done = pending.popleft()
                    res = results_of(done)
                    done = None
                    for r in res:
                        yield r
                group = ahead.result() if helper is not None else ahead
            while pending:
                done = pending.popleft()
                res = results_of(done)
                done = None
                for r in res:
                    yield r
        finally:
            if helper is not None:
                helper.shutdown(wait=True)       # (a staging in progress finishes: its pinned slot must not be refilled under it)
            # the caller stopped early, or a batch raised: whatever is still enqueued gives its forward and its beam-search
            # ticket back, otherwise the decoder handle stays "not collected" and every later call on this engine fails
            for left in [done[1] if done else None, job] + [pj[1] for pj in pending]:
                if isinstance(left, _BatchJob):
                    self._abandon(left)
            for h in handles:
                if hasattr(h, "set_inflight"):
                    h.set_inflight(1)
                if hasattr(h, "set_ring_windows"):
                    h.set_ring_windows(0)

    def _abandon(self, job):
        """Wait for an enqueued batch and drop its results."""
        for release in (job.collect_forward, (lambda: self.decoder.decode_collect(job.ticket)) if job.ticket is not None else None):
            if release is not None:
                try:
                    release()
                except Exception:      # the batch is being discarded: its own failure must not mask the caller's
                    pass

    def transcribe_device(self, pcm, n_samples, show_all=False, max_batch=None):
        """Clips that already sit back to back in GPU memory (int16 / float32 / float64, longest first), e.g. a
        shard received over RCCL (``parallel.recognize_sharded``): no host staging at all.  With ``max_batch`` the shard
        runs as a pipelined sequence of batches of at most that many clips (``transcribe_batches``)."""
        clips = DeviceClips(pcm, n_samples)
        if not max_batch or len(clips) <= max_batch:
            return self.transcribe_batch(clips, show_all=show_all) if len(clips) else []
        parts = [clips.part(lo, min(lo + max_batch, len(clips))) for lo in range(0, len(clips), max_batch)]
        return [r for res in self.transcribe_batches(parts, show_all=show_all) for r in res]

    def transcribe(self, recording, show_all=False):
        beams = self._finish_batch(self._enqueue_batch([recording]), True, warn=show_all)[0]
        return beams if show_all else beams[0]

    # ---- long recordings and files ----------------------------------------------------------------------------------
    def transcribe_long(self, recording, energy_threshold=600, step=1024, pause_threshold=0.55, phrase_threshold=0.2,
                        max_batch=32, show_all=False):
        """Long-form transcription: the energy gate of the reference's
        example_scripts/video_transcribe_simulation.py:68-143 cuts the recording into phrases (``dsmi_segment``: hop
        energies on the GPU), the phrases are transcribed in batches of at most ``max_batch`` (longest first), and
        ``[(start_sample, end_sample, transcription), ...]`` comes back in time order.  The recording is uploaded
        once; phrases are sliced on the device."""
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

    # ---- utterances that arrive in parts ----------------------------------------------------------------------------
    def enable_streaming(self, secondary_model=None, return_string_parts=True):
        if secondary_model:
            secondary_model = secondary_model.to(self.device)
            secondary_model.eval()
        self._session = _StreamingSession(secondary_model or None, bool(return_string_parts))
        self.greedy_decoder = GreedyDecoder(labels=self.labels, blank_index=self.labels.index('_'))
        self.audio_parser = InferenceSpectrogramAudioParser(audio_config=self.audio_config, device=self._device_index())

    def disable_streaming(self, keep_secondary_model=False):
        self.audio_parser = SpectrogramAudioParser(self.audio_config, device=self._device_index())
        self.greedy_decoder = None
        kept = self._session.secondary_model if (self._session and keep_secondary_model) else None
        self._session = _StreamingSession(kept, False)

    def reset_streaming_params(self):
        if self._session:
            self._session.clear()

    # attribute names of the reference object, for code that peeks at them
    secondary_model = property(lambda self: self._session.secondary_model if self._session else None)
    string_parts = property(lambda self: self._session.string_parts if self._session else False)
    iterating_transcript = property(lambda self: self._session.text if self._session else "")
    full_output = property(lambda self: self._session.outputs if self._session else [])
    spectrograms = property(lambda self: self._session.spectrograms if self._session else [])

    def _final_text(self, ses):
        """The utterance is over: the secondary model on all spectrograms, else the language-model decoder on all
        outputs, else the running greedy text."""
        import torch
        if ses.secondary_model:
            spect = torch.cat(ses.spectrograms, dim=1)
            spect = spect.view(1, 1, spect.size(0), spect.size(1)).to(self.device)
            probs, _ = ses.secondary_model(spect, torch.IntTensor([spect.size(3)]).int())
            text = self.decoder.decode(probs)[0][0][0]
        elif self.lm != "greedy":
            text = self.decoder.decode(torch.cat(ses.outputs, dim=1))[0][0][0]
        else:
            text = ses.text
        ses.clear()
        return text

    def streaming_transcribe(self, recording, is_last, is_first):
        """One part of an utterance through the streaming model.  Returns this part's text (or the whole text so far
        when string parts are off); on the first part nothing (the lookahead is filling); on the last part the final
        text of the utterance, provided more than one character was recognised."""
        ses = self._session
        spect = self.audio_parser.parse_audio(recording, is_last)
        said = ""
        if len(spect) != 0:
            if ses.secondary_model:
                ses.spectrograms.append(spect)
            spect = spect.view(1, 1, spect.size(0), spect.size(1)).to(self.device)
            probs = self.model(spect, is_first, is_last)
            if is_first:
                return ""
            ses.outputs.append(probs)
            piece = ses.extend_text(self.greedy_decoder.decode(probs)[0][0][0])
            said = piece if ses.string_parts else ses.text
        if not is_last:
            return said
        return self._final_text(ses) if len(ses.text) > 1 else ""
