/*
 * dsmi.h -- C ABI of libdsmi.so: the MI355X-native (gfx950) implementation of the
 * DanSpeech recognize() hot path.
 *
 * The reference (danspeech/danspeech, pure Python) has no FFI of its own; its only
 * boundary for this path is the Python call chain
 *     Recognizer.recognize            danspeech/Recognizer.py:82-95
 *       -> DanSpeechRecognizer.transcribe   danspeech/DanSpeechRecognizer.py:218-231
 *            -> SpectrogramAudioParser.parse_audio  danspeech/audio/parsers.py:50-72
 *            -> DeepSpeech.forward                  danspeech/deepspeech/model.py:496-515
 *            -> {Greedy,BeamCTC}Decoder.decode      danspeech/deepspeech/decoder.py:129-144,183-198
 * Each entry point below names the reference interface it replaces.  The Python
 * mirror of those classes (package danspeech_amd) binds these symbols with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success and a negative dsmi_status on failure;
 *    dsmi_last_error() gives the message of the last failure on that handle
 *    (NULL handle: last failure of a create call on this thread).
 *  - "dev" pointers are HIP device pointers on the handle's device; "host" pointers
 *    are ordinary host memory.  The caller owns every buffer passed in; the library
 *    owns weights and workspaces inside the handle.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is
 *    enqueued on it; functions that return host data synchronise that stream.
 *  - one handle = one GPU; distinct handles may be used from distinct threads, one
 *    handle is not re-entrant (same contract as a reference Recognizer instance).
 */
#ifndef DSMI_H
#define DSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libdsmi.so is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the file are its whole
 * export list (tests/test_abi.py compares `nm -D` with them). */
#if defined(DSMI_BUILD) && defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef enum {
    DSMI_OK = 0,
    DSMI_ERR_INVALID = -1,     /* bad argument / shape                               */
    DSMI_ERR_CONV = -2,        /* conv_layers outside 1..3 (reference ConvError, model.py:344-348) */
    DSMI_ERR_NOT_READY = -3,   /* tensor missing / finalize not called (ModelNotInitialized) */
    DSMI_ERR_UNSORTED = -4,    /* lengths not sorted descending (torch RuntimeError from
                                  pack_padded_sequence, model.py:117)                */
    DSMI_ERR_HIP = -5,         /* HIP runtime failure                                */
    DSMI_ERR_NOMEM = -6,
    DSMI_ERR_IO = -7,          /* LM file unreadable / malformed                     */
    DSMI_ERR_CAPACITY = -8,    /* batch/time exceeds dsmi_reserve()                  */
    DSMI_ERR_TIMEOUT = -9,     /* a persistent recurrent kernel's hand-off wait timed out and the batch could not be recomputed */
    DSMI_ERR_COMM = -10,       /* RCCL unavailable or an exchange failed (dsmi_comm_*)                                           */
    DSMI_RECOMPUTED = 1        /* dsmi_forward_status: the batch was recomputed on the per-step path; results valid now */
} dsmi_status;

enum { DSMI_RNN_GRU = 0, DSMI_RNN_LSTM = 1, DSMI_RNN_TANH = 2 };
enum { DSMI_WIN_HAMMING = 0, DSMI_WIN_HANN = 1, DSMI_WIN_BLACKMAN = 2, DSMI_WIN_BARTLETT = 3 };
/* Sample formats of dsmi_features.  The last three and DSMI_PCM_STEREO are the raw frames of a PCM WAV
 * file as danspeech.audio.load_audio reads them (resources.py:22-61): little-endian, unsigned 8-bit
 * biased by 128 (resources.py:551-554), and for two channels interleaved L,R frames folded on the
 * device into the SATURATING sum L+R of audioop.tomono(buf, width, 1, 1) (resources.py:302-303).
 * OR DSMI_PCM_STEREO into I16 / I24 / I32; sample counts and offsets are then in frames. */
enum { DSMI_PCM_I16 = 0, DSMI_PCM_F32 = 1, DSMI_PCM_F64 = 2, DSMI_PCM_U8 = 3, DSMI_PCM_I24 = 4, DSMI_PCM_I32 = 5,
       DSMI_PCM_STEREO = 16 };
enum { DSMI_PAD_REFLECT = 0, DSMI_PAD_CONSTANT = 1 };

/* Mirrors the arguments of DeepSpeech.__init__ (model.py:293-294) plus audio_conf
 * (danspeech/deepspeech/utils.py:1-8). */
typedef struct {
    int32_t conv_layers;        /* 1..3                                   */
    int32_t rnn_type;           /* DSMI_RNN_*                             */
    int32_t rnn_hidden_size;
    int32_t rnn_layers;
    int32_t bidirectional;      /* 0/1                                    */
    int32_t context;            /* Lookahead context (unidirectional)     */
    int32_t n_labels;           /* len(labels)                            */
    int32_t sample_rate;        /* audio_conf["sampling_rate"]  (fixes n_freq, model.py:340-355) */
    double  window_size;        /* audio_conf["window_size"], seconds     */
} dsmi_model_desc;

/* Mirrors AudioParser.__init__ (danspeech/audio/parsers.py:18-30, 43-48): audio_conf. */
typedef struct {
    int32_t sample_rate;        /* audio_conf["sampling_rate"]            */
    double  window_size;        /* seconds; n_fft = int(rate * size), double arithmetic as in Python */
    double  window_stride;      /* seconds; hop   = int(rate * stride)    */
    int32_t window;             /* DSMI_WIN_*                             */
    int32_t normalize;          /* 0/1                                    */
    int32_t pad_mode;           /* DSMI_PAD_*: librosa's centre padding (reflect <= 0.9, constant >= 0.10) */
} dsmi_frontend_desc;

typedef struct dsmi_model dsmi_model;        /* a DeepSpeech instance on one GPU          */
typedef struct dsmi_frontend dsmi_frontend;  /* a SpectrogramAudioParser on one GPU       */
typedef struct dsmi_decoder dsmi_decoder;    /* a GreedyDecoder / BeamCTCDecoder on one GPU */

/* ---- lifecycle: replaces DeepSpeech.__init__ / load_model (model.py:293-425, 599-624) */
int dsmi_model_create(const dsmi_model_desc* desc, int device, dsmi_model** out);
/* One call per state_dict entry, reference names ("rnns.0.rnn.weight_ih_l0_reverse", ...);
 * `data` is host float32, row-major, `shape[ndim]`.  Unknown names are ignored
 * (num_batches_tracked). */
int dsmi_model_load_tensor(dsmi_model* m, const char* name, const float* data_host,
                           const int64_t* shape, int ndim);
/* Checks completeness, repacks weights into kernel layouts, uploads.  After this the
 * handle is immutable apart from workspaces. */
int dsmi_model_finalize(dsmi_model* m);
/* Pre-sizes workspaces so that no allocation happens on the timed path for
 * batches <= max_batch of <= max_frames spectrogram frames. */
int dsmi_reserve(dsmi_model* m, int max_batch, int max_frames);
void dsmi_model_destroy(dsmi_model* m);
const char* dsmi_last_error(const dsmi_model* m);

/* ---- DeepSpeech.get_seq_lens (model.py:540-551); pure host arithmetic */
int dsmi_seq_lens(const dsmi_model* m, const int32_t* lens_host, int n, int32_t* out_lens_host);

/* ---- SpectrogramAudioParser (parsers.py:37-72), batched.
 * pcm_dev: B clips back to back, clip b has n_samples_host[b] samples starting at
 * sample offset sum(n_samples_host[:b]); dtype DSMI_PCM_* (float arrays as load_audio returns
 * them, or a WAV file's raw frames: the file never has to be decoded on the host).
 * feat_dev: [B][n_freq][t_stride] float32, frames past a clip's own count are zero.
 * frames_host[b] = 1 + n_samples[b] / hop.  Asynchronous on `stream`. */
int dsmi_frontend_create(const dsmi_frontend_desc* desc, int device, dsmi_frontend** out);
void dsmi_frontend_destroy(dsmi_frontend* f);
const char* dsmi_frontend_last_error(const dsmi_frontend* f);
int dsmi_features(dsmi_frontend* f, const void* pcm_dev, int pcm_dtype, const int64_t* n_samples_host,
                  int B, float* feat_dev, int t_stride, int32_t* frames_host, void* stream);

/* ---- InferenceSpectrogramAudioParser.parse_audio (parsers.py:102-164), the arithmetic half: STFT of the
 * samples WITHOUT centre padding (librosa.stft(center=False), :137-138: 1 + (n - n_fft)/hop frames), log1p|.|,
 * then the adaptive normalisation of :146-161.  state3 = {input_mean, input_std, alpha} is read and updated
 * exactly as the parser's attributes are (alpha += 0.1; running mean/std halved with the chunk's np.mean /
 * np.std; while alpha < 1 they are mixed with the NST dataset statistics 5.492418704733003 /
 * 1.7552755216970917, :89-94).  The sample bookkeeping of :112-133 (hop carry-over) stays with the caller.
 * feat_dev [n_freq][t_stride]; *frames_host = frames written.  Synchronous (the statistics pass through
 * the host).  Start an utterance with state3 = {0, 0, 0} (parser.reset(), :166-170). */
/* Host only.  load_audio hands recognize() float64 samples (reference resources.py:640) that are, for every audio FILE, integers in
 * int16's range: such a clip travels to the device as int16 -- a quarter of the bytes, the same features bit for bit (dsmi_features
 * widens to float64 exactly).  Returns 1 when every one of the n samples is an integer in [-32768, 32767] and dst holds them, 0
 * when one is not (dst is then unspecified and the caller uploads the float64 samples).  Used by the staging of
 * dsmi_recognize_enqueue and of the Python pipeline. */
int dsmi_pack_pcm_i16(const double* src, int64_t n, int16_t* dst);
/* Staged clips -> device memory by a KERNEL that reads the pinned host buffer over the bus (src_pinned: hipHostMalloc'ed / pinned by
 * the caller's framework, i.e. mapped into the device's address space; bytes a multiple of 2), asynchronous on `stream`.  The
 * batch pipeline uploads with it instead of hipMemcpyAsync: the runtime hands a copy to a DMA engine, and the first copies a
 * process's streams give the engines hold the calling thread for 6-12 ms each -- well into a process's second call
 * (profiles/r06_second_call_stall.txt).  device: where dst_dev lives. */
int dsmi_upload(int device, void* dst_dev, const void* src_pinned, int64_t bytes, void* stream);
int dsmi_features_stream(dsmi_frontend* f, const void* pcm_dev, int pcm_dtype, int64_t n_samples, double* state3,
                         float* feat_dev, int t_stride, int32_t* frames_host, void* stream);

/* ---- Chunked unidirectional inference: DeepSpeech(streaming_inference_model=True).streaming_forward
 * (model.py:517-537) with the state MaskConvStream (:156-201), BatchRNNStream (:204-238) and LookaheadStream
 * (:241-283) carry between calls.  One dsmi_stream = one utterance in flight (B = 1) on a unidirectional
 * 2-conv model (the only streaming shape the reference can run: streaming_init, :427-494); a model serves any
 * number of streams.  feat_dev [n_freq][T] float32 (one chunk of the streaming parser's output);
 * probs_dev [T_out_cap][n_labels].  *T_out = frames written: 0 on the first pass (is_first), when the
 * lookahead only buffers (streaming_forward returns None, :529-530).  is_last flushes the lookahead with its
 * right padding and clears all carried state.  Asynchronous on `stream`. */
typedef struct dsmi_stream dsmi_stream;
int dsmi_stream_create(dsmi_model* m, dsmi_stream** out);
void dsmi_stream_destroy(dsmi_stream* s);
const char* dsmi_stream_last_error(const dsmi_stream* s);
int dsmi_stream_reset(dsmi_stream* s);
int dsmi_stream_forward(dsmi_stream* s, const float* feat_dev, int T, int is_first, int is_last, float* probs_dev,
                        int T_out_cap, int32_t* T_out, void* stream);

/* ---- Offline long-form segmentation: the energy gate of
 * example_scripts/video_transcribe_simulation.py:68-143 over one recording.
 * Hop i covers samples [i*step, (i+1)*step) for every i with (i+1)*step < n_samples (:94); its energy is
 * sqrt(sum(x*x)/step) in float64 (:100), summed in numpy's pairwise order so that comparisons with
 * energy_threshold fall exactly as they do in the script.  A phrase starts at the first hop above the
 * threshold, two hops early when that is not negative (:103-113), ends once more than pause_hops
 * consecutive hops stay at or below it (:119-128), and is reported when it held more than phrase_hops
 * hops apart from that pause (:131-132).  A phrase still open at the end of the audio is dropped, as in
 * the script.  step must be 128 * 2^k.  seg_start/seg_end_host[max_segments] receive sample ranges
 * [start, end); *n_segments the number found (DSMI_ERR_CAPACITY if more than max_segments).
 * energies_host (optional, one float64 per hop) receives the hop energies.  Synchronous. */
int dsmi_segment(dsmi_frontend* f, const void* pcm_dev, int pcm_dtype, int64_t n_samples, int step,
                 double energy_threshold, int pause_hops, int phrase_hops,
                 int64_t* seg_start_host, int64_t* seg_end_host, int max_segments, int* n_segments,
                 double* energies_host, void* stream);

/* ---- DeepSpeech.forward (model.py:496-515), eval mode.
 * feat_dev [B][1][n_freq][T] float32 (T = max frames, zero past each clip's length),
 * lens_host[B] sorted descending.  probs_dev [B][T_out][n_labels] float32 softmax
 * probabilities where T_out = seq_lens(T); out_lens_host[B]. */
int dsmi_forward(dsmi_model* m, const float* feat_dev, const int32_t* lens_host, int B, int T,
                 float* probs_dev, int32_t* out_lens_host, void* stream);
/* dsmi_forward is asynchronous on `stream`; dsmi_forward_status blocks until the handle's OLDEST dsmi_forward whose
 * status has not been collected yet has finished, and says whether its results are valid (the reference's forward is
 * synchronous and cannot fail this way: torch's CPU kernels do not depend on co-residency).  Call it once per forward,
 * in order, before consuming that forward's probs_dev; its feat_dev, probs_dev and stream must still be alive.  Up to 4
 * forwards may be in flight uncollected (a pipelining caller enqueues batch i+1 before collecting batch i).
 *   DSMI_OK          results valid.
 *   DSMI_RECOMPUTED  a hand-off wait inside the persistent recurrent kernel timed out (not all of its workgroups were
 *                    resident: another kernel or process occupied compute units).  The SAME batch has been recomputed
 *                    with one launch per time step into the same probs_dev before this call returned: results valid
 *                    now, and the handle keeps using the per-step path.  dsmi_last_error() holds a description.
 *   < 0              the recompute itself failed.
 * Uncollected forwards that finished well are forgotten by the next dsmi_forward; one that finished with a timeout
 * makes the next dsmi_forward fail with DSMI_ERR_TIMEOUT (its results were invalid and may have been consumed). */
int dsmi_forward_status(dsmi_model* m);
/* 1 when the handle's oldest uncollected dsmi_forward has finished on the device (dsmi_forward_status would not block), 0 when
 * it is still running, < 0 on error; never blocks.  For a host that keeps several handles busy and refills whichever
 * finishes first (no reference counterpart). */
int dsmi_forward_ready(dsmi_model* m);
/* Tells the handle how many batches the caller keeps in flight on this device (each on its own handle and stream).
 * 1 (default): kernels chosen for the latency of one batch (one 16-clip tile per workgroup, the whole device).  2: the
 * recurrent layers use the paired-tile variant (a workgroup carries both tiles of a 17..32-clip batch: 100 CUs for
 * BASELINE's 5 x BiGRU 800) where the shape allows it, so that the two batches' recurrent layers run side by side on
 * disjoint halves of the chip.  Results are the same either way (within the parity bound). */
int dsmi_model_set_inflight(dsmi_model* m, int batches);
/* How many ring windows (rnn_persist_ring*.hip: H / 32 workgroups per direction each, 50 CUs for 5 x BiGRU 800) the recurrent layers of
 * this handle's NEXT forwards take side by side.  0 (default): what set_inflight implies -- one window with batches in flight, as
 * many as the batch has tile pairs (up to four) for a lone batch.  2: a caller that knows that only two forwards will share the
 * chip (the last forwards of a short call) gives each two windows of half the tiles: 2.7 instead of 3.2 ms per 64-clip layer.
 * Shapes the ring kernels do not take ignore it.  Results are the same either way (within the parity bound). */
int dsmi_model_set_ring_windows(dsmi_model* m, int windows);
/* Number of batches / layers this handle had to recompute after a hand-off timeout (0 in normal operation). */
int dsmi_recompute_count(const dsmi_model* m);

/* Stage-level entry points (same arithmetic as inside dsmi_forward), used by the
 * parity tests against the reference's MaskConv (model.py:65-81) and BatchRNN
 * (model.py:114-122) golden vectors.
 * conv_out_dev: [B][C_out*F_out][T_out] float32.
 * rnn layer: x_dev/y_dev are [T][B][I] / [T][B][H] float32 (reference T x N x H). */
int dsmi_conv_stack(dsmi_model* m, const float* feat_dev, const int32_t* lens_host, int B, int T,
                    float* conv_out_dev, void* stream);
int dsmi_rnn_layer(dsmi_model* m, int layer, const float* x_dev, const int32_t* out_lens_host,
                   int B, int T_out, float* y_dev, void* stream);

/* ---- Decoder.__init__ (decoder.py:35-43): labels as n_labels UTF-8 strings (labels may be
 * multi-byte, e.g. the Danish letters), blank_index as DanSpeechRecognizer passes it
 * (labels.index('_'), DanSpeechRecognizer.py:92,94). */
int dsmi_decoder_create(int device, const char* const* labels_utf8, int n_labels, int blank_index,
                        dsmi_decoder** out);
void dsmi_decoder_destroy(dsmi_decoder* d);
const char* dsmi_decoder_last_error(const dsmi_decoder* d);

/* ---- GreedyDecoder.decode (decoder.py:183-198 + 151-181).
 * probs_dev [B][T_out][n_labels]; sizes_host[B] or NULL (= T_out for all).
 * ids_host/offsets_host: [B][T_out] int32, first n_out_host[b] entries valid.  Synchronises. */
int dsmi_greedy(dsmi_decoder* d, const float* probs_dev, const int32_t* sizes_host, int B, int T_out,
                int32_t* ids_host, int32_t* offsets_host, int32_t* n_out_host, void* stream);
/* The same call in two halves for a host that keeps batches in flight (no reference counterpart: GreedyDecoder.decode is a
 * host loop, decoder.py:166-198).  _enqueue launches the kernel and the copies of its results on `stream` -- behind the
 * dsmi_forward that writes probs_dev -- and returns at once; _collect waits for them and fills the arrays of dsmi_greedy.
 * One decode per handle at a time. */
int dsmi_greedy_enqueue(dsmi_decoder* d, const float* probs_dev, const int32_t* sizes_host, int B, int T_out, void* stream);
int dsmi_greedy_collect(dsmi_decoder* d, int32_t* ids_host, int32_t* offsets_host, int32_t* n_out_host);

/* ---- measurement hooks (no reference counterpart; SURVEY 5 "tracing/profiling": none).
 * level 0: off.  level 1: per-stage hipEvents around the last dsmi_forward (synchronises).
 * level 2: sampled launches of each kernel kind are dispatched with their own begin/end
 * timestamps (hipExtLaunchKernelGGL), fully asynchronous; read back with dsmi_kernel_stats.
 * Sampled = every launch of the recurrent kinds (6, 10), every fourth of the others.
 * stage for dsmi_stage_time_us: 0 conv, 2 input GEMMs + recurrent steps, 3 head, 4 total. */
int dsmi_set_profiling(dsmi_model* m, int level);
/* kind: 0 stft, 1 conv1, 2 conv2, 3 conv3, 4 layer-0 input GEMM, 5 input GEMM (layers >= 1),
 * 6 recurrent step (per-step path), 7 head, 8 greedy, 9 beam, 10 persistent recurrent layer.  launches = dispatches since the last reset,
 * samples = how many of them were timed, avg_us = mean duration of the timed ones,
 * flops/bytes_per_launch = algorithmic work (SURVEY 8d formulas) averaged over all launches. */
int dsmi_kernel_stats(dsmi_model* m, int kind, int64_t* launches, int64_t* samples, double* avg_us,
                      double* flops_per_launch, double* bytes_per_launch);
int dsmi_reset_kernel_stats(dsmi_model* m);
/* Diagnostics build of the recurrent step kernel: per-wave phase timestamps (100 MHz
 * s_memrealtime ticks) of the launch for `step` of `layer`; stamps_host[D*nwg][8][8]. */
int dsmi_debug_persist_stamps(dsmi_model* m, int layer, int B, int T_out, uint64_t* stamps_host, int64_t n_words);
int dsmi_debug_step_stamps(dsmi_model* m, int layer, int B, int T_out, int step, uint64_t* stamps_host,
                           int64_t n_words);
double dsmi_stage_time_us(const dsmi_model* m, int stage);
/* Kernel launches the last dsmi_forward issued for stage 2 (recurrent steps) and their
 * summed algorithmic FLOPs (SURVEY 8d formula, recurrent part), for roofline maths. */
int dsmi_last_forward_stats(const dsmi_model* m, int64_t* n_step_launches, double* step_flops,
                            double* total_flops);

/* ---- BeamCTCDecoder (decoder.py:91-144): what ctcdecode.CTCBeamDecoder(labels, lm_path, alpha,
 * beta, cutoff_top_n, cutoff_prob, beam_width, num_processes, blank_index).decode(probs, sizes)
 * does for the reference (third-party, not in the reference tree; restated in oracle/beam.py).
 * dsmi_decoder_set_lm: lm_path = an n-gram model of order <= 6 -- a KenLM binary (.klm, what the reference's
 * language_models.* factories return, danspeech/language_models/dsl_3gram.py:16-20: data structures `probing` and
 * `trie`, unquantised) or ARPA text -- or NULL/"" for none; the file type is told by its first bytes.
 * dsmi_beam outputs, all host: tokens/tsteps [B][beam][T_out] int32 (token ids and the frame
 * of each token's strongest emission), lens [B][beam], scores [B][beam] (ctcdecode's
 * "-approx_ctc", lower is better; beams are ordered best first).  Synchronises. */
int dsmi_decoder_set_lm(dsmi_decoder* d, const char* lm_path, double alpha, double beta);
int dsmi_beam(dsmi_decoder* d, const float* probs_dev, const int32_t* sizes_host, int B, int T_out,
              int beam_width, int cutoff_top_n, double cutoff_prob, int32_t* tokens_host,
              int32_t* tsteps_host, int32_t* lens_host, float* scores_host, void* stream);
/* The two halves of dsmi_beam, for callers that keep the GPU busy meanwhile (ctcdecode decodes on a host thread pool while
 * the caller waits; here the search itself is a kernel): dsmi_beam_enqueue launches the search asynchronously on `stream`
 * (probs_dev must stay valid until the collect); dsmi_beam_collect waits for it, copies the results (on that stream, into
 * pinned memory of the handle) and fills the arrays.  One search per decoder handle at a time.
 * dsmi_decoder_beam_stats: how often the last collected search left its fast path, summed over the batch -- counts4 =
 * {dormant prefixes that re-entered the beam, node-pool hops walked for them, exact rankings of a threshold bin from the
 * list, exact rankings against all candidates}; diagnostics for the tests and profiles, no reference counterpart. */
int dsmi_beam_enqueue(dsmi_decoder* d, const float* probs_dev, const int32_t* sizes_host, int B, int T_out,
                      int beam_width, int cutoff_top_n, double cutoff_prob, void* stream);
int dsmi_beam_collect(dsmi_decoder* d, int32_t* tokens_host, int32_t* tsteps_host, int32_t* lens_host, float* scores_host);
int dsmi_decoder_beam_stats(const dsmi_decoder* d, int32_t* counts4);
/* Diagnostics: phase boundaries of the last collected search's first utterance, 64 frames from the middle of the clip x 8 stamps
 * (100 MHz ticks; 0 = frame start, 1..6 = after the frame's six barriers); tools/beam_stamps.py prints the anatomy. */
int dsmi_debug_beam_stamps(const dsmi_decoder* d, uint64_t* stamps_host, int64_t n_words);

/* ---- Host-only view of a language model file (no GPU involved): what dsmi_decoder_set_lm would load.
 * kind: 0 ARPA text, 1 KenLM probing binary, 2 KenLM trie binary.  Word ids are the file's own (KenLM's WordIndex for
 * binaries, <unk> = 0).  dsmi_lm_lookup: 1 = the n-gram ids[0..n) is in the model (its log10 probability and back-off
 * weight are returned), 0 = absent.  dsmi_lm_cond_log10 = log10 p(ids[n-1] | ids[0..n-1)) by back-off, the quantity the
 * beam search's scorer adds (ctcdecode Scorer::get_log_cond_prob / KenLM BaseScore). */
typedef struct dsmi_lm dsmi_lm;
int dsmi_lm_open(const char* path, dsmi_lm** out);
void dsmi_lm_close(dsmi_lm* lm);
const char* dsmi_lm_last_error(const dsmi_lm* lm);
int dsmi_lm_info(const dsmi_lm* lm, int* order, int64_t* vocab_size, int* kind);
int dsmi_lm_word_index(const dsmi_lm* lm, const char* word_utf8);
int dsmi_lm_lookup(const dsmi_lm* lm, const int32_t* ids, int n, float* log10_prob, float* log10_backoff);
double dsmi_lm_cond_log10(const dsmi_lm* lm, const int32_t* ids, int n);

/* ---- What a handle was built with (a host that received handles from elsewhere, and dsmi_session_create). */
int dsmi_model_info(const dsmi_model* m, dsmi_model_desc* desc, int* device);
int dsmi_frontend_info(const dsmi_frontend* f, int* n_freq, int* hop, int* device);
int dsmi_decoder_info(const dsmi_decoder* d, int* n_labels, int* blank_index, int* device);
const char* dsmi_decoder_label(const dsmi_decoder* d, int index);          /* labels[index] as UTF-8, NULL out of range */

/* ---- Recognizer.recognize (Recognizer.py:158-189) -> DanSpeechRecognizer.transcribe (DanSpeechRecognizer.py:191-231)
 * over a batch of recordings, as ONE call for hosts without the Python layer: stage + upload the clips, spectrograms,
 * network, decoder, label strings, results in the caller's order -- the sequence of dsmi_features, dsmi_forward,
 * dsmi_forward_status, dsmi_greedy / dsmi_beam calls danspeech_amd/DanSpeechRecognizer.py issues.  A session binds one
 * frontend, one model and one decoder of the same device and owns the buffers between the stages and a stream; the
 * handles must outlive it.  One batch per session at a time; a host that keeps two batches in flight uses two sessions
 * (each with its own model and frontend handle, dsmi_model_set_inflight(2) on both) and alternates
 * dsmi_recognize_enqueue / dsmi_recognize_collect between them.
 *   clips_host[b]      clip b's samples in host memory, n_samples_host[b] of them (frames for DSMI_PCM_STEREO), any order
 *   beam_width         0: greedy (a recogniser without language model, DanSpeechRecognizer.py:94); > 0: beam search with
 *                      the decoder's dsmi_decoder_set_lm settings, best beam returned
 *   text_utf8          [B][text_stride]: clip b's transcript, NUL-terminated; one longer than text_stride - 1 bytes is cut
 *                      at a label boundary and text_bytes_host[b] (optional) holds its full length
 *   scores_host        optional [B]: the best beam's score (0 for greedy)
 * Returns DSMI_OK, DSMI_RECOMPUTED (results valid, see dsmi_forward_status) or < 0. */
typedef struct dsmi_session dsmi_session;
int dsmi_session_create(dsmi_frontend* f, dsmi_model* m, dsmi_decoder* d, dsmi_session** out);
void dsmi_session_destroy(dsmi_session* s);
const char* dsmi_session_last_error(const dsmi_session* s);
int dsmi_recognize_batch(dsmi_session* s, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int B,
                         int beam_width, int cutoff_top_n, double cutoff_prob,
                         char* text_utf8, int text_stride, int32_t* text_bytes_host, float* scores_host);
/* The two halves: everything up to the probabilities, asynchronous; then wait + decode + strings. */
int dsmi_recognize_enqueue(dsmi_session* s, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int B);
/* Clips already back to back in device memory, LONGEST FIRST (DSMI_ERR_UNSORTED otherwise) -- a shard as
 * dsmi_comm_scatter delivers it: no host staging.  pcm_dev must stay valid until the collect; results are in that order. */
int dsmi_recognize_enqueue_device(dsmi_session* s, const void* pcm_dev, const int64_t* n_samples_host, int pcm_dtype, int B);
int dsmi_recognize_collect(dsmi_session* s, int beam_width, int cutoff_top_n, double cutoff_prob,
                           char* text_utf8, int text_stride, int32_t* text_bytes_host, float* scores_host);

/* ---- Utterance-level data parallelism, one process (or host thread) per GPU, for hosts without torch.distributed
 * (the Python layer: danspeech_amd/parallel.py).  The reference has no distributed code; clips are independent
 * on this path, weights are replicated, and the only exchanges are the input scatter and the result gather: grouped
 * ncclSend / ncclRecv between the root and the other ranks over RCCL, which is bound at run time (dlopen; DSMI_RCCL_LIBRARY
 * overrides the library name) so that libdsmi.so has no load-time dependency on it.
 * dsmi_plan_shards (host only): clips sorted by length, descending and stable, dealt round-robin -- clip i goes to rank
 * rank_of[i] as that rank's slot_of[i]-th clip, so every shard is longest first with a similar length mix.
 * dsmi_comm_unique_id: rank 0 creates the 128-byte id and ships it to the other ranks by the host's own means.
 * dsmi_comm_scatter: the root passes its clips (the other ranks NULL / 0); every rank receives *shard_count clips back to
 * back in device memory at *shard_pcm_dev (library-owned, valid until the next scatter), their lengths and their
 * positions in the root's list (at most shard_cap), the sample type and the batch's total clip count.
 * dsmi_comm_gather_text: every rank passes its shard's transcripts [shard_count][text_stride] with those positions; the
 * root's all_text [total_count][text_stride] receives them in the root's order.  Both synchronise `stream`. */
typedef struct dsmi_comm dsmi_comm;
int dsmi_plan_shards(const int64_t* n_samples, int n, int world, int32_t* rank_of, int32_t* slot_of);
int dsmi_comm_unique_id(void* id128);
int dsmi_comm_init(const void* id128, int rank, int world, int device, dsmi_comm** out);
void dsmi_comm_destroy(dsmi_comm* c);
const char* dsmi_comm_last_error(const dsmi_comm* c);
int dsmi_comm_scatter(dsmi_comm* c, int root, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int n,
                      const void** shard_pcm_dev, int64_t* shard_n_samples, int32_t* shard_index, int shard_cap,
                      int* shard_count, int* shard_dtype, int* total_count, void* stream);
int dsmi_comm_gather_text(dsmi_comm* c, int root, const char* text, int text_stride, const int32_t* shard_index, int shard_count,
                          int total_count, char* all_text, void* stream);

#if defined(DSMI_BUILD) && defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DSMI_H */
