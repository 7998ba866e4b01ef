// Decoders on the GPU (gfx950): the handle behind dsmi_decoder*, greedy decode, and the launcher / collector of the CTC
// prefix beam search with an optional word-level n-gram scorer (the search kernel itself: beam_kernel.inc).
//
// Replaces GreedyDecoder.decode and BeamCTCDecoder.decode of the reference
// (danspeech/deepspeech/decoder.py:183-198, 129-144).  The reference's beam search is the
// un-vendored third-party ctcdecode (C++, thread pool over the batch); here one workgroup
// decodes one utterance, the batch runs in parallel across CUs, and nothing leaves the GPU
// until the final beams.  Algorithm and the deliberate float64 carry: oracle/beam.py restates ctcdecode's published
// algorithm, oracle/beam_flat.py the kernel's own formulation of it (implicit prefix trie held on chip, edge tuples, one-pass
// histogram selection); the parity tests compare with both.  DESIGN.md 4 "Beam search" describes the frame's six phases.
#include "common.h"
#include <cstring>
#include <mutex>
#include "lm.h"
#include "lm.cpp.inc"
#include "lm_klm.cpp.inc"

#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace dsmi;

#define DSMI_WAIT_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#include "beam_kernel.inc"

// ------------------------------------------------------------------------------------------------
struct dsmi_decoder {
    int device = 0;
    std::vector<std::string> labels;
    int blank = 0, space = -2;
    std::string err;
    // greedy scratch
    size_t greedy_cap = 0;
    int32_t *g_raw = nullptr, *g_ids = nullptr, *g_offs = nullptr, *g_nout = nullptr, *g_sizes = nullptr;
    // dsmi_greedy_enqueue / _collect: pinned host images of the results and of the sizes, the event behind the last copy
    int32_t* gh = nullptr; size_t gh_cap = 0; hipEvent_t g_done = nullptr; bool greedy_pending = false; int gp_B = 0, gp_T = 0;
    // LM
    bool has_lm = false;
    HostLM lm;
    double alpha = 0, beta = 0;
    LmEntry* d_tab = nullptr; int32_t *d_next = nullptr, *d_word = nullptr;
    unsigned char* d_klm = nullptr;      // device copy of a KenLM probing binary's search memory
    // beam workspace
    size_t ws_bytes = 0;
    unsigned char* ws = nullptr;
    // the beam search between dsmi_beam_enqueue and dsmi_beam_collect: geometry, pinned copies of its outputs, completion event
    bool beam_pending = false;
    int pb_B = 0, pb_To = 0, pb_beam = 0;
    size_t pb_out = 0, pb_out_bytes = 0; hipStream_t pb_stream = nullptr;
    int32_t stats[4] = {0, 0, 0, 0};      // of the last collected search: see dsmi_decoder_beam_stats
    uint64_t stamps[64 * 8] = {0};
    int32_t* pin_sz = nullptr; size_t pin_sz_cap = 0;
    unsigned char* pin = nullptr; size_t pin_bytes = 0;
    hipEvent_t beam_done = nullptr;
    hipStream_t copy_stream = nullptr;      // the collect's device-to-host copies: the device's collect stream (collect_stream)
};

// ONE stream per device for the device-to-host copies of every decoder handle's collect, made at the first collect and kept for
// the life of the process.  (It was a stream per handle: a pipeline with a beam search keeps six decoder handles, and the runtime
// deals every stream of a process onto GPU_MAX_HW_QUEUES hardware queues in turn -- a collect whose stream shares a queue with a
// lane's stream waits behind that lane's 40-ms forward.  The copies are host-synchronous and short: one stream serves them all.)
static hipStream_t collect_stream(int device) {
    static std::mutex mu;
    static hipStream_t per_device[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) return nullptr;
    if (!per_device[device]) {
        // highest priority: the runtime keeps the queues of a priority apart from the others', so this stream shares none with a lane
        int lo = 0, hi = 0;
        (void)hipSetDevice(device);
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        if (hipStreamCreateWithPriority(&per_device[device], hipStreamNonBlocking, hi) != hipSuccess) per_device[device] = nullptr;
    }
    return per_device[device];
}

static thread_local std::string g_dec_error;

#define DEC_HIP(d, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { (d)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return DSMI_ERR_HIP; } \
    } while (0)

extern "C" int dsmi_decoder_create(int device, const char* const* labels, int n_labels, int blank, dsmi_decoder** out) {
    if (!labels || !out || n_labels < 1 || n_labels > 128 || blank < 0 || blank >= n_labels) { g_dec_error = "bad decoder arguments"; return DSMI_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_dec_error = "no such HIP device"; return DSMI_ERR_HIP; }
    dsmi_decoder* d = new dsmi_decoder();
    d->device = device;
    d->blank = blank;
    for (int i = 0; i < n_labels; ++i) {
        d->labels.push_back(labels[i] ? labels[i] : "");
        if (d->labels.back() == " ") d->space = i;     // Decoder.__init__, decoder.py:39-42
    }
    *out = d;
    return DSMI_OK;
}

static void free_lm(dsmi_decoder* d) {
    if (d->d_tab) (void)hipFree(d->d_tab);
    if (d->d_next) (void)hipFree(d->d_next);
    if (d->d_word) (void)hipFree(d->d_word);
    if (d->d_klm) (void)hipFree(d->d_klm);
    d->d_tab = nullptr; d->d_next = nullptr; d->d_word = nullptr; d->d_klm = nullptr;
    d->has_lm = false;
    d->lm = HostLM();
}

extern "C" void dsmi_decoder_destroy(dsmi_decoder* d) {
    if (d && d->pin) { (void)hipSetDevice(d->device); (void)hipDeviceSynchronize(); (void)hipHostFree(d->pin); d->pin = nullptr; }
    if (d && d->beam_done) { (void)hipEventDestroy(d->beam_done); d->beam_done = nullptr; }
    if (d) d->copy_stream = nullptr;        // (the device's collect stream: not the handle's to destroy)
    if (d && d->pin_sz) { (void)hipHostFree(d->pin_sz); d->pin_sz = nullptr; }
    if (d && d->gh) { (void)hipSetDevice(d->device); (void)hipDeviceSynchronize(); (void)hipHostFree(d->gh); d->gh = nullptr; }
    if (d && d->g_done) { (void)hipEventDestroy(d->g_done); d->g_done = nullptr; }
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)d->g_raw, (void*)d->g_ids, (void*)d->g_offs, (void*)d->g_nout, (void*)d->g_sizes, (void*)d->ws}) if (p) (void)hipFree(p);
    free_lm(d);
    delete d;
}

extern "C" const char* dsmi_decoder_last_error(const dsmi_decoder* d) { return d ? d->err.c_str() : g_dec_error.c_str(); }

extern "C" int dsmi_decoder_info(const dsmi_decoder* d, int* n_labels, int* blank_index, int* device) {
    if (!d) return DSMI_ERR_INVALID;
    if (n_labels) *n_labels = (int)d->labels.size();
    if (blank_index) *blank_index = d->blank;
    if (device) *device = d->device;
    return DSMI_OK;
}

extern "C" const char* dsmi_decoder_label(const dsmi_decoder* d, int index) {
    return d && index >= 0 && index < (int)d->labels.size() ? d->labels[(size_t)index].c_str() : nullptr;
}

extern "C" int dsmi_decoder_set_lm(dsmi_decoder* d, const char* path, double alpha, double beta) {
    if (!d) return DSMI_ERR_INVALID;
    DEC_HIP(d, hipSetDevice(d->device));
    DEC_HIP(d, hipDeviceSynchronize());
    free_lm(d);
    d->alpha = alpha; d->beta = beta;
    if (!path || !*path) return DSMI_OK;
    const std::string msg = d->lm.load(path, d->labels);         // ARPA text or KenLM binary (probing / trie)
    if (!msg.empty()) { d->lm = HostLM(); d->err = msg; return DSMI_ERR_IO; }
    if (d->lm.order > kMaxOrder) { d->err = "n-gram order above 6 is not supported"; d->lm = HostLM(); return DSMI_ERR_IO; }
    if (d->lm.kind == 1) {
        DEC_HIP(d, hipMalloc((void**)&d->d_klm, d->lm.klm_blob.size()));
        DEC_HIP(d, hipMemcpy(d->d_klm, d->lm.klm_blob.data(), d->lm.klm_blob.size(), hipMemcpyHostToDevice));
    }
    DEC_HIP(d, hipMalloc((void**)&d->d_tab, sizeof(LmEntry) * d->lm.table.size()));
    DEC_HIP(d, hipMalloc((void**)&d->d_next, sizeof(int32_t) * d->lm.trie_next.size()));
    DEC_HIP(d, hipMalloc((void**)&d->d_word, sizeof(int32_t) * d->lm.trie_word.size()));
    DEC_HIP(d, hipMemcpy(d->d_tab, d->lm.table.data(), sizeof(LmEntry) * d->lm.table.size(), hipMemcpyHostToDevice));
    DEC_HIP(d, hipMemcpy(d->d_next, d->lm.trie_next.data(), sizeof(int32_t) * d->lm.trie_next.size(), hipMemcpyHostToDevice));
    DEC_HIP(d, hipMemcpy(d->d_word, d->lm.trie_word.data(), sizeof(int32_t) * d->lm.trie_word.size(), hipMemcpyHostToDevice));
    d->has_lm = true;
    return DSMI_OK;
}

static int greedy_reserve(dsmi_decoder* d, int B, int To) {
    if ((size_t)B * To > d->greedy_cap) {
        DEC_HIP(d, hipDeviceSynchronize());
        for (void* p : {(void*)d->g_raw, (void*)d->g_ids, (void*)d->g_offs, (void*)d->g_nout, (void*)d->g_sizes}) if (p) (void)hipFree(p);
        d->greedy_cap = (size_t)B * To;
        DEC_HIP(d, hipMalloc((void**)&d->g_raw, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_ids, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_offs, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_nout, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_sizes, sizeof(int32_t) * d->greedy_cap));
    }
    return DSMI_OK;
}

extern "C" int dsmi_greedy(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To,
                           int32_t* ids, int32_t* offsets, int32_t* n_out, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    if (!probs || !ids || !offsets || !n_out || B < 1 || To < 1) { d->err = "bad greedy arguments"; return DSMI_ERR_INVALID; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = greedy_reserve(d, B, To))) return rc;
    if (sizes) DEC_HIP(d, hipMemcpyAsync(d->g_sizes, sizes, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    launch_greedy(probs, sizes ? d->g_sizes : nullptr, B, To, (int)d->labels.size(), d->blank, d->g_raw, d->g_ids, d->g_offs, d->g_nout, s);
    DEC_HIP(d, hipMemcpyAsync(ids, d->g_ids, sizeof(int32_t) * (size_t)B * To, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(offsets, d->g_offs, sizeof(int32_t) * (size_t)B * To, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(n_out, d->g_nout, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipStreamSynchronize(s));
    DEC_HIP(d, hipGetLastError());
    return DSMI_OK;
}

// The same in two halves for a caller that keeps batches in flight: _enqueue launches the kernel and the copies of its results
// (into pinned memory of the handle's own) on `stream` -- behind the forward that writes `probs` -- and returns at once;
// _collect waits for them and hands the arrays over.  The host never stands in a copy queue of a busy device while a stream it
// could be feeding runs dry.
extern "C" int dsmi_greedy_enqueue(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    if (!probs || B < 1 || To < 1) { d->err = "bad greedy arguments"; return DSMI_ERR_INVALID; }
    if (d->greedy_pending) { d->err = "the previous greedy decode has not been collected"; return DSMI_ERR_INVALID; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = greedy_reserve(d, B, To))) return rc;
    const size_t rows = (size_t)B * To, need = 2 * rows + 2 * (size_t)B;         // ids, offsets, n_out, sizes
    if (need > d->gh_cap) {
        DEC_HIP(d, hipDeviceSynchronize());
        if (d->gh) (void)hipHostFree(d->gh);
        d->gh = nullptr;
        DEC_HIP(d, hipHostMalloc((void**)&d->gh, sizeof(int32_t) * need, hipHostMallocDefault));
        d->gh_cap = need;
    }
    if (!d->g_done) DEC_HIP(d, hipEventCreateWithFlags(&d->g_done, hipEventDisableTiming));
    int32_t *h_ids = d->gh, *h_offs = d->gh + rows, *h_n = d->gh + 2 * rows, *h_sz = d->gh + 2 * rows + B;
    if (sizes) {
        std::memcpy(h_sz, sizes, sizeof(int32_t) * B);
        DEC_HIP(d, hipMemcpyAsync(d->g_sizes, h_sz, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    }
    launch_greedy(probs, sizes ? d->g_sizes : nullptr, B, To, (int)d->labels.size(), d->blank, d->g_raw, d->g_ids, d->g_offs, d->g_nout, s);
    DEC_HIP(d, hipMemcpyAsync(h_ids, d->g_ids, sizeof(int32_t) * rows, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(h_offs, d->g_offs, sizeof(int32_t) * rows, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(h_n, d->g_nout, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipEventRecord(d->g_done, s));
    DEC_HIP(d, hipGetLastError());
    d->greedy_pending = true; d->gp_B = B; d->gp_T = To;
    return DSMI_OK;
}

extern "C" int dsmi_greedy_collect(dsmi_decoder* d, int32_t* ids, int32_t* offsets, int32_t* n_out) {
    if (!d) return DSMI_ERR_INVALID;
    if (!d->greedy_pending) { d->err = "no greedy decode to collect"; return DSMI_ERR_INVALID; }
    if (!ids || !offsets || !n_out) { d->err = "bad greedy arguments"; return DSMI_ERR_INVALID; }
    DEC_HIP(d, hipSetDevice(d->device));
    // the pinned image may be reused by the next enqueue only once its copies have landed: `pending` is cleared behind the wait,
    // and a wait that fails falls back to the device's (the copies are then over, whatever state they are in)
    if (hipEventSynchronize(d->g_done) != hipSuccess) {
        (void)hipDeviceSynchronize();
        d->greedy_pending = false;
        d->err = "hipEventSynchronize failed while collecting the greedy decode";
        return DSMI_ERR_HIP;
    }
    d->greedy_pending = false;
    const size_t rows = (size_t)d->gp_B * d->gp_T;
    std::memcpy(ids, d->gh, sizeof(int32_t) * rows);
    std::memcpy(offsets, d->gh + rows, sizeof(int32_t) * rows);
    std::memcpy(n_out, d->gh + 2 * rows, sizeof(int32_t) * d->gp_B);
    return DSMI_OK;
}

// Launches the beam search of a batch, asynchronous on `stream`; dsmi_beam_collect copies the results.
extern "C" int dsmi_beam_enqueue(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, int beam, int cutoff_top_n,
                                 double cutoff_prob, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    const int C = (int)d->labels.size();
    if (!probs || B < 1 || To < 1 || beam < 1) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    if (d->beam_pending) { d->err = "the previous beam search has not been collected"; return DSMI_ERR_INVALID; }
    // 1024 threads per utterance: with 512 the per-frame phases are cheaper (no register spills) but a beam of 128 leaves too few
    // pair threads (24 us per frame against 13; at beam 64 the two are equal: profiles/r03_beam_anatomy.txt)
    constexpr int BT = 1024;
    const size_t lds = carve(beam, C, BT).bytes;
    // thread roles (beam_kernel.inc): ceil(beam / 64) waves for the entries, as many again for the (entry, space) pairs when a
    // scorer is set, the other threads `per` (entry, label) pairs each
    const int EW = (beam + 63) / 64, RW = d->has_lm ? 2 * EW : EW;
    const size_t per = BT > 64 * RW ? ((size_t)beam * C + (BT - 64 * RW) - 1) / (BT - 64 * RW) : ~(size_t)0;
    if (lds > 160 * 1024 - 256 || per > 24 || C > MAXC) { d->err = "beam_width * (n_labels + 1) exceeds the on-chip candidate buffer"; return DSMI_ERR_CAPACITY; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    // ---- workspace carve
    const int ncap = 2 + To * beam;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_nodes = off; off += al((size_t)B * ncap * sizeof(NodeRec));
    size_t o_sizes = off; off += al((size_t)B * 4);
    // outputs, contiguous, in the order of the pinned image: tokens, steps, lens, counts, scores
    const size_t o_out = off;
    size_t o_tok = off; off += al((size_t)B * beam * To * 4);
    size_t o_step = off; off += al((size_t)B * beam * To * 4);
    size_t o_len = off; off += al((size_t)B * beam * 4);
    size_t o_n = off; off += al((size_t)B * 4);
    size_t o_score = off; off += al((size_t)B * beam * 8);
    size_t o_dbg = off; off += al((size_t)B * 16);
    size_t o_stamps = off; off += al(64 * 8 * 8);
    const size_t out_bytes = off - o_out;
    if (off > d->ws_bytes) {
        DEC_HIP(d, hipDeviceSynchronize());
        if (d->ws) (void)hipFree(d->ws);
        d->ws = nullptr; d->ws_bytes = 0;
        DEC_HIP(d, hipMalloc((void**)&d->ws, off));
        d->ws_bytes = off;
    }
    if (out_bytes > d->pin_bytes) {
        DEC_HIP(d, hipDeviceSynchronize());
        if (d->pin) (void)hipHostFree(d->pin);
        d->pin = nullptr; d->pin_bytes = 0;
        DEC_HIP(d, hipHostMalloc((void**)&d->pin, out_bytes + out_bytes / 4, hipHostMallocDefault));
        d->pin_bytes = out_bytes + out_bytes / 4;
    }
    if (!d->beam_done) DEC_HIP(d, hipEventCreateWithFlags(&d->beam_done, hipEventDisableTiming));
    unsigned char* w = d->ws;
    BeamArgs a{};
    a.probs = probs; a.T = To; a.C = C; a.blank = d->blank; a.space = d->space; a.beam = beam;
    a.cutoff_top_n = cutoff_top_n; a.cutoff_prob = (float)cutoff_prob;
    a.has_lm = d->has_lm ? 1 : 0; a.order = d->has_lm ? d->lm.order : 1; a.alpha = d->alpha; a.beta = d->beta;
    a.lm = d->lm.view(); a.lm.tab = d->d_tab; a.lm.klm.base = d->d_klm; a.trie_next = d->d_next; a.trie_word = d->d_word; a.unk = d->lm.unk; a.bos = d->lm.bos;
    a.ncap = ncap;
    a.nodes = (NodeRec*)(w + o_nodes); a.dbg = (int32_t*)(w + o_dbg); a.stamps = (unsigned long long*)(w + o_stamps);
    a.out_tok = (int32_t*)(w + o_tok); a.out_step = (int32_t*)(w + o_step); a.out_len = (int32_t*)(w + o_len); a.out_n = (int32_t*)(w + o_n);
    a.out_score = (double*)(w + o_score);
    a.sizes = nullptr;
    if (sizes) {
        // through pinned memory: an asynchronous copy from pageable memory makes the HOST wait for the stream to get there
        // (i.e. for the forward this search was queued behind), and the caller's array need not outlive this call
        if ((size_t)B > d->pin_sz_cap) {
            if (d->pin_sz) { DEC_HIP(d, hipDeviceSynchronize()); (void)hipHostFree(d->pin_sz); d->pin_sz = nullptr; d->pin_sz_cap = 0; }
            DEC_HIP(d, hipHostMalloc((void**)&d->pin_sz, sizeof(int32_t) * std::max(B, 256), hipHostMallocDefault));
            d->pin_sz_cap = (size_t)std::max(B, 256);
        }
        std::memcpy(d->pin_sz, sizes, sizeof(int32_t) * B);
        DEC_HIP(d, hipMemcpyAsync(w + o_sizes, d->pin_sz, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
        a.sizes = (const int32_t*)(w + o_sizes);
    }
    DEC_HIP(d, hipMemsetAsync(w + o_len, 0, (size_t)B * beam * 4, s));
    DEC_HIP(d, hipMemsetAsync(w + o_dbg, 0, (size_t)B * 16, s));
    DEC_HIP(d, hipMemsetAsync(w + o_stamps, 0, 64 * 8 * 8, s));
    // candidates per thread: a compile-time bound, so that they live in registers
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(B), dim3(BT), lds, s, a);
        return hipSuccess;
    };
    // the reference's alphabet (33 labels) at its default beam (64) and at BASELINE's 128: everything a compile-time constant
    if (C == 33 && beam == 64) DEC_HIP(d, launch(beam_kernel<1024, 3, 64, 33>));
    else if (C == 33 && beam == 128) DEC_HIP(d, launch(beam_kernel<1024, 6, 128, 33>));
    else if (per <= 3) DEC_HIP(d, launch(beam_kernel<1024, 3>));
    else if (per <= 6) DEC_HIP(d, launch(beam_kernel<1024, 6>));
    else if (per <= 12) DEC_HIP(d, launch(beam_kernel<1024, 12>));
    else DEC_HIP(d, launch(beam_kernel<1024, 24>));      // beams beyond ~170 with a scorer: correct, not tuned (registers spill)
    DEC_HIP(d, hipGetLastError());
    // Only the kernel is queued here.  A device-to-host copy queued behind it would sit in a DMA queue until the search is
    // over, and with it whatever upload another stream has been given the same engine for -- the next batch's samples: its
    // forward then starts only when this search has ended (measured: two batches in flight ran one after the other).
    DEC_HIP(d, hipEventRecord(d->beam_done, s));
    d->beam_pending = true; d->pb_B = B; d->pb_To = To; d->pb_beam = beam; d->pb_out = o_out; d->pb_out_bytes = out_bytes; d->pb_stream = s;
    return DSMI_OK;
}

// Waits for the enqueued beam search and hands its results over (layouts of dsmi_beam).
extern "C" int dsmi_beam_collect(dsmi_decoder* d, int32_t* tokens, int32_t* tsteps, int32_t* lens, float* scores) {
    if (!d) return DSMI_ERR_INVALID;
    if (!d->beam_pending) { d->err = "no beam search enqueued"; return DSMI_ERR_INVALID; }
    if (!tokens || !tsteps || !lens || !scores) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    d->beam_pending = false;
    DEC_HIP(d, hipSetDevice(d->device));
    DEC_HIP(d, hipEventSynchronize(d->beam_done));
    const int B = d->pb_B, To = d->pb_To, beam = d->pb_beam;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t tok_bytes = al((size_t)B * beam * To * 4);
    // lengths, counts and scores first (they are behind the two token arrays in the image) ...
    const unsigned char* dev = d->ws + d->pb_out;
    // (on a stream of the handle's own: a blocking copy would go to the null stream and wait there for whatever forward the
    // caller has queued meanwhile, and the search's stream may already hold the next batch's search of another handle)
    if (!d->copy_stream) d->copy_stream = collect_stream(d->device);
    if (!d->copy_stream) { d->err = "no stream for the collect's copies"; return DSMI_ERR_HIP; }
    DEC_HIP(d, hipMemcpyAsync(d->pin + 2 * tok_bytes, dev + 2 * tok_bytes, d->pb_out_bytes - 2 * tok_bytes, hipMemcpyDeviceToHost, d->copy_stream));
    DEC_HIP(d, hipStreamSynchronize(d->copy_stream));
    const unsigned char* q0 = d->pin;
    const int32_t* p_tok = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* p_step = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* h_len = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * beam * 4);
    const int32_t* h_n = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * 4);
    const double* h_score = reinterpret_cast<const double*>(q0); q0 += al((size_t)B * beam * 8);
    const int32_t* h_dbg = reinterpret_cast<const int32_t*>(q0);
    for (int k = 0; k < 4; ++k) { d->stats[k] = 0; for (int b = 0; b < B; ++b) d->stats[k] += h_dbg[4 * b + k]; }
    std::memcpy(d->stamps, reinterpret_cast<const unsigned char*>(h_dbg) + al((size_t)B * 16), sizeof(d->stamps));
    // ... then, of the token arrays, only the columns that hold tokens: transcripts are a fraction of T_out long
    int maxlen = 0;
    for (size_t i = 0; i < (size_t)B * beam; ++i) maxlen = std::max(maxlen, (int)h_len[i]);
    if (maxlen > 0) {
        DEC_HIP(d, hipMemcpy2DAsync(d->pin, (size_t)To * 4, dev, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost, d->copy_stream));
        DEC_HIP(d, hipMemcpy2DAsync(d->pin + tok_bytes, (size_t)To * 4, dev + tok_bytes, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost, d->copy_stream));
        DEC_HIP(d, hipStreamSynchronize(d->copy_stream));
    }
    for (size_t r = 0; r < (size_t)B * beam && maxlen > 0; ++r) {
        std::memcpy(tokens + r * To, p_tok + r * To, (size_t)maxlen * 4);
        std::memcpy(tsteps + r * To, p_step + r * To, (size_t)maxlen * 4);
    }
    // ---- ctcdecode "approx_ctc": strip the word bonus and the LM weight; score = -approx
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < beam; ++p) {
            const size_t q = (size_t)b * beam + p;
            if (p >= h_n[b]) { lens[q] = 0; scores[q] = 0.f; continue; }
            lens[q] = h_len[q];
            double approx = h_score[q];
            if (d->has_lm) {
                std::vector<int32_t> words;
                std::string cur;
                auto flush = [&]() {
                    if (cur.empty()) return;
                    auto it = d->lm.word2id.find(cur);
                    words.push_back(it == d->lm.word2id.end() ? -1 : it->second);
                    cur.clear();
                };
                const int32_t* tk = p_tok + q * To;
                for (int k = 0; k < h_len[q]; ++k) { if (tk[k] == d->space) flush(); else cur += d->labels[tk[k]]; }
                flush();
                approx -= (double)h_len[q] * d->beta;
                approx -= d->lm.sent_ln(words) * d->alpha;
            }
            scores[q] = (float)-approx;
        }
    return DSMI_OK;
}

extern "C" int dsmi_decoder_beam_stats(const dsmi_decoder* d, int32_t* counts4) {
    if (!d || !counts4) return DSMI_ERR_INVALID;
    for (int k = 0; k < 4; ++k) counts4[k] = d->stats[k];
    return DSMI_OK;
}

extern "C" int dsmi_debug_beam_stamps(const dsmi_decoder* d, uint64_t* stamps_host, int64_t n_words) {
    if (!d || !stamps_host || n_words < 64 * 8) return DSMI_ERR_INVALID;
    std::memcpy(stamps_host, d->stamps, sizeof(d->stamps));
    return DSMI_OK;
}

extern "C" int dsmi_beam(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, int beam, int cutoff_top_n,
                         double cutoff_prob, int32_t* tokens, int32_t* tsteps, int32_t* lens, float* scores, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    if (!tokens || !tsteps || !lens || !scores) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    const int rc = dsmi_beam_enqueue(d, probs, sizes, B, To, beam, cutoff_top_n, cutoff_prob, stream);
    if (rc) return rc;
    return dsmi_beam_collect(d, tokens, tsteps, lens, scores);
}


// ---- host-only view of a language model file (no GPU needed): lets callers and the CPU tests inspect what
// dsmi_decoder_set_lm would load.  See include/dsmi.h.
struct dsmi_lm { dsmi::HostLM lm; std::string err; };
static thread_local std::string g_lm_error;

extern "C" int dsmi_lm_open(const char* path, dsmi_lm** out) {
    if (!path || !out) { g_lm_error = "null argument"; return DSMI_ERR_INVALID; }
    dsmi_lm* h = new dsmi_lm();
    const std::string msg = h->lm.load(path, std::vector<std::string>());
    if (!msg.empty()) { g_lm_error = msg; delete h; return DSMI_ERR_IO; }
    *out = h;
    return DSMI_OK;
}
extern "C" void dsmi_lm_close(dsmi_lm* h) { delete h; }
extern "C" const char* dsmi_lm_last_error(const dsmi_lm* h) { return h ? h->err.c_str() : g_lm_error.c_str(); }
extern "C" int dsmi_lm_info(const dsmi_lm* h, int* order, int64_t* vocab_size, int* kind) {
    if (!h) return DSMI_ERR_INVALID;
    if (order) *order = h->lm.order;
    if (vocab_size) *vocab_size = (int64_t)h->lm.vocab.size();
    if (kind) *kind = h->lm.kind;
    return DSMI_OK;
}
extern "C" int dsmi_lm_word_index(const dsmi_lm* h, const char* word_utf8) {
    if (!h || !word_utf8) return DSMI_ERR_INVALID;
    auto it = h->lm.word2id.find(word_utf8);
    return it == h->lm.word2id.end() ? -1 : it->second;
}
extern "C" int dsmi_lm_lookup(const dsmi_lm* h, const int32_t* ids, int n, float* log10_prob, float* log10_backoff) {
    if (!h || !ids || n < 1 || n > h->lm.order) return DSMI_ERR_INVALID;
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= (int32_t)h->lm.vocab.size()) return DSMI_ERR_INVALID;
    float lp = 0.f, bo = 0.f;
    const bool found = dsmi::lm_lookup(h->lm.view(), ids, n, &lp, &bo);
    if (log10_prob) *log10_prob = lp;
    if (log10_backoff) *log10_backoff = bo;
    return found ? 1 : 0;
}
extern "C" double dsmi_lm_cond_log10(const dsmi_lm* h, const int32_t* ids, int n) {
    if (!h || !ids || n < 1 || n > h->lm.order) return 0.0 / 0.0;
    return (double)dsmi::lm_cond_log10(h->lm.view(), ids, n - 1, ids[n - 1], h->lm.unk);
}
