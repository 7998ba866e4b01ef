// The fused recognise pipeline for hosts without the Python layer: what Recognizer.recognize ->
// DanSpeechRecognizer.transcribe (reference danspeech/Recognizer.py:158-189, DanSpeechRecognizer.py:191-231) does for one
// recording, over a batch: stage + upload the clips, spectrograms, network, decoder, label strings, caller's order.
// Written against the public C ABI only (dsmi_features, dsmi_forward, dsmi_forward_status, dsmi_greedy, dsmi_beam) -- the
// same sequence danspeech_amd/DanSpeechRecognizer.py issues -- plus the buffers between the stages, which a session owns.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <immintrin.h>
#include <cstring>
#include <numeric>
#include <thread>

struct dsmi_session {
    dsmi_frontend* f = nullptr; dsmi_model* m = nullptr; dsmi_decoder* d = nullptr;
    int device = 0, n_freq = 0, hop = 0, n_labels = 0;
    hipStream_t stream = nullptr;
    std::string err;
    unsigned char* pin = nullptr; size_t pin_cap = 0;        // pinned staging of the batch's samples, longest clip first
    unsigned char* pcm = nullptr; size_t pcm_cap = 0;        // the same on the device
    float* feat = nullptr; size_t feat_cap = 0;              // [B][1][n_freq][T]
    float* probs = nullptr; size_t probs_cap = 0;            // [B][T_out][n_labels]
    // the batch between dsmi_recognize_enqueue and dsmi_recognize_collect
    bool pending = false;
    int B = 0, T_out = 0;
    std::vector<int> order;                                   // order[pos] = caller's index of the clip run at position pos
    std::vector<int32_t> out_lens;
};

static thread_local std::string g_session_error;

namespace {

int sfail(dsmi_session* s, int code, const std::string& msg) { s->err = msg; return code; }

// bytes per sample (per frame for stereo) of a DSMI_PCM_* code; 0 = unknown
int sample_bytes(int dtype) {
    static const int w[6] = {2, 4, 8, 1, 3, 4};               // I16 F32 F64 U8 I24 I32
    const int base = dtype & 15;
    if (dtype < 0 || base > DSMI_PCM_I32 || (dtype & ~(15 | DSMI_PCM_STEREO))) return 0;
    return w[base] * ((dtype & DSMI_PCM_STEREO) ? 2 : 1);
}

template <class T>
bool grow(dsmi_session* s, T** p, size_t* cap, size_t need) {
    if (need <= *cap) return true;
    if (*p) { (void)hipStreamSynchronize(s->stream); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = need + need / 4;
    if (hipMalloc((void**)p, want) != hipSuccess) return false;
    *cap = want;
    return true;
}

}  // namespace

extern "C" const char* dsmi_session_last_error(const dsmi_session* s) { return s ? s->err.c_str() : g_session_error.c_str(); }

extern "C" int dsmi_session_create(dsmi_frontend* f, dsmi_model* m, dsmi_decoder* d, dsmi_session** out) {
    if (!f || !m || !d || !out) { g_session_error = "bad session arguments"; return DSMI_ERR_INVALID; }
    int fdev = 0, mdev = 0, ddev = 0, n_freq = 0, hop = 0, n_labels = 0;
    dsmi_model_desc desc;
    if (dsmi_frontend_info(f, &n_freq, &hop, &fdev) || dsmi_model_info(m, &desc, &mdev) || dsmi_decoder_info(d, &n_labels, nullptr, &ddev)) {
        g_session_error = "bad handle"; return DSMI_ERR_INVALID;
    }
    if (fdev != mdev || ddev != mdev) { g_session_error = "frontend, model and decoder live on different devices"; return DSMI_ERR_INVALID; }
    if (n_labels != desc.n_labels) { g_session_error = "the decoder's label count is not the model's"; return DSMI_ERR_INVALID; }
    if (hipSetDevice(mdev) != hipSuccess) { g_session_error = "hipSetDevice failed"; return DSMI_ERR_HIP; }
    dsmi_session* s = new dsmi_session;
    s->f = f; s->m = m; s->d = d; s->device = mdev; s->n_freq = n_freq; s->hop = hop; s->n_labels = n_labels;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; g_session_error = "hipStreamCreate failed"; return DSMI_ERR_HIP; }
    *out = s;
    return DSMI_OK;
}

extern "C" void dsmi_session_destroy(dsmi_session* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->pin) (void)hipHostFree(s->pin);
    if (s->pcm) (void)hipFree(s->pcm);
    if (s->feat) (void)hipFree(s->feat);
    if (s->probs) (void)hipFree(s->probs);
    delete s;
}

// spectrograms + network for B clips back to back at pcm_dev, longest first; asynchronous on the session's stream
static int run_from_device(dsmi_session* s, const void* pcm_dev, const int64_t* n, int pcm_dtype, int B) {

    std::vector<int32_t> frames((size_t)B);
    const int T = 1 + (int)(n[0] / s->hop);
    if (!grow(s, &s->feat, &s->feat_cap, sizeof(float) * (size_t)B * s->n_freq * T)) return sfail(s, DSMI_ERR_NOMEM, "hipMalloc failed");
    int rc = dsmi_features(s->f, pcm_dev, pcm_dtype, n, B, s->feat, T, frames.data(), s->stream);
    if (rc) return sfail(s, rc, std::string("dsmi_features: ") + dsmi_frontend_last_error(s->f));
    int32_t t_in = T, t_out = 0;
    if ((rc = dsmi_seq_lens(s->m, &t_in, 1, &t_out))) return sfail(s, rc, std::string("dsmi_seq_lens: ") + dsmi_last_error(s->m));
    if (!grow(s, &s->probs, &s->probs_cap, sizeof(float) * (size_t)B * t_out * s->n_labels)) return sfail(s, DSMI_ERR_NOMEM, "hipMalloc failed");
    s->out_lens.assign((size_t)B, 0);
    rc = dsmi_forward(s->m, s->feat, frames.data(), B, T, s->probs, s->out_lens.data(), s->stream);
    if (rc) return sfail(s, rc, std::string("dsmi_forward: ") + dsmi_last_error(s->m));
    s->B = B; s->T_out = t_out; s->pending = true;
    return DSMI_OK;
}

// See include/dsmi.h: float64 samples that are integers in int16's range, as int16.  Eight samples per step on AVX2 (the host code is
// built for the x86-64 baseline; the wide path is chosen at run time): truncate to int32, widen back, compare with the sample (a
// fraction, a NaN or anything beyond int32 differs), compare with int16's range, saturating pack.
__attribute__((target("avx2"))) static int pack_i16_avx2(const double* src, int64_t n, int16_t* dst) {
    const __m256d lo = _mm256_set1_pd(-32768.0), hi = _mm256_set1_pd(32767.0);
    int64_t i = 0;
    for (; i + 8 <= n;) {
        const int64_t stop = std::min<int64_t>(n - 7, i + 4096);
        __m256d bad = _mm256_setzero_pd();
        for (; i < stop; i += 8) {
            const __m256d a = _mm256_loadu_pd(src + i), b = _mm256_loadu_pd(src + i + 4);
            const __m128i qa = _mm256_cvttpd_epi32(a), qb = _mm256_cvttpd_epi32(b);
            bad = _mm256_or_pd(bad, _mm256_or_pd(_mm256_cmp_pd(_mm256_cvtepi32_pd(qa), a, _CMP_NEQ_UQ), _mm256_cmp_pd(_mm256_cvtepi32_pd(qb), b, _CMP_NEQ_UQ)));
            bad = _mm256_or_pd(bad, _mm256_or_pd(_mm256_or_pd(_mm256_cmp_pd(a, lo, _CMP_LT_OQ), _mm256_cmp_pd(a, hi, _CMP_GT_OQ)),
                                                 _mm256_or_pd(_mm256_cmp_pd(b, lo, _CMP_LT_OQ), _mm256_cmp_pd(b, hi, _CMP_GT_OQ))));
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm_packs_epi32(qa, qb));
        }
        if (_mm256_movemask_pd(bad)) return 0;
    }
    for (; i < n; ++i) {
        const double v = src[i];
        const double c = (v >= -32768.0 && v <= 32767.0) ? v : 0.0;
        const int16_t q = (int16_t)(int32_t)c;
        if ((double)q != v) return 0;
        dst[i] = q;
    }
    return 1;
}

extern "C" int dsmi_pack_pcm_i16(const double* src, int64_t n, int16_t* dst) {
    if (!src || !dst || n < 0) return 0;
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return pack_i16_avx2(src, n, dst);
    for (int64_t i0 = 0; i0 < n; i0 += 4096) {
        const int64_t m = std::min<int64_t>(4096, n - i0);
        int bad = 0;
        for (int64_t i = 0; i < m; ++i) {
            const double v = src[i0 + i];
            const double c = (v >= -32768.0 && v <= 32767.0) ? v : 0.0;       // (a NaN fails both comparisons and then differs from c)
            const int16_t q = (int16_t)(int32_t)c;
            bad |= (int)((double)q != v);
            dst[i0 + i] = q;
        }
        if (bad) return 0;
    }
    return 1;
}

// See include/dsmi.h: the upload as a kernel.  64 workgroups (a quarter of the CUs, a wave's 1-KiB reads in flight on each) read the
// pinned buffer 16 bytes per lane and write device memory; a tail of fewer than 16 bytes goes 2 bytes at a time.
using up_u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__global__ __launch_bounds__(256) void upload_kernel(up_u32x4* __restrict__ dst, const up_u32x4* __restrict__ src, size_t n16, size_t tail2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && threadIdx.x < tail2)
        reinterpret_cast<uint16_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint16_t*>(src + n16)[threadIdx.x];
}

extern "C" int dsmi_upload(int device, void* dst_dev, const void* src_pinned, int64_t bytes, void* stream) {
    if (!dst_dev || !src_pinned || bytes < 0 || (bytes & 1) || ((uintptr_t)dst_dev & 15) || ((uintptr_t)src_pinned & 15)) return DSMI_ERR_INVALID;
    if (bytes == 0) return DSMI_OK;
    if (hipSetDevice(device) != hipSuccess) return DSMI_ERR_HIP;
    const size_t n16 = (size_t)bytes / 16, tail2 = ((size_t)bytes % 16) / 2;
    const int grid = (int)std::min<size_t>(64, std::max<size_t>(1, (n16 + 255) / 256));
    hipLaunchKernelGGL(upload_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (up_u32x4*)dst_dev, (const up_u32x4*)src_pinned, n16, tail2);
    return hipGetLastError() == hipSuccess ? DSMI_OK : DSMI_ERR_HIP;
}

// Stage, upload, spectrograms, network: everything up to the probabilities, asynchronous on the session's stream.
extern "C" int dsmi_recognize_enqueue(dsmi_session* s, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int B) {
    if (!s) return DSMI_ERR_INVALID;
    const int sb = sample_bytes(pcm_dtype);
    if (!clips_host || !n_samples_host || B < 1 || !sb) return sfail(s, DSMI_ERR_INVALID, "bad recognise arguments");
    if (s->pending) return sfail(s, DSMI_ERR_INVALID, "the previous batch has not been collected");
    if (hipSetDevice(s->device) != hipSuccess) return sfail(s, DSMI_ERR_HIP, "hipSetDevice failed");
    size_t total = 0;
    for (int b = 0; b < B; ++b) {
        if (!clips_host[b] || n_samples_host[b] < 1) return sfail(s, DSMI_ERR_INVALID, "empty clip");
        total += (size_t)n_samples_host[b];
    }
    // longest first (stable), the order pack_padded_sequence wants (reference model.py:117)
    s->order.resize((size_t)B);
    std::iota(s->order.begin(), s->order.end(), 0);
    std::stable_sort(s->order.begin(), s->order.end(), [&](int a, int b) { return n_samples_host[a] > n_samples_host[b]; });
    std::vector<int64_t> n((size_t)B);
    std::vector<size_t> off((size_t)B + 1, 0);
    for (int pos = 0; pos < B; ++pos) {
        n[(size_t)pos] = n_samples_host[s->order[(size_t)pos]];
        off[(size_t)pos + 1] = off[(size_t)pos] + (size_t)n[(size_t)pos] * sb;
    }
    const size_t bytes = total * sb;
    if (bytes > s->pin_cap) {
        (void)hipStreamSynchronize(s->stream);
        if (s->pin) (void)hipHostFree(s->pin);
        s->pin = nullptr; s->pin_cap = 0;
        if (hipHostMalloc((void**)&s->pin, bytes + bytes / 4, hipHostMallocDefault) != hipSuccess) return sfail(s, DSMI_ERR_NOMEM, "hipHostMalloc failed");
        s->pin_cap = bytes + bytes / 4;
    } else {
        (void)hipStreamSynchronize(s->stream);       // the previous batch's upload has left the staging buffer
    }
    // float64 clips (what load_audio returns) travel as int16 where that is exact (dsmi_pack_pcm_i16): a quarter of the upload
    std::atomic<int> exact{pcm_dtype == DSMI_PCM_F64 ? 1 : 0};
    auto copy = [&](int lo, int hi, bool pack) {
        for (int pos = lo; pos < hi; ++pos) {
            const void* src = clips_host[s->order[(size_t)pos]];
            if (!pack) std::memcpy(s->pin + off[(size_t)pos], src, off[(size_t)pos + 1] - off[(size_t)pos]);
            else if (exact.load(std::memory_order_relaxed) &&
                     !dsmi_pack_pcm_i16((const double*)src, n[(size_t)pos], (int16_t*)(s->pin + off[(size_t)pos] / 4))) exact.store(0);
        }
    };
    auto fill = [&](bool pack) {
        if (bytes >= ((size_t)8 << 20) && B >= 8) {       // tens of megabytes: four host threads fill the pinned buffer
            std::thread th[3];
            for (int q = 0; q < 3; ++q) th[q] = std::thread(copy, B * (q + 1) / 4, B * (q + 2) / 4, pack);
            copy(0, B / 4, pack);
            for (auto& t : th) t.join();
        } else {
            copy(0, B, pack);
        }
    };
    if (exact.load()) fill(true);
    const bool as_i16 = exact.load() != 0;
    if (!as_i16) fill(false);
    const size_t up = as_i16 ? bytes / 4 : bytes;
    if (!grow(s, &s->pcm, &s->pcm_cap, up)) return sfail(s, DSMI_ERR_NOMEM, "hipMalloc failed");
    // (by kernel, as the Python pipeline does: see dsmi_upload; an odd byte count -- 8-bit samples -- goes through the copy engine)
    if ((up & 1) ? hipMemcpyAsync(s->pcm, s->pin, up, hipMemcpyHostToDevice, s->stream) != hipSuccess
                 : dsmi_upload(s->device, s->pcm, s->pin, (int64_t)up, s->stream) != DSMI_OK) return sfail(s, DSMI_ERR_HIP, "upload failed");
    return run_from_device(s, s->pcm, n.data(), as_i16 ? DSMI_PCM_I16 : pcm_dtype, B);
}

// The same for clips that already sit back to back in device memory, longest first -- a shard dsmi_comm_scatter delivered:
// no host staging at all (danspeech_amd DanSpeechRecognizer.transcribe_device).  pcm_dev must stay valid until the collect.
extern "C" int dsmi_recognize_enqueue_device(dsmi_session* s, const void* pcm_dev, const int64_t* n_samples_host, int pcm_dtype, int B) {
    if (!s) return DSMI_ERR_INVALID;
    if (!pcm_dev || !n_samples_host || B < 1 || !sample_bytes(pcm_dtype)) return sfail(s, DSMI_ERR_INVALID, "bad recognise arguments");
    if (s->pending) return sfail(s, DSMI_ERR_INVALID, "the previous batch has not been collected");
    if (hipSetDevice(s->device) != hipSuccess) return sfail(s, DSMI_ERR_HIP, "hipSetDevice failed");
    for (int b = 0; b < B; ++b) {
        if (n_samples_host[b] < 1) return sfail(s, DSMI_ERR_INVALID, "empty clip");
        if (b && n_samples_host[b] > n_samples_host[b - 1]) return sfail(s, DSMI_ERR_UNSORTED, "device-resident clips must come longest first");
    }
    s->order.resize((size_t)B);
    std::iota(s->order.begin(), s->order.end(), 0);
    return run_from_device(s, pcm_dev, n_samples_host, pcm_dtype, B);
}

// Wait for the forward (a timed-out batch has been recomputed by then), decode, build the strings.
extern "C" int dsmi_recognize_collect(dsmi_session* s, int beam_width, int cutoff_top_n, double cutoff_prob,
                                      char* text_utf8, int text_stride, int32_t* text_bytes_host, float* scores_host) {
    if (!s) return DSMI_ERR_INVALID;
    if (!s->pending) return sfail(s, DSMI_ERR_INVALID, "no batch enqueued");
    if (!text_utf8 || text_stride < 1 || beam_width < 0) return sfail(s, DSMI_ERR_INVALID, "bad collect arguments");
    s->pending = false;
    if (hipSetDevice(s->device) != hipSuccess) return sfail(s, DSMI_ERR_HIP, "hipSetDevice failed");
    const int status = dsmi_forward_status(s->m);
    if (status < 0) return sfail(s, status, std::string("dsmi_forward_status: ") + dsmi_last_error(s->m));
    int rc = 0;
    const int B = s->B, To = s->T_out;
    std::vector<int32_t> ids, lens;
    std::vector<float> scores;
    size_t row = 0;                                           // ids of clip position pos start at pos * row
    if (beam_width == 0) {
        std::vector<int32_t> offs((size_t)B * To);
        ids.resize((size_t)B * To); lens.resize((size_t)B);
        rc = dsmi_greedy(s->d, s->probs, s->out_lens.data(), B, To, ids.data(), offs.data(), lens.data(), s->stream);
        if (rc) return sfail(s, rc, std::string("dsmi_greedy: ") + dsmi_decoder_last_error(s->d));
        row = (size_t)To;
    } else {
        std::vector<int32_t> ts((size_t)B * beam_width * To), bl((size_t)B * beam_width);
        ids.resize((size_t)B * beam_width * To); scores.resize((size_t)B * beam_width);
        rc = dsmi_beam(s->d, s->probs, s->out_lens.data(), B, To, beam_width, cutoff_top_n, cutoff_prob, ids.data(), ts.data(), bl.data(),
                       scores.data(), s->stream);
        if (rc) return sfail(s, rc, std::string("dsmi_beam: ") + dsmi_decoder_last_error(s->d));
        row = (size_t)beam_width * To;                        // the best beam is the first
        lens.resize((size_t)B);
        for (int pos = 0; pos < B; ++pos) lens[(size_t)pos] = bl[(size_t)pos * beam_width];
    }
    for (int pos = 0; pos < B; ++pos) {
        const int b = s->order[(size_t)pos];
        char* dst = text_utf8 + (size_t)b * text_stride;
        int used = 0, full = 0;
        bool room = true;
        for (int i = 0; i < lens[(size_t)pos]; ++i) {
            const char* lab = dsmi_decoder_label(s->d, ids[(size_t)pos * row + i]);
            const int len = lab ? (int)std::strlen(lab) : 0;
            full += len;
            if (room && used + len < text_stride) { std::memcpy(dst + used, lab, (size_t)len); used += len; }
            else room = false;                                // truncated on a label boundary; text_bytes_host tells
        }
        dst[used] = 0;
        if (text_bytes_host) text_bytes_host[b] = full;
        if (scores_host) scores_host[b] = beam_width ? scores[(size_t)pos * beam_width] : 0.f;
    }
    return status;                                            // DSMI_OK or DSMI_RECOMPUTED
}

extern "C" int dsmi_recognize_batch(dsmi_session* s, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int B,
                                    int beam_width, int cutoff_top_n, double cutoff_prob,
                                    char* text_utf8, int text_stride, int32_t* text_bytes_host, float* scores_host) {
    const int rc = dsmi_recognize_enqueue(s, clips_host, n_samples_host, pcm_dtype, B);
    if (rc) return rc;
    return dsmi_recognize_collect(s, beam_width, cutoff_top_n, cutoff_prob, text_utf8, text_stride, text_bytes_host, scores_host);
}
