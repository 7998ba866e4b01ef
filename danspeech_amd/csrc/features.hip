// Batched spectrogram front end on the GPU (gfx950).
//
// Replaces SpectrogramAudioParser.parse_audio (reference danspeech/audio/parsers.py:50-72):
//   D = librosa.stft(y, n_fft, hop, win_length=n_fft, window=scipy window, center=True)
//   spect = log1p(|D|)  ->  (spect - mean) / std       (torch mean / unbiased std over F*T)
// librosa computes the rFFT of the float64 windowed frame in float64 and stores complex64;
// |.| and log1p then run in float32.  The kernel keeps that precision ladder: the DFT sums
// are float64 (direct O(n_fft^2) DFT against a host-computed float64 twiddle table: 161 bins
// x 320 taps, far below an FFT's break-even on this machine and free of any library), the
// real/imaginary parts are rounded to float32, then hypotf and log1pf.  Since round 5 the shipped window length (320) with
// float64 / float32 / int16 samples runs the same float64 sums on the matrix pipe (stft_mfma_kernel below: v_mfma_f64_16x16x4_f64,
// the frame folded twice); the direct kernel serves every other window length and the WAV-frame sample types.
// Mean and unbiased std are accumulated in float64 per clip (two passes, fixed order).
#include "common.h"
#include "host_logic.h"

#include <cmath>
#include <cstring>
#include <algorithm>

using namespace dsmi;

// The handle behind dsmi_frontend* : one SpectrogramAudioParser on one GPU.
struct dsmi_frontend {
    dsmi_frontend_desc desc{};
    int device = 0;
    int n_fft = 0, hop = 0, n_freq = 0;
    std::string err;
    double* tw = nullptr;    // [n_fft][2] cos, sin
    double* win = nullptr;   // [n_fft]
    int64_t* offs = nullptr; // device: per-clip sample offset, n_samples [2][cap], then float64 partial statistics [cap][NSL][2]
    int cap = 0;
    // pinned staging of the per-batch offsets / lengths (an async copy must not read pageable memory that is gone or
    // overwritten when the copy engine gets to it): a ring of slots, each reused only after its copy has completed
    static constexpr int kStage = 4;
    int64_t* stage = nullptr; int stage_cap = 0, stage_next = 0;
    hipEvent_t stage_ev[kStage] = {nullptr, nullptr, nullptr, nullptr}; bool stage_used[kStage] = {false, false, false, false};
};

// host[0..n) -> dev[0..n) on stream s through the frontend's pinned ring
static bool fe_stage_copy(dsmi_frontend* f, int64_t* dev, const int64_t* host, int n, hipStream_t s) {
    if (n > f->stage_cap) {
        if (hipDeviceSynchronize() != hipSuccess) return false;
        if (f->stage) (void)hipHostFree(f->stage);
        f->stage = nullptr;
        const int cap = std::max(n, 256);
        if (hipHostMalloc((void**)&f->stage, sizeof(int64_t) * (size_t)cap * dsmi_frontend::kStage, hipHostMallocDefault) != hipSuccess) return false;
        f->stage_cap = cap;
        for (int i = 0; i < dsmi_frontend::kStage; ++i) {
            if (!f->stage_ev[i] && hipEventCreateWithFlags(&f->stage_ev[i], hipEventDisableTiming) != hipSuccess) return false;
            f->stage_used[i] = false;
        }
    }
    const int slot = f->stage_next++ % dsmi_frontend::kStage;
    if (f->stage_used[slot] && hipEventSynchronize(f->stage_ev[slot]) != hipSuccess) return false;
    int64_t* h = f->stage + (size_t)slot * f->stage_cap;
    std::memcpy(h, host, sizeof(int64_t) * n);
    if (hipMemcpyAsync(dev, h, sizeof(int64_t) * n, hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (hipEventRecord(f->stage_ev[slot], s) != hipSuccess) return false;
    f->stage_used[slot] = true;
    return true;
}

static thread_local std::string g_fe_error;

namespace {

constexpr int FT = 8;   // frames per workgroup
constexpr int PAD_NONE = 2;   // internal third pad mode: no centre padding

// One integer sample of a WAV frame stream (resources.py:551-554 for the 8-bit bias).
__device__ __forceinline__ int64_t ld_int(const void* p, int base, int64_t i) {
    switch (base) {
        case DSMI_PCM_I16: return ((const int16_t*)p)[i];
        case DSMI_PCM_U8: return (int64_t)((const uint8_t*)p)[i] - 128;
        case DSMI_PCM_I32: return ((const int32_t*)p)[i];
        default: {
            const uint8_t* q = (const uint8_t*)p + 3 * i;
            const int32_t v = (int32_t)q[0] | ((int32_t)q[1] << 8) | ((int32_t)q[2] << 16);
            return v >= (1 << 23) ? v - (1 << 24) : v;
        }
    }
}

// Sample i of a clip at its integer scale as float64 (load_audio, resources.py:630-640); two channels fold
// into the saturating sum of audioop.tomono(buf, width, 1, 1) (resources.py:302-303).
__device__ __forceinline__ double ld_sample(const void* p, int dtype, int64_t i) {
    if (dtype == DSMI_PCM_F64) return ((const double*)p)[i];
    if (dtype == DSMI_PCM_F32) return (double)((const float*)p)[i];
    const int base = dtype & 15;
    if (!(dtype & DSMI_PCM_STEREO)) return (double)ld_int(p, base, i);
    const int64_t lim = base == DSMI_PCM_I16 ? (1ll << 15) : (base == DSMI_PCM_I24 ? (1ll << 23) : (1ll << 31));
    const int64_t v = ld_int(p, base, 2 * i) + ld_int(p, base, 2 * i + 1);
    return (double)(v < -lim ? -lim : (v > lim - 1 ? lim - 1 : v));
}

__global__ __launch_bounds__(256) void stft_logmag_kernel(const void* pcm, int dtype, const int64_t* offs, const int64_t* nsamp,
                                                          const double* tw, const double* win, int n_fft, int hop, int n_freq,
                                                          int pad_mode, float* feat, int t_stride) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* s_tw = sm;                 // [n_fft][2]
    double* s_x = sm + 2 * n_fft;      // [n_fft][FT]  (frame fastest: one broadcast b128-friendly row per tap)
    const int b = blockIdx.y, t0 = blockIdx.x * FT, tid = threadIdx.x;
    const int64_t N = nsamp[b], off = offs[b];
    const bool centred = pad_mode != PAD_NONE;            // PAD_NONE: librosa.stft(center=False), the streaming parser
    const int nfr = centred ? 1 + (int)(N / hop) : 1 + (int)((N - n_fft) / hop);
    if (t0 >= nfr) return;
    for (int i = tid; i < 2 * n_fft; i += 256) s_tw[i] = tw[i];
    const int half = centred ? n_fft / 2 : 0;
    for (int i = tid; i < n_fft * FT; i += 256) {
        const int f = i % FT, n = i / FT;
        const int t = t0 + f;
        double v = 0.0;
        if (t < nfr) {
            int64_t s = (int64_t)t * hop + n - half;
            bool ok = true;
            if (s < 0) { if (pad_mode == DSMI_PAD_REFLECT) s = -s; else ok = false; }
            else if (s >= N) { if (pad_mode == DSMI_PAD_REFLECT) s = 2 * (N - 1) - s; else ok = false; }
            if (ok) {
                v = ld_sample(pcm, dtype, off + s) * win[n];
            }
        }
        s_x[i] = v;
    }
    __syncthreads();
    // Real input: x[n] and x[N-n] meet the same cosine and opposite sines, so fold the frame once
    // (e[n] = x[n] + x[N-n], o[n] = x[n] - x[N-n], n = 1 .. N/2-1) and run half as many multiply-adds per bin.
    const bool fold = (n_fft & 1) == 0;
    const int nh = n_fft / 2;
    if (fold) {
        for (int i = tid; i < (nh - 1) * FT; i += 256) {
            const int f = i % FT, n = 1 + i / FT;
            const double a = s_x[n * FT + f], c = s_x[(n_fft - n) * FT + f];
            s_x[n * FT + f] = a + c;
            s_x[(n_fft - n) * FT + f] = a - c;
        }
        __syncthreads();
    }
    for (int k = tid; k < n_freq; k += 256) {
        double re[FT], im[FT];
        int idx = 0;
        if (fold) {
            const double sgn = (k & 1) ? -1.0 : 1.0;          // cos(pi k) of the tap n = N/2
#pragma unroll
            for (int f = 0; f < FT; ++f) { re[f] = fma(sgn, s_x[nh * FT + f], s_x[f]); im[f] = 0.0; }
            idx = k;
            for (int n = 1; n < nh; ++n) {
                const double c = s_tw[2 * idx], s = s_tw[2 * idx + 1];
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    re[f] = fma(s_x[n * FT + f], c, re[f]);
                    im[f] = fma(-s_x[(n_fft - n) * FT + f], s, im[f]);
                }
                idx += k;
                if (idx >= n_fft) idx -= n_fft;
            }
        } else {
#pragma unroll
            for (int f = 0; f < FT; ++f) { re[f] = 0.0; im[f] = 0.0; }
            for (int n = 0; n < n_fft; ++n) {
                const double c = s_tw[2 * idx], s = s_tw[2 * idx + 1];
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const double x = s_x[n * FT + f];
                    re[f] = fma(x, c, re[f]);
                    im[f] = fma(-x, s, im[f]);
                }
                idx += k;
                if (idx >= n_fft) idx -= n_fft;
            }
        }
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int t = t0 + f;
            if (t < nfr) feat[((size_t)b * n_freq + k) * t_stride + t] = log1pf(hypotf((float)re[f], (float)im[f]));
        }
    }
}

// The same transform on the matrix pipe, for the even n_fft the models use (320): v_mfma_f64_16x16x4_f64, float64 in and out -- the
// precision ladder above is kept (a split-fp16 product with float32 accumulation is not: the partial sums of a frame are as large as
// the frame, a quiet bin 50 dB below it is then off by 5e-4 in log1p -- measured, profiles/r05_stft_forms.txt).
// The direct kernel above is bound by LDS bandwidth: every multiply-add reads its sample from LDS (a b128 per two), four SIMDs
// against one LDS port -- a quarter of the float64 rate, and 95 of a workgroup's 256 lanes have no bin.  Here a wave OWNS a tile of 16
// frames: their twice-folded samples (four sequences of 20 or 21 k-steps of four taps: see the kernel) live in its registers as the
// MFMAs' B operands for the five tiles of even and the five of odd bins below 160 (bin 160 is a signed sum on the vector pipe); the A
// operand is the twiddle of (bin, tap), read from a 5-KB table by index (bin * tap mod n_fft, carried by one add and one wrap per k-step).  D[bin][frame]: sixteen lanes of a result register are sixteen consecutive frames of one bin.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int MF = 64;   // frames per workgroup: four waves, one 16-frame tile each

// SKIP (timing experiments, DSMI_DEBUG_STFT_SKIP; results are garbage): 1 no hypotf / log1pf, 2 no MFMAs, 4 no sample / window loads
template <int NFFT, typename T, int SKIP = 0>          // T: double, float or int16_t samples (one channel); the other WAV-frame types take the direct kernel
__global__ __launch_bounds__(256, 2) void stft_mfma_kernel(const T* pcm, const int64_t* offs, const int64_t* nsamp,
                                                           const double* tw, const double* win, int hop, int pad_mode, float* feat, int t_stride) {
    constexpr int NH = NFFT / 2, NQ = NFFT / 4, NFREQ = NH + 1, KQ = NQ / 4;       // NQ taps after the second fold: KQ k-steps of four (+ 1)
    static_assert(NQ % 16 == 0, "taps in k-steps of four; even and odd bins in tiles of sixteen each");
    __shared__ __attribute__((aligned(16))) double s_tw[2 * NFFT];     // [idx][cos, sin]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t N = nsamp[b], off = offs[b];
    const bool centred = pad_mode != PAD_NONE;
    const int nfr = centred ? 1 + (int)(N / hop) : 1 + (int)((N - NFFT) / hop);
    const int t0 = blockIdx.x * MF;
    if (t0 >= nfr) return;
    for (int i = tid; i < 2 * NFFT; i += 256) s_tw[i] = tw[i];
    const int half = centred ? NH : 0;
    const int t = t0 + wid * 16 + (lane & 15), j = lane >> 4;          // B operand: lane holds (tap 4 ks + j, frame lane & 15)
    const bool live = t < nfr;
    auto xw = [&](int n) -> double {                                    // windowed sample n of this lane's frame
        int64_t sidx = (int64_t)t * hop + n - half;
        bool ok = live;
        if (sidx < 0) { sidx = -sidx; ok = ok && pad_mode == DSMI_PAD_REFLECT; }
        else if (sidx >= N) { sidx = 2 * (N - 1) - sidx; ok = ok && pad_mode == DSMI_PAD_REFLECT; }
        if (SKIP & 4) return ok ? (double)(sidx & 255) * (double)(n + 1) : 0.0;
        return ok ? (double)pcm[off + sidx] * win[n] : 0.0;
    };
    // Two folds.  (1) x[n] and x[NFFT - n] meet the same cosine and opposite sines: e[n] = x[n] + x[NFFT - n], o[n] = x[n] - x[NFFT - n]
    // (n = 1 .. NH - 1), e[0] = x[0], e[NH] = x[NH]; re[k] = sum_{n <= NH} e[n] cos(2 pi k n / NFFT), im[k] = sum o[n] sin(...).
    // (2) tap NH - n meets (-1)^k times tap n's cosine and -(-1)^k times its sine, so even and odd bins see different halves:
    //        even k:  re = sum_{n <= NQ} F[n] cos,  F[n] = e[n] + e[NH - n] (F[NQ] = e[NQ]);   im = sum_{n < NQ} P[n] sin,  P[n] = o[n] - o[NH - n]
    //        odd k:   re = sum_{n <  NQ} G[n] cos,  G[n] = e[n] - e[NH - n];                   im = sum_{n <= NQ} Q[n] sin, Q[n] = o[n] + o[NH - n] (Q[NQ] = o[NQ])
    // A quarter of the direct transform's multiply-adds; the twiddle of (bin, tap) is the same table entry (k n mod NFFT) in all four.
    double F[KQ + 1], G[KQ], P[KQ], Q[KQ + 1];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
        const int n = 4 * ks + j;                                       // 0 .. NQ - 1
        const double x0 = xw(n), x1 = n == 0 ? 0.0 : xw(NFFT - n), x2 = xw(NH - n), x3 = n == 0 ? 0.0 : xw(NH + n);
        const double en = x0 + x1, em = n == 0 ? x2 : x2 + x3;          // e[n], e[NH - n]   (e[NH] = x[NH])
        const double on = n == 0 ? 0.0 : x0 - x1, om = n == 0 ? 0.0 : x2 - x3;      // o[n], o[NH - n]   (o[0] = o[NH] = 0)
        F[ks] = en + em; G[ks] = en - em;
        P[ks] = on - om; Q[ks] = on + om;
    }
    {
        const double xa = j == 0 ? xw(NQ) : 0.0, xb = j == 0 ? xw(NFFT - NQ) : 0.0;      // tap NQ: lane group 0 of the last k-step
        F[KQ] = xa + xb; Q[KQ] = xa - xb;
    }
    __syncthreads();
    if (t0 + wid * 16 >= nfr) return;                                   // a clip's last workgroup: tiles past its last frame
    const int to = t0 + wid * 16 + (lane & 15);
    // The last bin (k = NH, even) would be a tile of its own: on the vector pipe instead.  cos(pi n) = (-1)^n and a lane's taps 4 ks + j
    // share the parity of j: the lane's F summed, signed, and added across the four tap groups of a frame; the sine is zero.
    {
        double ny = 0.0;
#pragma unroll
        for (int ks = 0; ks <= KQ; ++ks) ny += F[ks];
        ny = (j & 1) ? -ny : ny;
        ny += __shfl_xor(ny, 16, 64);
        ny += __shfl_xor(ny, 32, 64);
        if (j == 0 && to < nfr) feat[((size_t)b * NFREQ + NH) * t_stride + to] = log1pf(fabsf((float)ny));
    }
    auto tile = [&](int bt, auto odd_tag) {                             // bins 2 (16 bt + i) (+ 1), i = 0 .. 15
        constexpr int ODD = decltype(odd_tag)::value;
        const int bin = 2 * (bt * 16 + (lane & 15)) + ODD;              // A operand: lane holds (bin of row lane & 15, tap 4 ks + j)
        int idx = (bin * j) % NFFT;
        const int step = (4 * bin) % NFFT;
        f64x4 re = {0.0, 0.0, 0.0, 0.0}, im = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks <= KQ; ++ks) {
            const double2 cs = *reinterpret_cast<const double2*>(&s_tw[2 * idx]);
            if (SKIP & 2) { re[0] += cs.x * (ks < KQ ? G[ks] : 0.0) + F[ks]; im[0] += cs.y * Q[ks] + (ks < KQ ? P[ks] : 0.0); }
            else if (ODD) {
                if (ks < KQ) re = __builtin_amdgcn_mfma_f64_16x16x4f64(cs.x, G[ks], re, 0, 0, 0);
                im = __builtin_amdgcn_mfma_f64_16x16x4f64(cs.y, Q[ks], im, 0, 0, 0);
            } else {
                re = __builtin_amdgcn_mfma_f64_16x16x4f64(cs.x, F[ks], re, 0, 0, 0);
                if (ks < KQ) im = __builtin_amdgcn_mfma_f64_16x16x4f64(cs.y, P[ks], im, 0, 0, 0);
            }
            idx += step;
            idx = idx >= NFFT ? idx - NFFT : idx;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                   // D: column lane & 15 (frame), row (lane >> 4) + 4 r (bin of the tile)
            const int k = 2 * (bt * 16 + (lane >> 4) + 4 * r) + ODD;
            if (to < nfr) feat[((size_t)b * NFREQ + k) * t_stride + to] = (SKIP & 1) ? (float)re[r] + (float)im[r] : log1pf(hypotf((float)re[r], (float)im[r]));
        }
    };
    for (int bt = 0; bt < NQ / 16; ++bt) {
        tile(bt, std::integral_constant<int, 0>{});
        tile(bt, std::integral_constant<int, 1>{});
    }
}

__device__ double block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
    return tot;
}

// Per-clip mean and unbiased std over the clip's own n_freq x frames values (parsers.py:66-70), then
// normalise in place and zero the tail frames.  NSL workgroups per clip: (1) every slice writes its float64 sum and
// sum of squares, (2) every slice adds the NSL partials in a fixed order (deterministic, unlike atomics) and normalises.
constexpr int NSL = 16;

__global__ __launch_bounds__(1024) void clip_stats_kernel(const float* feat, const int64_t* nsamp, int hop, int n_freq, int t_stride, double* stats) {
    __shared__ double sh[16];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nfr = 1 + (int)(nsamp[b] / hop);
    const float* fb = feat + (size_t)b * n_freq * t_stride;
    const int total = n_freq * t_stride;
    double s = 0.0, q = 0.0;
    for (int i = blockIdx.x * 1024 + tid; i < total; i += NSL * 1024)
        if (i % t_stride < nfr) { const double v = (double)fb[i]; s += v; q += v * v; }
    s = block_sum(s, sh);
    q = block_sum(q, sh);
    if (tid == 0) { stats[((size_t)b * NSL + blockIdx.x) * 2] = s; stats[((size_t)b * NSL + blockIdx.x) * 2 + 1] = q; }
}

__global__ __launch_bounds__(1024) void normalize_kernel(float* feat, const int64_t* nsamp, int hop, int n_freq, int t_stride, int normalize,
                                                         const double* stats) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nfr = 1 + (int)(nsamp[b] / hop);
    float* fb = feat + (size_t)b * n_freq * t_stride;
    const int total = n_freq * t_stride;
    float meanf = 0.f, stdf = 1.f;
    if (normalize) {
        const double cnt = (double)n_freq * nfr;
        double sum = 0.0, sq = 0.0;
        for (int k = 0; k < NSL; ++k) { sum += stats[((size_t)b * NSL + k) * 2]; sq += stats[((size_t)b * NSL + k) * 2 + 1]; }
        const double mean = sum / cnt;
        const double var = (sq - cnt * mean * mean) / (cnt - 1.0);
        meanf = (float)mean; stdf = (float)sqrt(var);
    }
    for (int i = blockIdx.x * 1024 + tid; i < total; i += NSL * 1024) {
        const bool in = i % t_stride < nfr;
        if (normalize) fb[i] = in ? (fb[i] - meanf) / stdf : 0.f;
        else if (!in) fb[i] = 0.f;
    }
}

// Streaming parser statistics: np.mean / np.std (population) over the [n_freq][nfr] chunk, one workgroup.
__global__ __launch_bounds__(1024) void chunk_stats_kernel(const float* feat, int n_freq, int nfr, int t_stride, double* out2) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, total = n_freq * nfr;
    double s = 0.0;
    for (int i = tid; i < total; i += 1024) s += (double)feat[(size_t)(i / nfr) * t_stride + i % nfr];
    const double mean = block_sum(s, sh) / (double)total;
    double q = 0.0;
    for (int i = tid; i < total; i += 1024) { const double d = (double)feat[(size_t)(i / nfr) * t_stride + i % nfr] - mean; q += d * d; }
    const double var = block_sum(q, sh) / (double)total;
    if (tid == 0) { out2[0] = mean; out2[1] = sqrt(var); }
}

__global__ void chunk_normalize_kernel(float* feat, int n_freq, int nfr, int t_stride, float mean, float stdv) {
    const int total = n_freq * t_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        feat[i] = (i % t_stride < nfr) ? (feat[i] - mean) / stdv : 0.f;
}

// Hop energies for dsmi_segment: one wave per hop.  numpy sums a contiguous float64 array pairwise:
// blocks of 128 elements, each with 8 strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// blocks combined by halving.  Lane = block*8 + accumulator reproduces that order for step = 128 * 2^k
// (k <= 3 in one pass; longer hops loop over groups of 8 blocks and finish the tree in lane 0's registers).
__global__ __launch_bounds__(256) void hop_energy_kernel(const void* pcm, int dtype, int64_t nhops, int step, double* energy) {
    const int lane = threadIdx.x & 63;
    const int64_t hop = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (hop >= nhops) return;
    const int nblk = step / 128;                 // power of two
    const int blk = lane >> 3, acc = lane & 7;
    double part[8];                              // sums of groups of 8 blocks (step <= 8192)
    int ngroups = 0;
    for (int g0 = 0; g0 < nblk; g0 += 8, ++ngroups) {
        double r = 0.0;
        if (g0 + blk < nblk) {
            const int64_t base = hop * step + (int64_t)(g0 + blk) * 128 + acc;
            const double x0 = ld_sample(pcm, dtype, base);
            r = __dmul_rn(x0, x0);
            for (int j = 1; j < 16; ++j) {
                const double x = ld_sample(pcm, dtype, base + 8 * j);
                r = __dadd_rn(r, __dmul_rn(x, x));
            }
        }
        // accumulators of a block, then blocks pairwise (a + b == b + a exactly, so xor-shuffles keep the tree)
        for (int o = 1; o < 64; o <<= 1) {
            if (o >= 8 && (o >> 3) >= nblk) break;
            r = __dadd_rn(r, __shfl_xor(r, o, 64));
        }
        part[ngroups] = r;
    }
    for (int n = ngroups; n > 1; n >>= 1)
        for (int i = 0; i < n / 2; ++i) part[i] = __dadd_rn(part[2 * i], part[2 * i + 1]);
    if (lane == 0) energy[hop] = sqrt(part[0] / (double)step);
}

// the matrix-pipe form serves n_fft = 320 (every shipped audio_conf: 16 kHz, 20 ms) and one-channel float64 / float32 / int16 samples
// (float64 is what load_audio and the recognizer hand over); any other window length or WAV-frame type, or DSMI_DEBUG_STFT=direct
// (experiments, the parity tests' second form), takes the direct kernel
bool stft_on_mfma(int n_fft, int dtype) {
    static const bool direct = [] { const char* e = exp_env("DSMI_DEBUG_STFT"); return e && std::string(e) == "direct"; }();
    return n_fft == 320 && !direct && (dtype == DSMI_PCM_F64 || dtype == DSMI_PCM_F32 || dtype == DSMI_PCM_I16);
}

void launch_stft_mfma(dim3 grid, hipStream_t s, const void* pcm, int dtype, const int64_t* offs, const int64_t* nsamp, const double* tw,
                      const double* win, int hop, int pad_mode, float* feat, int t_stride) {
#ifdef DSMI_EXPERIMENTS
    static const int skip = [] { const char* e = exp_env("DSMI_DEBUG_STFT_SKIP"); return e ? std::atoi(e) : 0; }();
    if (skip && dtype == DSMI_PCM_F64) {
#define STFT_SKIP(S) case S: hipLaunchKernelGGL((stft_mfma_kernel<320, double, S>), grid, dim3(256), 0, s, (const double*)pcm, offs, nsamp, tw, win, hop, pad_mode, feat, t_stride); return;
        switch (skip) { STFT_SKIP(1) STFT_SKIP(2) STFT_SKIP(4) STFT_SKIP(3) STFT_SKIP(6) STFT_SKIP(7) default: break; }
#undef STFT_SKIP
    }
#endif
    if (dtype == DSMI_PCM_F64) hipLaunchKernelGGL((stft_mfma_kernel<320, double>), grid, dim3(256), 0, s, (const double*)pcm, offs, nsamp, tw, win, hop, pad_mode, feat, t_stride);
    else if (dtype == DSMI_PCM_F32) hipLaunchKernelGGL((stft_mfma_kernel<320, float>), grid, dim3(256), 0, s, (const float*)pcm, offs, nsamp, tw, win, hop, pad_mode, feat, t_stride);
    else hipLaunchKernelGGL((stft_mfma_kernel<320, int16_t>), grid, dim3(256), 0, s, (const int16_t*)pcm, offs, nsamp, tw, win, hop, pad_mode, feat, t_stride);
}

}  // namespace

extern "C" int dsmi_features_stream(dsmi_frontend* f, const void* pcm, int dtype, int64_t n_samples, double* state3, float* feat,
                                    int t_stride, int32_t* frames, void* stream) {
    if (!f) return DSMI_ERR_INVALID;
    auto bad = [&](int code, const char* msg) { f->err = msg; return code; };
    const int base = dtype & 15;
    const bool stereo_ok = base == DSMI_PCM_I16 || base == DSMI_PCM_I24 || base == DSMI_PCM_I32;
    if (!pcm || !state3 || !feat || !frames || dtype < 0 || base > DSMI_PCM_I32 || (dtype & ~(15 | DSMI_PCM_STEREO)) ||
        ((dtype & DSMI_PCM_STEREO) && !stereo_ok))
        return bad(DSMI_ERR_INVALID, "bad streaming features arguments");
    if (n_samples < f->n_fft) return bad(DSMI_ERR_INVALID, "fewer samples than one STFT window (the parser drops such a last chunk, parsers.py:106-110)");
    const int nfr = 1 + (int)((n_samples - f->n_fft) / f->hop);
    if (nfr > t_stride) return bad(DSMI_ERR_INVALID, "t_stride smaller than the chunk's frame count");
    if (hipSetDevice(f->device) != hipSuccess) return bad(DSMI_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    if (f->cap < 2) {       // offs doubles as scratch for the statistics: [0] offset, [cap] n_samples, then two doubles
        if (f->offs) { (void)hipStreamSynchronize(s); (void)hipFree(f->offs); f->offs = nullptr; }
        if (hipMalloc((void**)&f->offs, sizeof(int64_t) * (2 + 2 * NSL) * 2) != hipSuccess) return bad(DSMI_ERR_NOMEM, "hipMalloc failed");
        f->cap = 2;
    }
    const int64_t host[2] = {0, n_samples};
    if (!fe_stage_copy(f, f->offs, &host[0], 1, s) || !fe_stage_copy(f, f->offs + f->cap, &host[1], 1, s))
        return bad(DSMI_ERR_HIP, "staging the chunk's offset / length failed");
    if (stft_on_mfma(f->n_fft, dtype)) {
        launch_stft_mfma(dim3(ceil_div(nfr, MF), 1), s, pcm, dtype, f->offs, f->offs + f->cap, f->tw, f->win, f->hop, PAD_NONE, feat, t_stride);
    } else {
        const size_t lds = sizeof(double) * ((size_t)2 * f->n_fft + (size_t)f->n_fft * FT);
        hipLaunchKernelGGL(stft_logmag_kernel, dim3(ceil_div(nfr, FT), 1), dim3(256), lds, s, pcm, dtype, f->offs, f->offs + f->cap,
                           f->tw, f->win, f->n_fft, f->hop, f->n_freq, PAD_NONE, feat, t_stride);
    }
    double* stats_dev = nullptr;
    if (hipMalloc((void**)&stats_dev, 2 * sizeof(double)) != hipSuccess) return bad(DSMI_ERR_NOMEM, "hipMalloc failed");
    hipLaunchKernelGGL(chunk_stats_kernel, dim3(1), dim3(1024), 0, s, feat, f->n_freq, nfr, t_stride, stats_dev);
    double st[2] = {0, 0};
    const bool ok = hipMemcpyAsync(st, stats_dev, sizeof(st), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(stats_dev);
    if (!ok) return bad(DSMI_ERR_HIP, "streaming feature kernels failed");
    // parsers.py:146-158 (np.mean / np.std of a float32 array are float32 values)
    const double dataset_mean = 5.492418704733003, dataset_std = 1.7552755216970917, alpha_increment = 0.1;
    double input_mean = state3[0], input_std = state3[1], alpha = state3[2];
    alpha += alpha_increment;
    input_mean = (input_mean + (double)(float)st[0]) / 2;
    input_std = (input_std + (double)(float)st[1]) / 2;
    double mean = input_mean, stdv = input_std;
    if (alpha < 1.0) {
        mean = input_mean * alpha + (1 - alpha) * dataset_mean;
        stdv = input_std * alpha + (1 - alpha) * dataset_std;
    }
    state3[0] = input_mean; state3[1] = input_std; state3[2] = alpha;
    hipLaunchKernelGGL(chunk_normalize_kernel, dim3(64), dim3(256), 0, s, feat, f->n_freq, nfr, t_stride, (float)mean, (float)stdv);
    if (hipGetLastError() != hipSuccess) return bad(DSMI_ERR_HIP, "streaming feature kernels failed to launch");
    *frames = nfr;
    return DSMI_OK;
}

extern "C" int dsmi_segment(dsmi_frontend* f, const void* pcm, int dtype, int64_t n_samples, int step, double energy_threshold,
                            int pause_hops, int phrase_hops, int64_t* seg_start, int64_t* seg_end, int max_segments,
                            int* n_segments, double* energies_host, void* stream) {
    if (!f) return DSMI_ERR_INVALID;
    auto bad = [&](int code, const char* msg) { f->err = msg; return code; };
    const int base = dtype & 15;
    const bool stereo_ok = base == DSMI_PCM_I16 || base == DSMI_PCM_I24 || base == DSMI_PCM_I32;
    if (!pcm || !seg_start || !seg_end || !n_segments || max_segments < 0 || n_samples < 0 || dtype < 0 || base > DSMI_PCM_I32 ||
        (dtype & ~(15 | DSMI_PCM_STEREO)) || ((dtype & DSMI_PCM_STEREO) && !stereo_ok))
        return bad(DSMI_ERR_INVALID, "bad segment arguments");
    if (step < 128 || step > 8192 || (step & (step - 1)) != 0) return bad(DSMI_ERR_INVALID, "step must be 128 * 2^k, at most 8192");
    if (hipSetDevice(f->device) != hipSuccess) return bad(DSMI_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    *n_segments = 0;
    // hops i with i*step + step < n_samples  (video_transcribe_simulation.py:94)
    const int64_t nhops = n_samples > step ? (n_samples - 1) / step : 0;
    if (nhops == 0) return DSMI_OK;
    double* e_dev = nullptr;
    if (hipMalloc((void**)&e_dev, sizeof(double) * nhops) != hipSuccess) return bad(DSMI_ERR_NOMEM, "hipMalloc failed");
    hipLaunchKernelGGL(hop_energy_kernel, dim3((unsigned)((nhops + 3) / 4)), dim3(256), 0, s, pcm, dtype, nhops, step, e_dev);
    std::vector<double> e(nhops);
    const bool ok = hipMemcpyAsync(e.data(), e_dev, sizeof(double) * nhops, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess;
    (void)hipFree(e_dev);
    if (!ok) return bad(DSMI_ERR_HIP, "hop energy kernel failed");
    if (energies_host) std::copy(e.begin(), e.end(), energies_host);
    // ---- the script's state machine (:84-143), one pass over the hop energies (host_logic.h)
    const int found = dsmi::segment_phrases(e.data(), nhops, step, energy_threshold, pause_hops, phrase_hops, seg_start, seg_end, max_segments);
    *n_segments = found;
    if (found > max_segments) return bad(DSMI_ERR_CAPACITY, "more phrases than max_segments");
    return DSMI_OK;
}

extern "C" int dsmi_frontend_create(const dsmi_frontend_desc* d, int device, dsmi_frontend** out) {
    if (!d || !out) { g_fe_error = "null argument"; return DSMI_ERR_INVALID; }
    const int n = (int)(d->sample_rate * d->window_size);       // parsers.py:47
    const int hop = (int)(d->sample_rate * d->window_stride);   // parsers.py:48
    if (n < 2 || hop < 1 || d->window < 0 || d->window > 3) { g_fe_error = "audio_conf gives n_fft < 2 or hop < 1"; return DSMI_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
        g_fe_error = "no such HIP device";
        return DSMI_ERR_HIP;
    }
    dsmi_frontend* f = new dsmi_frontend();
    f->desc = *d; f->device = device; f->n_fft = n; f->hop = hop; f->n_freq = n / 2 + 1;
    std::vector<double> tw(2 * (size_t)n), win(n);
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < n; ++j) { tw[2 * j] = std::cos(2.0 * pi * j / n); tw[2 * j + 1] = std::sin(2.0 * pi * j / n); }
    for (int j = 0; j < n; ++j) {   // scipy.signal.windows.* with sym=True (parsers.py:9-10, 27: a callable window)
        const double x = n > 1 ? (double)j / (n - 1) : 0.0;
        switch (d->window) {
            case DSMI_WIN_HANN: win[j] = 0.5 - 0.5 * std::cos(2 * pi * x); break;
            case DSMI_WIN_BLACKMAN: win[j] = 0.42 - 0.5 * std::cos(2 * pi * x) + 0.08 * std::cos(4 * pi * x); break;
            case DSMI_WIN_BARTLETT: win[j] = 1.0 - std::fabs(2.0 * x - 1.0); break;
            default: win[j] = 0.54 - 0.46 * std::cos(2 * pi * x); break;
        }
    }
    if (hipMalloc((void**)&f->tw, sizeof(double) * tw.size()) != hipSuccess ||
        hipMalloc((void**)&f->win, sizeof(double) * win.size()) != hipSuccess ||
        hipMemcpy(f->tw, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->win, win.data(), sizeof(double) * win.size(), hipMemcpyHostToDevice) != hipSuccess) {
        g_fe_error = "frontend: HIP allocation failed";
        delete f;
        return DSMI_ERR_HIP;
    }
    *out = f;
    return DSMI_OK;
}

extern "C" void dsmi_frontend_destroy(dsmi_frontend* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize();
    if (f->tw) (void)hipFree(f->tw);
    if (f->win) (void)hipFree(f->win);
    if (f->offs) (void)hipFree(f->offs);
    if (f->stage) (void)hipHostFree(f->stage);
    for (hipEvent_t e : f->stage_ev) if (e) (void)hipEventDestroy(e);
    delete f;
}

extern "C" const char* dsmi_frontend_last_error(const dsmi_frontend* f) { return f ? f->err.c_str() : g_fe_error.c_str(); }

extern "C" int dsmi_frontend_info(const dsmi_frontend* f, int* n_freq, int* hop, int* device) {
    if (!f) return DSMI_ERR_INVALID;
    if (n_freq) *n_freq = f->n_freq;
    if (hop) *hop = f->hop;
    if (device) *device = f->device;
    return DSMI_OK;
}

extern "C" int dsmi_features(dsmi_frontend* m, const void* pcm, int dtype, const int64_t* n_samples, int B, float* feat,
                             int t_stride, int32_t* frames, void* stream) {
    if (!m) return DSMI_ERR_INVALID;
    auto bad = [&](int code, const char* msg) { m->err = msg; return code; };
    dsmi_frontend* f = m;
    const int base = dtype & 15;
    const bool stereo_ok = base == DSMI_PCM_I16 || base == DSMI_PCM_I24 || base == DSMI_PCM_I32;
    if (!pcm || !n_samples || !feat || B < 1 || dtype < 0 || base > DSMI_PCM_I32 || (dtype & ~(15 | DSMI_PCM_STEREO)) ||
        ((dtype & DSMI_PCM_STEREO) && !stereo_ok))
        return bad(DSMI_ERR_INVALID, "bad features arguments (8-bit and float PCM cannot be stereo)");
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(m->device) != hipSuccess) return bad(DSMI_ERR_HIP, "hipSetDevice failed");
    std::vector<int64_t> host(2 * (size_t)B);
    int64_t off = 0;
    int maxfr = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t N = n_samples[b];
        if (N < 1) return bad(DSMI_ERR_INVALID, "empty clip");
        if (m->desc.pad_mode == DSMI_PAD_REFLECT && N <= m->n_fft / 2) return bad(DSMI_ERR_INVALID, "clip shorter than the reflect padding (n_fft/2)");
        host[b] = off; host[B + b] = N; off += N;
        const int nf = 1 + (int)(N / m->hop);
        if (nf > t_stride) return bad(DSMI_ERR_INVALID, "t_stride smaller than a clip's frame count");
        if (frames) frames[b] = nf;
        maxfr = std::max(maxfr, nf);
    }
    if (B > f->cap) {
        if (f->offs) { (void)hipStreamSynchronize(s); (void)hipFree(f->offs); }
        if (hipMalloc((void**)&f->offs, sizeof(int64_t) * (2 + 2 * NSL) * B) != hipSuccess) return bad(DSMI_ERR_NOMEM, "hipMalloc failed");
        f->cap = B;
    }
    if (!fe_stage_copy(f, f->offs, host.data(), B, s) || !fe_stage_copy(f, f->offs + f->cap, host.data() + B, B, s))
        return bad(DSMI_ERR_HIP, "staging the clips' offsets / lengths failed");
    EvPair ev;
    if (stft_on_mfma(m->n_fft, dtype)) {
        launch_stft_mfma(dim3(ceil_div(maxfr, MF), B), s, pcm, dtype, f->offs, f->offs + f->cap, f->tw, f->win, m->hop, m->desc.pad_mode, feat, t_stride);
    } else {
        const size_t lds = sizeof(double) * ((size_t)2 * m->n_fft + (size_t)m->n_fft * FT);
        DSMI_LAUNCH(stft_logmag_kernel, dim3(ceil_div(maxfr, FT), B), dim3(256), lds, s, ev, pcm, dtype, f->offs, f->offs + f->cap,
                    f->tw, f->win, m->n_fft, m->hop, m->n_freq, m->desc.pad_mode, feat, t_stride);
    }
    double* stats = reinterpret_cast<double*>(f->offs + 2 * (size_t)f->cap);       // [B][NSL][2] behind the offsets / lengths
    if (m->desc.normalize) {
        hipLaunchKernelGGL(clip_stats_kernel, dim3(NSL, B), dim3(1024), 0, s, feat, f->offs + f->cap, m->hop, m->n_freq, t_stride, stats);
    }
    hipLaunchKernelGGL(normalize_kernel, dim3(NSL, B), dim3(1024), 0, s, feat, f->offs + f->cap, m->hop, m->n_freq, t_stride, m->desc.normalize, stats);
    if (hipGetLastError() != hipSuccess) return bad(DSMI_ERR_HIP, "feature kernels failed to launch");
    return DSMI_OK;
}
