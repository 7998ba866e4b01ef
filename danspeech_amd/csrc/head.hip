// Output head and decode-side kernels (gfx950):
//   * head_kernel      SequenceWise(BatchNorm1d -> Linear(H, C, bias=False)) + eval softmax
//                      (reference danspeech/deepspeech/model.py:414-420, 511-514, 84-93), fused
//                      with the direction sum of the last BatchRNN (model.py:121)
//   * lookahead_kernel Lookahead + Hardtanh(0,20) for unidirectional models (model.py:125-148, 407-411)
//   * greedy_kernel    GreedyDecoder.decode: argmax + CTC collapse (decoder.py:183-198, 166-181)
//   * small copies used by the stage-level API
#include "common.h"

namespace dsmi {

std::vector<float> pack_fc(const float* w, int C, int H) {
    const int nt = ceil_div(C, 32), nq = round_up(H, 8) / 8;
    std::vector<float> out((size_t)nt * nq * 64 * 4, 0.f);
    for (int tl = 0; tl < nt; ++tl)
        for (int q = 0; q < nq; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int cls = tl * 32 + (lane & 31), hk = lane >> 5;
                if (cls >= C) continue;
                for (int c = 0; c < 4; ++c) {
                    const int k = 8 * q + 4 * hk + c;
                    if (k < H) out[(((size_t)tl * nq + q) * 64 + lane) * 4 + c] = w[(size_t)cls * H + k];
                }
            }
    return out;
}

struct HeadArgs {
    const float* x1; const float* x2; const float* bn_a; const float* bn_b; const float* wp; float* probs;
    int M, Hs, nq, C, T, B;
};

// One workgroup = 32 rows (t,b) x all classes, its four waves taking every fourth 8-deep slice of K (round 3: four times the
// waves in flight for the same strided row reads -- 95 -> 45 us for cfgA); wave 0 adds the partial tiles and finishes.
// A = W_fc tile (row i = class), B = normalised activations (col j = row of the batch), so a lane ends up with 16 classes
// per tile of ONE row: the softmax needs a single cross-half exchange (lane ^ 32).
template <int NT>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs p) {
    __shared__ float part[3][NT][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, hk = lane >> 5;
    const int m = blockIdx.x * 32 + li;
    const bool valid = m < p.M;
    f32x16 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(p.wp) + lane;
    const size_t xoff = (size_t)(valid ? m : 0) * p.Hs;

    for (int q = wv; q < p.nq; q += 4) {
        const int k = 8 * q + 4 * hk;
        f32x4 xv = *reinterpret_cast<const f32x4*>(p.x1 + xoff + k);
        if (p.x2) xv += *reinterpret_cast<const f32x4*>(p.x2 + xoff + k);
        xv = xv * *reinterpret_cast<const f32x4*>(p.bn_a + k) + *reinterpret_cast<const f32x4*>(p.bn_b + k);
        f32x4 wf[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) wf[c] = wp[((size_t)c * p.nq + q) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < NT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[c][e], xv[e], acc[c], 0, 0, 0);
    }
    if (wv > 0)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wv - 1][c][r][lane] = acc[c][r];
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)             // fixed order: the sum does not depend on scheduling
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] += part[w][c][r][lane];
    // softmax over classes: this lane holds class = 32c + (r&3) + 8(r>>2) + 4hk of row m
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (cls < p.C) mx = fmaxf(mx, acc[c][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            const float e = cls < p.C ? expf(acc[c][r] - mx) : 0.f;
            acc[c][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (!valid) return;
    const int t = m / p.B, b = m % p.B;
    float* out = p.probs + ((size_t)b * p.T + t) * p.C;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            if (cls < p.C) out[cls] = acc[c][r] / sum;
        }
}

void launch_head(const HeadLaunch& h, hipStream_t s) {
    HeadArgs a;
    a.x1 = h.x1; a.x2 = h.x2; a.bn_a = h.bn_a; a.bn_b = h.bn_b; a.wp = h.w_packed; a.probs = h.probs;
    a.M = h.T * h.B; a.Hs = round_up(h.H, 8); a.nq = a.Hs / 8; a.C = h.C; a.T = h.T; a.B = h.B;
    const int nt = ceil_div(h.C, 32);
    dim3 grid(ceil_div(a.M, 32));
    switch (nt) {
        case 1: DSMI_LAUNCH(head_kernel<1>, grid, dim3(256), 0, s, h.ev, a); break;
        case 2: DSMI_LAUNCH(head_kernel<2>, grid, dim3(256), 0, s, h.ev, a); break;
        case 3: DSMI_LAUNCH(head_kernel<3>, grid, dim3(256), 0, s, h.ev, a); break;
        default: DSMI_LAUNCH(head_kernel<4>, grid, dim3(256), 0, s, h.ev, a); break;
    }
}

// ---- Lookahead: depthwise conv over future frames + Hardtanh ------------------------------
__global__ void lookahead_kernel(const float* x, const float* w, float* y, int T, int B, int H, int Hs, int ctx) {
    const size_t n = (size_t)T * B * Hs;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int h = idx % Hs;
        const size_t tb = idx / Hs;
        const int t = tb / B;
        float acc = 0.f;
        if (h < H) {
            for (int k = 0; k < ctx; ++k)
                if (t + k < T) acc += x[idx + (size_t)k * B * Hs] * w[h * ctx + k];
            acc = fminf(fmaxf(acc, 0.f), 20.f);
        }
        y[idx] = acc;
    }
}

void launch_lookahead(const float* x, const float* w, float* y, int T, int B, int H, int context, hipStream_t s) {
    const int Hs = round_up(H, 8);
    const size_t n = (size_t)T * B * Hs;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(lookahead_kernel, dim3(blocks), dim3(256), 0, s, x, w, y, T, B, H, Hs, context);
}

// ---- greedy CTC decode: one workgroup per utterance ---------------------------------------
__global__ __launch_bounds__(256) void greedy_kernel(const float* probs, const int32_t* sizes, int T, int C, int blank,
                                                     int32_t* raw, int32_t* ids, int32_t* offs, int32_t* n_out) {
    __shared__ int cnt[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = probs + (size_t)b * T * C;
    int32_t* rb = raw + (size_t)b * T;
    // argmax, first maximum wins (torch.max over dim 2)
    for (int t = tid; t < T; t += 256) {
        const float* row = pb + (size_t)t * C;
        float best = row[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = row[c];
            if (v > best) { best = v; bi = c; }
        }
        rb[t] = bi;
    }
    __syncthreads();
    const int size = sizes ? min(sizes[b], T) : T;
    const int per = (T + 255) / 256;
    const int s0 = tid * per, s1 = min(s0 + per, size);
    int local = 0;
    for (int t = s0; t < s1; ++t) {
        const int c = rb[t];
        local += (c != blank && (t == 0 || c != rb[t - 1])) ? 1 : 0;
    }
    cnt[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive Hillis-Steele scan
        const int v = tid >= off ? cnt[tid - off] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    int pos = cnt[tid] - local;
    for (int t = s0; t < s1; ++t) {
        const int c = rb[t];
        if (c != blank && (t == 0 || c != rb[t - 1])) {
            ids[(size_t)b * T + pos] = c;
            offs[(size_t)b * T + pos] = t;
            ++pos;
        }
    }
    if (tid == 255) n_out[b] = cnt[255];
}

void launch_greedy(const float* probs, const int32_t* sizes_dev, int B, int T, int C, int blank,
                   int32_t* raw, int32_t* ids, int32_t* offsets, int32_t* n_out, hipStream_t s) {
    hipLaunchKernelGGL(greedy_kernel, dim3(B), dim3(256), 0, s, probs, sizes_dev, T, C, blank, raw, ids, offsets, n_out);
}

// ---- small helpers ---------------------------------------------------------------------------
__global__ void add2_kernel(const float* a, const float* b, float* y, size_t rows, int H, int Hs) {
    const size_t n = rows * H;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / H;
        const int h = idx % H;
        float v = a[r * Hs + h];
        if (b) v += b[r * Hs + h];
        y[idx] = v;
    }
}
void launch_add2(const float* a, const float* b, float* y, size_t rows, int H, int Hs, hipStream_t s) {
    const size_t n = rows * H;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(add2_kernel, dim3(blocks), dim3(256), 0, s, a, b, y, rows, H, Hs);
}

__global__ void pad_rows_kernel(const float* x, float* y, size_t rows, int I, int Is) {
    const size_t n = rows * Is;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / Is;
        const int i = idx % Is;
        y[idx] = i < I ? x[r * I + i] : 0.f;
    }
}
void launch_pad_rows(const float* x, float* y, size_t rows, int I, int Is, hipStream_t s) {
    const size_t n = rows * Is;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, s, x, y, rows, I, Is);
}

}  // namespace dsmi
