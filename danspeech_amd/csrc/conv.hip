// Fused Conv2d + bias + BatchNorm2d(eval) + Hardtanh(0,20) + time mask as an implicit GEMM on
// the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces one (Conv2d, BatchNorm2d, Hardtanh) triple of the reference's conv stack plus the
// MaskConv zeroing that follows each of the three modules
// (danspeech/deepspeech/model.py:65-81 and 358-396).  Masking only ever replaces values by 0
// and nothing downstream of a masked frame feeds an unmasked one inside the triple, so
//     y[b,co,f,t] = t < out_len[b] ? clip((conv + bias) * bn_a + bn_b, 0, 20) : 0
// is exactly what the three mask passes leave behind.
//
// MFMA roles: A = weights (row i = output channel), B = input patch (col j = output time
// step), so D[co][t] puts consecutive t on consecutive lanes and the epilogue stores 128-byte
// runs along time.  The instruction's K = 2 is one kernel column kt at two adjacent kernel
// rows: lane half h = lane>>5 takes kf = 2q+h (KF is odd, the missing last row has zero
// weights).  One workgroup = 4 waves = 4 consecutive output rows f of one clip x 64 output
// steps x all output channels; the input rows they share are staged once in LDS per block
// of 4 input channels (double-buffered).  Weights are pre-packed on the host to
// [co-tile][ci][q][lane][12] so each lane fetches its 11 kt taps with three 16-byte loads
// straight from L2 (every workgroup streams the same packed weights).
#include "common.h"

namespace dsmi {

template <int L> struct CG;
template <> struct CG<0> { static constexpr int CI = 1, CO = 32, KF = 41, KT = 11, SF = 2, ST = 2, PF = 20, PT = 5, CC = 1; };
template <> struct CG<1> { static constexpr int CI = 32, CO = 32, KF = 21, KT = 11, SF = 2, ST = 1, PF = 10, PT = 5, CC = 4; };
template <> struct CG<2> { static constexpr int CI = 32, CO = 96, KF = 21, KT = 11, SF = 2, ST = 1, PF = 10, PT = 5, CC = 4; };

constexpr int NF = 4;      // output rows per workgroup (one per wave)
constexpr int NTT = 2;     // 32-step time tiles per wave
constexpr int TT = 32 * NTT;

template <int L> struct CD {
    using G = CG<L>;
    static constexpr int NQ = (G::KF + 1) / 2;                 // kernel-row pairs
    static constexpr int ROWS = G::SF * (NF - 1) + 2 * NQ;     // staged input rows
    static constexpr int WI = G::ST * (TT - 1) + G::KT;        // staged input columns
    static constexpr int RS = (WI + 3) / 4 * 4 + 1;            // row stride (odd: spreads banks for ST=2)
    static constexpr int NCO = G::CO / 32;
    static constexpr int CHUNK = G::CC * ROWS * RS;            // floats per staged chunk
    static constexpr int NBUF = (G::CI > G::CC) ? 2 : 1;
};

std::vector<float> pack_conv_weights(const float* w, int layer) {
    const ConvSpec& s = kConvSpecs[layer];
    const int nq = (s.kf + 1) / 2, nco = s.co / 32;
    std::vector<float> out((size_t)nco * s.ci * nq * 64 * 12, 0.f);
    for (int ct = 0; ct < nco; ++ct)
        for (int ci = 0; ci < s.ci; ++ci)
            for (int q = 0; q < nq; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = ct * 32 + (lane & 31), kf = 2 * q + (lane >> 5);
                    if (kf >= s.kf) continue;
                    for (int kt = 0; kt < s.kt; ++kt)
                        out[((((size_t)ct * s.ci + ci) * nq + q) * 64 + lane) * 12 + kt] =
                            w[(((size_t)co * s.ci + ci) * s.kf + kf) * s.kt + kt];
                }
    return out;
}

struct ConvArgs {
    const float* x; float* y; const float* wp; const float* bias; const float* bn_a; const float* bn_b;
    const int32_t* out_lens;
    int B, fi, fo, ti, to, xs, ys;
    uint16_t* y_sp;    // when set: write [b][f][plane 2][t][32] fp16 terms (hi, lo unscaled) instead of y (feeds conv_split.hip)
};

using f16x4c = __attribute__((ext_vector_type(4))) _Float16;
using u32x4c = __attribute__((ext_vector_type(4))) unsigned int;

template <int L>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvArgs p) {
    using G = CG<L>;
    using D = CD<L>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, hk = lane >> 5;
    const int t0 = blockIdx.x * TT, f0 = blockIdx.y * NF, b = blockIdx.z;
    const int f = f0 + wv;
    const int olen = p.out_lens[b];

    if (t0 >= olen && p.y_sp) {   // fully masked tile, split channels-last output
        for (int idx = tid; idx < NF * 2 * TT * 4; idx += 256) {
            const int part = idx & 3, tl = (idx >> 2) % TT, pl = (idx / (4 * TT)) % 2, ff = idx / (4 * TT * 2);
            if (f0 + ff < p.fo && t0 + tl < p.to)
                *reinterpret_cast<u32x4c*>(reinterpret_cast<_Float16*>(p.y_sp) + ((((size_t)b * p.fo + f0 + ff) * 2 + pl) * p.to + t0 + tl) * 32 + part * 8) = u32x4c{0, 0, 0, 0};
        }
        return;
    }
    if (t0 >= olen) {   // fully masked tile: zeros, no arithmetic
        for (int idx = tid; idx < G::CO * NF * TT; idx += 256) {
            const int tl = idx % TT, ff = (idx / TT) % NF, co = idx / (TT * NF);
            if (f0 + ff < p.fo && t0 + tl < p.to)
                p.y[(((size_t)b * G::CO + co) * p.fo + f0 + ff) * p.ys + t0 + tl] = 0.f;
        }
        return;
    }

    f32x16 acc[D::NCO][NTT];
#pragma unroll
    for (int c = 0; c < D::NCO; ++c)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][tt][r] = 0.f;

    const int fin0 = G::SF * f0 - G::PF;      // input row of staged row 0
    const int tin0 = G::ST * t0 - G::PT;      // input column of staged column 0
    const f32x4* wbase = reinterpret_cast<const f32x4*>(p.wp) + (size_t)lane * 3;
    constexpr int NCHUNK = G::CI / G::CC;

    for (int ch = 0; ch < NCHUNK; ++ch) {
        float* buf = smem + (D::NBUF == 2 ? (ch & 1) * D::CHUNK : 0);
        // ---- stage CC input channels: ROWS x WI window, zero outside the tensor
        for (int idx = tid; idx < G::CC * D::ROWS * D::RS; idx += 256) {
            const int col = idx % D::RS, row = (idx / D::RS) % D::ROWS, cc = idx / (D::RS * D::ROWS);
            const int fin = fin0 + row, tin = tin0 + col;
            float v = 0.f;
            if (col < D::WI && fin >= 0 && fin < p.fi && tin >= 0 && tin < p.ti)
                v = p.x[(((size_t)b * G::CI + ch * G::CC + cc) * p.fi + fin) * p.xs + tin];
            buf[idx] = v;
        }
        __syncthreads();
        if (f < p.fo) {
            const float* rowbase = buf + (G::SF * wv + hk) * D::RS + G::ST * li;
#pragma unroll 1
            for (int cc = 0; cc < G::CC; ++cc) {
                const int ci = ch * G::CC + cc;
#pragma unroll 1
                for (int q = 0; q < D::NQ; ++q) {
                    f32x4 wr[D::NCO][3];
#pragma unroll
                    for (int c = 0; c < D::NCO; ++c) {
                        const f32x4* wpq = wbase + (((size_t)c * G::CI + ci) * D::NQ + q) * 64 * 3;
                        wr[c][0] = wpq[0]; wr[c][1] = wpq[1]; wr[c][2] = wpq[2];
                    }
                    const float* src = rowbase + (cc * D::ROWS + 2 * q) * D::RS;
                    float xv[NTT][G::KT];
#pragma unroll
                    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                        for (int kt = 0; kt < G::KT; ++kt) xv[tt][kt] = src[G::ST * 32 * tt + kt];
#pragma unroll
                    for (int kt = 0; kt < G::KT; ++kt)
#pragma unroll
                        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                            for (int c = 0; c < D::NCO; ++c)
                                acc[c][tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    wr[c][kt >> 2][kt & 3], xv[tt][kt], acc[c][tt], 0, 0, 0);
                }
            }
        }
        if (D::NBUF == 1) __syncthreads();
    }

    if (f >= p.fo) return;
    if (p.y_sp) {
        // ---- epilogue for a split-fp16 consumer: 4 consecutive channels per store, x = hi + lo, the lo term UNSCALED as
        // conv_split.hip and conv1_split.hip's epilogue have it since round 3 (not reached today: the fp32 first layer is tied to an
        // fp32 second one by DSMI_DENSE_MODE; kept consistent so that routing it into a split layer cannot be silently 2x off)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int t = t0 + tt * 32 + li;
            if (t >= p.to) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4c h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = q + 8 * g + 4 * hk;
                    float v = (acc[0][tt][4 * g + q] + p.bias[co]) * p.bn_a[co] + p.bn_b[co];
                    v = fminf(fmaxf(v, 0.f), 20.f);
                    v = t < olen ? v : 0.f;
                    const _Float16 a = (_Float16)v;
                    h[q] = a; l[q] = (_Float16)(v - (float)a);
                }
                _Float16* base = reinterpret_cast<_Float16*>(p.y_sp) + ((((size_t)b * p.fo + f) * 2) * (size_t)p.to + t) * 32 + 8 * g + 4 * hk;
                *reinterpret_cast<f16x4c*>(base) = h;
                *reinterpret_cast<f16x4c*>(base + (size_t)p.to * 32) = l;
            }
        }
        return;
    }
    // ---- epilogue: D[i][j]: i = co (regs), j = lane&31 = time
#pragma unroll
    for (int c = 0; c < D::NCO; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            const float bi = p.bias[co], a = p.bn_a[co], bb = p.bn_b[co];
            float* yrow = p.y + (((size_t)b * G::CO + co) * p.fo + f) * p.ys;
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const int t = t0 + tt * 32 + li;
                if (t >= p.to) continue;
                float v = (acc[c][tt][r] + bi) * a + bb;
                v = fminf(fmaxf(v, 0.f), 20.f);
                yrow[t] = t < olen ? v : 0.f;
            }
        }
}

template <int L>
static void launch_layer(const ConvLaunch& c, hipStream_t s) {
    using D = CD<L>;
    ConvArgs a{c.x, c.y, c.wp, c.bias, c.bn_a, c.bn_b, c.out_lens_dev, c.B, c.fi, c.fo, c.ti, c.to, c.xs, c.ys, c.y_sp};
    dim3 grid(ceil_div(c.to, TT), ceil_div(c.fo, NF), c.B);
    const size_t lds = (size_t)D::NBUF * D::CHUNK * sizeof(float);
    DSMI_LAUNCH(conv_kernel<L>, grid, dim3(256), lds, s, c.ev, a);
}

void launch_conv(const ConvLaunch& c, hipStream_t s) {
    switch (c.layer) {
        case 0: launch_layer<0>(c, s); break;
        case 1: launch_layer<1>(c, s); break;
        default: launch_layer<2>(c, s); break;
    }
}

}  // namespace dsmi
