// Conv2d (32 input channels) + bias + BatchNorm2d(eval) + Hardtanh(0,20) + time mask on the bf16
// MFMA with three-term split operands (fp32-grade, see gemm.hip / tools/exp/bf16x6_test.hip).
//
// Replaces the 2nd and 3rd (Conv2d, BatchNorm2d, Hardtanh) triples of the reference's conv stack
// and their MaskConv zeroing (danspeech/deepspeech/model.py:65-81, 372-374, 389-391), i.e. the
// layers that hold 89 % / 97 % of the conv FLOPs.  conv.hip (fp32 MFMA) keeps the
// first layer (1 input channel) and remains the plain fp32 statement of all three.
//
// Implicit GEMM with the INPUT CHANNELS as the MFMA's K: one v_mfma_f32_32x32x16_bf16 contracts
// 16 input channels of one kernel tap (kf, kt); A = weights (row = output channel, pre-split and
// packed on the host in lane order), B = input (col = output time step).  The input arrives
// already split and channels-last, [b][f][plane][t][32 ci] bf16, written by the previous layer's
// epilogue, so that a lane's 8-channel B fragment is ONE aligned 16-byte LDS read and staging is a
// plain copy.  Workgroup = 4 waves = 4 consecutive output rows f x 64 output steps x all output
// channels; per kernel row kf the four input rows it needs (one per wave) are staged in LDS with
// an 80-byte pitch per time step (16 consecutive steps hit 64 distinct banks), the next kf's
// rows are in flight in registers during the MFMAs.
#include "common.h"

namespace dsmi {

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int BNF = 4;                 // output rows per workgroup (one per wave)
constexpr int BTT = 64;                // output steps per workgroup (2 MFMA column tiles per wave)
constexpr int KT = 11, KF = 21, SF = 2, PF = 10, PT = 5, CI = 32;
constexpr int WIN = BTT + KT - 1;      // staged time steps per row: 74
constexpr int PITCH = 40;              // bf16 per staged time step (32 ci + 8 pad = 80 B)
constexpr int ROWPLANE = WIN * PITCH;  // bf16 per (row, plane)
constexpr int NCHUNK = BNF * 3 * WIN * 4;           // 16-byte chunks staged per kf: 3552
constexpr int CPT = (NCHUNK + 255) / 256;           // chunks per thread: 14

struct Conv3Args {
    const uint16_t* x3; const uint16_t* wp3; const float* bias; const float* bn_a; const float* bn_b;
    const int32_t* out_lens; float* y; uint16_t* y3;
    int B, fi, fo, ti, to, ys;
};

__device__ __forceinline__ void csplit3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// Shared epilogue piece: 4 consecutive channels of one (b, f, t) -> three 8-byte stores, channels-last split.
__device__ __forceinline__ void store_split4(uint16_t* y3, size_t bf_index, int t_stride, int t, int c0, const float (&v)[4]) {
    bf16x4 h, m, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) { __bf16 a, b2, d; csplit3(v[c], a, b2, d); h[c] = a; m[c] = b2; l[c] = d; }
    __bf16* base = reinterpret_cast<__bf16*>(y3) + ((bf_index * 3) * (size_t)t_stride + t) * 32 + c0;
    *reinterpret_cast<bf16x4*>(base) = h;
    *reinterpret_cast<bf16x4*>(base + (size_t)t_stride * 32) = m;
    *reinterpret_cast<bf16x4*>(base + (size_t)2 * t_stride * 32) = l;
}

template <int NCO, bool SPLIT_OUT>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(Conv3Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
    __bf16* Xs = reinterpret_cast<__bf16*>(csm);      // [4 rows][3 planes][WIN][PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hk = lane >> 5;
    const int t0 = blockIdx.x * BTT, f0 = blockIdx.y * BNF, b = blockIdx.z;
    const int f = f0 + wv;
    const int olen = p.out_lens[b];
    constexpr int CO = 32 * NCO;

    if (t0 >= olen) {     // fully masked tile: zeros in the consumer's format
        if (SPLIT_OUT) {
            for (int idx = tid; idx < BNF * 3 * BTT * 4; idx += 256) {
                const int part = idx & 3, tl = (idx >> 2) % BTT, pl = (idx / (4 * BTT)) % 3, ff = idx / (4 * BTT * 3);
                if (f0 + ff < p.fo && t0 + tl < p.to)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.y3) + ((((size_t)b * p.fo + f0 + ff) * 3 + pl) * p.to + t0 + tl) * 32 + part * 8) = u32x4{0, 0, 0, 0};
            }
        } else {
            for (int idx = tid; idx < CO * BNF * BTT; idx += 256) {
                const int tl = idx % BTT, ff = (idx / BTT) % BNF, co = idx / (BTT * BNF);
                if (f0 + ff < p.fo && t0 + tl < p.to) p.y[(((size_t)b * CO + co) * p.fo + f0 + ff) * p.ys + t0 + tl] = 0.f;
            }
        }
        return;
    }

    f32x16 acc[NCO][2];
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][tt][r] = 0.f;

    // ---- staging: chunk id -> (row, plane, step, 16-byte part); global source is a plain copy
    u32x4 stg[CPT];
    auto load_rows = [&](int kf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            u32x4 v = {0, 0, 0, 0};
            if (c < NCHUNK) {
                const int part = c & 3, st = (c >> 2) % WIN, rp = (c >> 2) / WIN;     // rp = row * 3 + plane
                const int row = rp / 3, pl = rp - row * 3;
                const int fin = SF * (f0 + row) + kf - PF, tin = t0 - PT + st;
                if (fin >= 0 && fin < p.fi && tin >= 0 && tin < p.ti)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(p.x3) +
                                                        ((((size_t)b * p.fi + fin) * 3 + pl) * p.ti + tin) * 32 + part * 8);
            }
            stg[i] = v;
        }
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            if (c < NCHUNK) {
                const int part = c & 3, st = (c >> 2) % WIN, rp = (c >> 2) / WIN;
                *reinterpret_cast<u32x4*>(Xs + rp * ROWPLANE + st * PITCH + part * 8) = stg[i];
            }
        }
    };

    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.wp3) + lane;
    const __bf16* xrow = Xs + (wv * 3) * ROWPLANE + li * PITCH + hk * 8;

    load_rows(0);
    for (int kf = 0; kf < KF; ++kf) {
        store_rows();
        __syncthreads();
        if (kf + 1 < KF) load_rows(kf + 1);
        if (f < p.fo) {
#pragma unroll 1
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    bf16x8 wf[NCO][3];
#pragma unroll
                    for (int c = 0; c < NCO; ++c)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[c][pl] = __builtin_bit_cast(bf16x8, wbase[((((size_t)kf * KT + kt) * 2 + half) * NCO + c) * 3 * 64 + pl * 64]);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        bf16x8 xf[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            xf[pl] = *reinterpret_cast<const bf16x8*>(xrow + pl * ROWPLANE + (tt * 32 + kt) * PITCH + half * 16);
#pragma unroll
                        for (int c = 0; c < NCO; ++c) {
                            f32x16 a = acc[c][tt];
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][2], xf[0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[2], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[0], a, 0, 0, 0);
                            acc[c][tt] = a;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    if (f >= p.fo) return;
    // ---- epilogue: D[i][j]: i = co (regs), j = lane&31 = time
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = t0 + tt * 32 + li;
            if (t >= p.to) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = c * 32 + q + 8 * g + 4 * hk;
                    float x = (acc[c][tt][4 * g + q] + p.bias[co]) * p.bn_a[co] + p.bn_b[co];
                    x = fminf(fmaxf(x, 0.f), 20.f);
                    v[q] = t < olen ? x : 0.f;
                }
                if (SPLIT_OUT) {
                    store_split4(p.y3, (size_t)b * p.fo + f, p.to, t, 8 * g + 4 * hk, v);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        p.y[(((size_t)b * CO + c * 32 + q + 8 * g + 4 * hk) * p.fo + f) * p.ys + t] = v[q];
                }
            }
        }
}

inline uint16_t c_bf16_rne(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float c_bf16_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace

// w [co][32][21][11] fp32 -> [kf][kt][half][co-tile][plane][lane][8] bf16 terms; lane (i = co in tile, h)
// element e holds input channel 16*half + 8*h + e.
std::vector<uint16_t> pack_conv_w3(const float* w, int co_total) {
    const int nco = co_total / 32;
    std::vector<uint16_t> out((size_t)KF * KT * 2 * nco * 3 * 64 * 8, 0);
    for (int kf = 0; kf < KF; ++kf)
        for (int kt = 0; kt < KT; ++kt)
            for (int half = 0; half < 2; ++half)
                for (int ct = 0; ct < nco; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = ct * 32 + (lane & 31), ci = 16 * half + 8 * (lane >> 5) + e;
                            const float x = w[(((size_t)co * CI + ci) * KF + kf) * KT + kt];
                            const uint16_t h1 = c_bf16_rne(x);
                            const float r1 = x - c_bf16_f32(h1);
                            const uint16_t h2 = c_bf16_rne(r1);
                            const uint16_t h3 = c_bf16_rne(r1 - c_bf16_f32(h2));
                            const size_t base = ((((((size_t)kf * KT + kt) * 2 + half) * nco + ct) * 3) * 64 + lane) * 8 + e;
                            out[base] = h1; out[base + 512] = h2; out[base + 1024] = h3;
                        }
    return out;
}

void launch_conv_bf16(const ConvBf16Launch& c, hipStream_t s) {
    Conv3Args a{c.x3, c.wp3, c.bias, c.bn_a, c.bn_b, c.out_lens_dev, c.y, c.y3, c.B, c.fi, c.fo, c.ti, c.to, c.ys};
    const dim3 grid(ceil_div(c.to, BTT), ceil_div(c.fo, BNF), c.B);
    const size_t lds = (size_t)BNF * 3 * ROWPLANE * 2;     // 71,040 B
#define LAUNCH_C(NCO, SP)                                                                                        \
    do {                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_kernel<NCO, SP>),                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
        DSMI_LAUNCH((conv_bf16_kernel<NCO, SP>), grid, dim3(256), lds, s, c.ev, a);                               \
    } while (0)
    if (c.co == 32) { if (c.y3) LAUNCH_C(1, true); else LAUNCH_C(1, false); }
    else { if (c.y3) LAUNCH_C(3, true); else LAUNCH_C(3, false); }
#undef LAUNCH_C
}

}  // namespace dsmi
