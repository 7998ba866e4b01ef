// First conv layer -- Conv2d(1 -> 32, k = 41 x 11, stride 2 x 2, pad 20 x 5) + bias + BatchNorm2d(eval) + Hardtanh(0, 20) +
// MaskConv (reference danspeech/deepspeech/model.py:65-81, 358-371) -- on the fp16 MFMA with two-term split operands
// (x = hi + lo * 2^-11, three products, two fp32 accumulators: gemm.hip / conv_split.hip), instead of the fp32 MFMA of
// conv.hip, which runs at 1/16 of the fp16 rate (conv.hip stays as the plain fp32 statement and the fallback).
//
// One input channel gives the MFMA's K nothing to contract, so K is the kernel's TIME axis: one v_mfma_f32_32x32x16_f16
// contracts the 11 taps of one kernel row kf (padded to 16 with zero weights): A = weights [32 output channels][16 taps],
// B = input patch [16 taps][32 output steps].  A lane's B fragment is 8 consecutive input samples of row 2 f + kf - 20
// starting at 2 t - 5 + 8 hk: the time stride of 2 makes the start of neighbouring output steps 4 bytes apart, so
//   * the two column tiles of a wave take the EVEN and the ODD output steps (t = t0 + 2 li, t0 + 2 li + 1), and
//   * the staged rows exist twice in LDS, the second copy shifted by two samples,
// which makes every fragment two aligned 8-byte LDS reads at consecutive 8-byte addresses across the lanes (no bank
// conflicts).  The window of a workgroup (8 output rows x 64 output steps: 55 input rows x 137 samples) is split into its
// two fp16 terms once while it is staged.  Weight fragments come from L2 (82 KB for all 41 kernel rows), one row ahead.
#include "common.h"

namespace dsmi {

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

constexpr int KF = 41, KT = 11, SF = 2, ST = 2, PF = 20, PT = 5, CO = 32;
constexpr int C1NF = 8;                        // output rows per workgroup (one per wave: 8 waves)
constexpr int C1NT = C1NF * 64;                // threads per workgroup
constexpr int C1TT = 64;                       // output steps per workgroup: an even and an odd column tile of 32
constexpr int ROWS = SF * (C1NF - 1) + KF;     // staged input rows: 55 (the 41-row halo is shared by 8 output rows)
constexpr int WIN = ST * (C1TT - 1) + KT + 5;  // samples a fragment may touch: 2 * 63 + 16 = 142
constexpr int PITCH = 144;                     // halfs per staged row (288 B: rows start 8-byte aligned)
constexpr int COPY = ROWS * PITCH;             // halfs per (plane, copy)
constexpr size_t C1_LDS = (size_t)2 * 2 * COPY * 2;   // planes x copies x halfs x 2 B = 63 360 B: two workgroups per CU

struct Conv1Args {
    const float* x; const uint16_t* wp; const float* bias; const float* bn_a; const float* bn_b; const int32_t* out_lens;
    float* y; uint16_t* y_sp;
    int B, fi, fo, ti, to, xs, ys;
};

template <bool SPLIT_OUT>
__global__ __launch_bounds__(C1NT, 4) void conv1_f16x3_kernel(Conv1Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char c1sm[];
    _Float16* Xs = reinterpret_cast<_Float16*>(c1sm);      // [plane 2][copy 2][ROWS][PITCH]; copy 1 holds the row shifted by 2 samples
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hk = lane >> 5;
    const int t0 = blockIdx.x * C1TT, f0 = blockIdx.y * C1NF, b = blockIdx.z;
    const int f = f0 + wv;
    const int olen = p.out_lens[b];

    if (t0 >= olen) {     // fully masked tile: zeros in the consumer's format
        if (SPLIT_OUT) {
            for (int idx = tid; idx < C1NF * 2 * C1TT * 4; idx += C1NT) {
                const int part = idx & 3, tl = (idx >> 2) % C1TT, pl = (idx / (4 * C1TT)) % 2, ff = idx / (4 * C1TT * 2);
                if (f0 + ff < p.fo && t0 + tl < p.to)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<_Float16*>(p.y_sp) + ((((size_t)b * p.fo + f0 + ff) * 2 + pl) * p.to + t0 + tl) * 32 + part * 8) = u32x4{0, 0, 0, 0};
            }
        } else {
            for (int idx = tid; idx < CO * C1NF * C1TT; idx += C1NT) {
                const int tl = idx % C1TT, ff = (idx / C1TT) % C1NF, co = idx / (C1TT * C1NF);
                if (f0 + ff < p.fo && t0 + tl < p.to) p.y[(((size_t)b * CO + co) * p.fo + f0 + ff) * p.ys + t0 + tl] = 0.f;
            }
        }
        return;
    }

    // ---- stage the window, split into (hi, lo * 2^11), twice: thread -> (row, pair of adjacent samples)
    const int fin0 = SF * f0 - PF, tin0 = ST * t0 - PT;        // input row / column of staged (0, 0)
    for (int idx = tid; idx < ROWS * (PITCH / 2); idx += C1NT) {
        const int row = idx / (PITCH / 2), q = 2 * (idx - row * (PITCH / 2));
        const int fin = fin0 + row;
        float v[2] = {0.f, 0.f};
        if (fin >= 0 && fin < p.fi) {
            const float* src = p.x + ((size_t)b * p.fi + fin) * p.xs;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int tin = tin0 + q + e;
                if (tin >= 0 && tin < p.ti && q + e < WIN) v[e] = src[tin];
            }
        }
        f16x2 h, l;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const _Float16 hi = (_Float16)v[e];
            h[e] = hi; l[e] = (_Float16)((v[e] - (float)hi) * kLoScale);
        }
        _Float16* r0 = Xs + row * PITCH + q;
        *reinterpret_cast<f16x2*>(r0) = h;                                   // plane 0, copy 0
        *reinterpret_cast<f16x2*>(r0 + 2 * COPY) = l;                        // plane 1, copy 0
        if (q >= 2) {
            *reinterpret_cast<f16x2*>(r0 + COPY - 2) = h;                    // plane 0, copy 1: sample q at index q - 2
            *reinterpret_cast<f16x2*>(r0 + 3 * COPY - 2) = l;                // plane 1, copy 1
        }
    }
    __syncthreads();
    if (f >= p.fo) return;

    f32x16 acc[2], acl[2];      // [column tile: even / odd steps]  hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[tt][r] = 0.f; acl[tt][r] = 0.f; }

    // B fragment of column tile tt, plane pl, kernel row kf: samples 2 (2 li + tt) + 8 hk .. + 7 of staged row 2 wv + kf
    //   tt = 0: copy 0 at index 4 li + 8 hk;   tt = 1: copy 1 at index (4 li + 2 + 8 hk) - 2 = 4 li + 8 hk
    const _Float16* xb = Xs + (SF * wv) * PITCH + 4 * li + 8 * hk;
    const u32x4* wq = reinterpret_cast<const u32x4*>(p.wp) + lane;          // [kf][plane][lane] 16-byte fragments
    u32x4 wn[2] = {wq[0], wq[64]};
#pragma unroll 1
    for (int kf = 0; kf < KF; ++kf) {
        const f16x8 wh = __builtin_bit_cast(f16x8, wn[0]), wl = __builtin_bit_cast(f16x8, wn[1]);
        if (kf + 1 < KF) { wn[0] = wq[(size_t)(kf + 1) * 128]; wn[1] = wq[(size_t)(kf + 1) * 128 + 64]; }     // next row's weights, from L2
        const _Float16* xr = xb + kf * PITCH;
        f16x8 xh[2], xl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const f16x4 a0 = *reinterpret_cast<const f16x4*>(xr + tt * COPY), a1 = *reinterpret_cast<const f16x4*>(xr + tt * COPY + 4);
            const f16x4 b0 = *reinterpret_cast<const f16x4*>(xr + (2 + tt) * COPY), b1 = *reinterpret_cast<const f16x4*>(xr + (2 + tt) * COPY + 4);
            xh[tt] = f16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            xl[tt] = f16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acl[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[tt], acl[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[tt], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acl[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[tt], acl[tt], 0, 0, 0);
    }

    // ---- epilogue: D[i = output channel][j = li]: output step t0 + 2 li + tt
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int t = t0 + 2 * li + tt;
        if (t >= p.to) continue;
        if (SPLIT_OUT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = q + 8 * g + 4 * hk;
                    float v = (acc[tt][4 * g + q] + acl[tt][4 * g + q] * kLoInv + p.bias[co]) * p.bn_a[co] + p.bn_b[co];
                    v = fminf(fmaxf(v, 0.f), 20.f);
                    v = t < olen ? v : 0.f;
                    const _Float16 a = (_Float16)v;
                    h[q] = a; l[q] = (_Float16)(v - (float)a);        // UNSCALED lo term: the operand format of conv_split.hip
                }
                _Float16* base = reinterpret_cast<_Float16*>(p.y_sp) + ((((size_t)b * p.fo + f) * 2) * (size_t)p.to + t) * 32 + 8 * g + 4 * hk;
                *reinterpret_cast<f16x4*>(base) = h;
                *reinterpret_cast<f16x4*>(base + (size_t)p.to * 32) = l;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * hk;
                float v = (acc[tt][r] + acl[tt][r] * kLoInv + p.bias[co]) * p.bn_a[co] + p.bn_b[co];
                v = fminf(fmaxf(v, 0.f), 20.f);
                p.y[(((size_t)b * CO + co) * p.fo + f) * p.ys + t] = t < olen ? v : 0.f;
            }
        }
    }
}

inline uint16_t c1_bits(_Float16 h) {
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

// w [32][1][41][11] fp32 -> [kf][plane][lane 64][8] fp16 terms (hi, lo * 2^11): lane (co = lane & 31, hk = lane >> 5) element e
// holds tap kt = 8 hk + e of kernel row kf (0 for kt >= 11).
std::vector<uint16_t> pack_conv1_w_split(const float* w) {
    std::vector<uint16_t> out((size_t)KF * 2 * 64 * 8, 0);
    for (int kf = 0; kf < KF; ++kf)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int co = lane & 31, kt = 8 * (lane >> 5) + e;
                if (kt >= KT) continue;
                const float x = w[((size_t)co * KF + kf) * KT + kt];
                const _Float16 h1 = (_Float16)x;
                const _Float16 h2 = (_Float16)((x - (float)h1) * kLoScale);
                const size_t base = (((size_t)kf * 2) * 64 + lane) * 8 + e;
                out[base] = c1_bits(h1); out[base + 512] = c1_bits(h2);
            }
    return out;
}

void launch_conv1_split(const ConvLaunch& c, const uint16_t* wp_sp, hipStream_t s) {
    Conv1Args a{c.x, wp_sp, c.bias, c.bn_a, c.bn_b, c.out_lens_dev, c.y, c.y_sp, c.B, c.fi, c.fo, c.ti, c.to, c.xs, c.ys};
    const dim3 grid(ceil_div(c.to, C1TT), ceil_div(c.fo, C1NF), c.B);
    if (c.y_sp) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_LDS);
        DSMI_LAUNCH(conv1_f16x3_kernel<true>, grid, dim3(C1NT), C1_LDS, s, c.ev, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1_LDS);
        DSMI_LAUNCH(conv1_f16x3_kernel<false>, grid, dim3(C1NT), C1_LDS, s, c.ev, a);
    }
}

}  // namespace dsmi
