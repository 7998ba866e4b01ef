// Conv2d (32 input channels) + bias + BatchNorm2d(eval) + Hardtanh(0,20) + time mask on the fp16
// MFMA with two-term split operands (x = hi + lo, the lo term unscaled, the weights packed times 2^6: three true-valued
// products into ONE fp32 accumulator -- the operand format of gemm.hip, round 3; better than fp32-MFMA accuracy,
// tools/exp/split_mfma_accuracy.hip).  Inputs are Hardtanh outputs in [0, 20] (their lo terms are at most 2^-7: normal fp16
// numbers or exact subnormals down to 2^-24) and the weights are range-checked at load time, so fp16's 65504 is never near.
//
// Replaces the 2nd and 3rd (Conv2d, BatchNorm2d, Hardtanh) triples of the reference's conv stack
// and their MaskConv zeroing (danspeech/deepspeech/model.py:65-81, 372-374, 389-391), i.e. the
// layers that hold 89 % / 97 % of the conv FLOPs.  conv.hip (fp32 MFMA) keeps the
// first layer (1 input channel) and remains the plain fp32 statement of all three.
//
// Implicit GEMM with the INPUT CHANNELS as the MFMA's K: one v_mfma_f32_16x16x32_f16 contracts all
// 32 input channels of one kernel tap (kf, kt) for 16 output channels x 16 output steps; A = weights
// (row = output channel, pre-split and packed on the host in lane order), B = input (col = output
// time step).  (Round 3: the 16 x 16 x 32 shape instead of 32 x 32 x 16 -- the same products at a
// higher sustained clock, see gemm.hip.)  The input arrives already split and channels-last,
// [b][f][plane 2][t][32 ci] fp16, written by the previous layer's epilogue, so that a lane's
// 8-channel B fragment is ONE aligned 16-byte LDS read and staging is a plain copy.  Workgroup =
// 4 waves = 4 consecutive output rows f x 64 output steps x 32 output channels; the input rows a
// kernel row kf needs (one per wave) live in LDS as two rings of four rows -- three of them are the
// rows kf - 2 used, so only ONE new row is staged per kernel row (in flight in registers during the
// MFMAs) --, 64 bytes per time step with the four 16-byte chunks of a step stored at
// chunk ^ ((step >> 1) & 3) (conflict-free for the ds_read_b128 lane groups at every tap offset),
// and the weight fragments of the tap after next are requested before the current one's MFMAs
// (they come from L2: ~600 cycles).
#include "common.h"
#include <type_traits>

namespace dsmi {

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
constexpr float kConvWScale = 64.f;    // weights are packed times 2^6 (exact), divided out in the epilogue
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int BNF = 4;                 // output rows per workgroup (one per wave)
constexpr int BTT = 64;                // output steps per workgroup (4 MFMA column tiles per wave)
constexpr int KT = 11, KF = 21, SF = 2, PF = 10, PT = 5, CI = 32;
constexpr int WIN = BTT + KT - 1;      // staged time steps per row: 74
constexpr int NPL = 2;                 // operand planes: hi, lo
constexpr int PITCH = 32;              // halfs per staged time step (32 ci = 64 B; chunks swizzled, see above)
constexpr int ROWPLANE = WIN * PITCH;  // halfs per (row, plane)
constexpr int RCHUNK = NPL * WIN * 4;               // 16-byte chunks of one staged input row (both planes): 592
constexpr int CPR = (RCHUNK + 255) / 256;           // chunks per thread and row: 3
constexpr int NSLOT = 2 * BNF;                      // staged rows held in LDS: two families of four (see the kernel)

struct ConvSplitArgs {
    const uint16_t* x_sp; const uint16_t* wp_sp; const float* bias; const float* bn_a; const float* bn_b;
    const int32_t* out_lens; float* y; uint16_t* y_sp;
    int B, fi, fo, ti, to, ys;
    int nco;        // 32-channel output tiles (1: 32 channels, 3: 96); blockIdx.z = b * nco + tile
};

// Shared epilogue piece: 4 consecutive channels of one (b, f, t) -> two 8-byte stores, channels-last split.
__device__ __forceinline__ void store_split4(uint16_t* y_sp, size_t bf_index, int t_stride, int t, int c0, const float (&v)[4]) {
    f16x4 h, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const _Float16 hi = (_Float16)v[c];
        h[c] = hi; l[c] = (_Float16)(v[c] - (float)hi);
    }
    _Float16* base = reinterpret_cast<_Float16*>(y_sp) + ((bf_index * NPL) * (size_t)t_stride + t) * 32 + c0;
    *reinterpret_cast<f16x4*>(base) = h;
    *reinterpret_cast<f16x4*>(base + (size_t)t_stride * 32) = l;
}

// One workgroup = one 32-channel output tile (the 96-channel third layer runs its tiles as separate workgroups); one
// accumulator per MFMA tile, 76 KB of LDS (eight staged input rows, see "staging" below), 136 registers: two workgroups per CU.
// XPIPE: a tap's eight x fragments are read from LDS during the MFMAs of the tap before it (a second set of 32 registers), one read
// per three MFMAs, instead of in front of their own MFMAs -- where each tap began with two exposed LDS round trips that only the
// SIMD's other wave could cover.
template <bool SPLIT_OUT, bool XPIPE = true>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(ConvSplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
    _Float16* Xs = reinterpret_cast<_Float16*>(csm);  // [2 families][4 slots][2 planes][WIN][PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * BTT, f0 = blockIdx.y * BNF;
    const int b = blockIdx.z / p.nco, ct = blockIdx.z - b * p.nco;
    const int f = f0 + wv;
    const int olen = p.out_lens[b];
    const int CO = 32 * p.nco;

    if (t0 >= olen) {     // fully masked tile: zeros in the consumer's format
        if (SPLIT_OUT) {
            for (int idx = tid; idx < BNF * NPL * BTT * 4; idx += 256) {
                const int part = idx & 3, tl = (idx >> 2) % BTT, pl = (idx / (4 * BTT)) % NPL, ff = idx / (4 * BTT * NPL);
                if (f0 + ff < p.fo && t0 + tl < p.to)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<_Float16*>(p.y_sp) + ((((size_t)b * p.fo + f0 + ff) * NPL + pl) * p.to + t0 + tl) * 32 + part * 8) = u32x4{0, 0, 0, 0};
            }
        } else {
            for (int idx = tid; idx < 32 * BNF * BTT; idx += 256) {
                const int tl = idx % BTT, ff = (idx / BTT) % BNF, co = ct * 32 + idx / (BTT * BNF);
                if (f0 + ff < p.fo && t0 + tl < p.to) p.y[(((size_t)b * CO + co) * p.fo + f0 + ff) * p.ys + t0 + tl] = 0.f;
            }
        }
        return;
    }

    f32x4 acc[2][4];                      // [16-channel tile][16-step tile]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[c][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging.  Kernel row kf needs input rows r0 + kf + 2 w for the four waves w (r0 = SF f0 - PF): rows of kf's parity.
    // Three of them are the rows kf - 2 used (wave w's row then is wave w - 1's now), so LDS holds two families of four rows (even
    // and odd kf) as rings, and every kf brings in ONE new row -- for kf + 2, into the slot of the row wave 0 is reading now: the
    // row is requested at the top of iteration kf, stored at the top of kf + 1 (everybody is past kf's barrier, and kf + 1 reads
    // the other family) and read from kf + 2 on.  27 rows staged per workgroup instead of 84, one barrier per kernel row.
    // Family p = kf & 1 holds rows r0 + p + 2 m in slot m & 3; wave w reads m = (kf >> 1) + w.
    const int r0 = SF * f0 - PF;
    auto load_row = [&](int fin, u32x4 (&dst)[CPR]) {          // chunk id -> (plane, step, 16-byte part); a plain copy
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
            const int c = tid + 256 * i;
            u32x4 v = {0, 0, 0, 0};
            if (c < RCHUNK) {
                const int part = c & 3, st = (c >> 2) % WIN, pl = (c >> 2) / WIN;
                const int tin = t0 - PT + st;
                if (fin >= 0 && fin < p.fi && tin >= 0 && tin < p.ti)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(p.x_sp) +
                                                        ((((size_t)b * p.fi + fin) * NPL + pl) * p.ti + tin) * 32 + part * 8);
            }
            dst[i] = v;
        }
    };
    auto store_row = [&](int slot8, const u32x4 (&src)[CPR]) {   // slot8 = family * 4 + slot
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
            const int c = tid + 256 * i;
            if (c < RCHUNK) {
                const int part = c & 3, st = (c >> 2) % WIN, pl = (c >> 2) / WIN;
                *reinterpret_cast<u32x4*>(Xs + (slot8 * NPL + pl) * ROWPLANE + st * PITCH + ((part ^ ((st >> 1) & 3)) * 8)) = src[i];
            }
        }
    };

    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.wp_sp) + lane;
    const int l15 = lane & 15, l4 = lane >> 4;
    // weight fragments of tap q = kf * KT + kt, two taps ahead of the MFMAs (they come from L2: a tap is 24 MFMAs = 384 cycles,
    // the other waves of the SIMD cover as much again); the ring's two slots alternate with q, and KT is odd, so the
    // kernel-row body exists in two versions (first tap in slot 0 / slot 1)
    f16x8 wq[2][2][NPL];                  // [slot][16-channel tile][plane]
    auto load_w = [&](int q, f16x8 (&dst)[2][NPL]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                dst[c][pl] = __builtin_bit_cast(f16x8, wbase[((((size_t)q * p.nco + ct) * 2 + c) * NPL + pl) * 64]);
    };
    constexpr int NQ = KF * KT;
    load_w(0, wq[0]);
    load_w(1, wq[1]);
    auto tap = [&](const _Float16* xrow, int kf, int kt, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int q = kf * KT + kt;
        f16x8 wf[2][NPL];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) wf[c][pl] = wq[slot][c][pl];
        load_w(min(q + 2, NQ - 1), wq[slot]);
        f16x8 xf[4][NPL];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int st = tt * 16 + kt + l15;          // staged step of this lane's column
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                xf[tt][pl] = *reinterpret_cast<const f16x8*>(xrow + pl * ROWPLANE + st * PITCH + ((l4 ^ ((st >> 1) & 3)) * 8));
        }
        // the three products of a pair run into one accumulator; plane pair by plane pair across the eight tiles
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    acc[c][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][pp == 0 ? 1 : 0], xf[tt][pp == 1 ? 1 : 0], acc[c][tt], 0, 0, 0);
    };
    // ---- the pipelined form of a tap
    auto read_x = [&](const _Float16* xrow, int kt, f16x8 (&xf)[4][NPL]) {
        const int st0 = kt + l15;                           // (16 tt does not move the swizzle: one base, immediates)
        const _Float16* base = xrow + st0 * PITCH + ((l4 ^ ((st0 >> 1) & 3)) * 8);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) xf[tt][pl] = *reinterpret_cast<const f16x8*>(base + pl * ROWPLANE + tt * 16 * PITCH);
    };
    auto tap_p = [&](const _Float16* xrow, int kf, int kt, auto slot_c, const f16x8 (&xf)[4][NPL], f16x8 (&xn)[4][NPL], bool more) {
        constexpr int slot = decltype(slot_c)::value;
        const int q = kf * KT + kt;
        f16x8 wf[2][NPL];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) wf[c][pl] = wq[slot][c][pl];
        load_w(min(q + 2, NQ - 1), wq[slot]);
        if (more) read_x(xrow, kt + 1, xn);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    acc[c][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c][pp == 0 ? 1 : 0], xf[tt][pp == 1 ? 1 : 0], acc[c][tt], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);     // three MFMAs
            }
        }
    };
    auto taps = [&](const _Float16* xrow, int kf, auto first_slot) {
        constexpr int S0 = decltype(first_slot)::value;
        if constexpr (XPIPE) {
            f16x8 xa[4][NPL], xb[4][NPL];
            read_x(xrow, 0, xa);
#pragma unroll 1
            for (int kt = 0; kt + 1 < KT; kt += 2) {
                tap_p(xrow, kf, kt, std::integral_constant<int, S0>{}, xa, xb, true);
                tap_p(xrow, kf, kt + 1, std::integral_constant<int, S0 ^ 1>{}, xb, xa, true);
            }
            tap_p(xrow, kf, KT - 1, std::integral_constant<int, S0>{}, xa, xb, false);
        } else {
#pragma unroll 1
            for (int kt = 0; kt + 1 < KT; kt += 2) {
                tap(xrow, kf, kt, std::integral_constant<int, S0>{});
                tap(xrow, kf, kt + 1, std::integral_constant<int, S0 ^ 1>{});
            }
            tap(xrow, kf, KT - 1, std::integral_constant<int, S0>{});
        }
    };

    // KT is odd, so the first tap's ring slot alternates with kf: the kernel rows are walked in pairs (straight-line code for an
    // even and an odd row; a branch between two versions of the body costs 80 registers at its merge)
    u32x4 stg[CPR];                       // the row in flight
    auto kernel_row = [&](int kf, auto first_slot) {
        const int fam = kf & 1, m0 = kf >> 1;
        if (kf >= 1 && kf + 1 < KF) store_row(((kf - 1) & 1) * 4 + (((kf - 1) >> 1) & 3), stg);     // requested during kf - 1, for kf + 1
        if (kf + 2 < KF) load_row(r0 + kf + 8, stg);
        if (f < p.fo) taps(Xs + ((fam * 4 + ((m0 + wv) & 3)) * NPL) * ROWPLANE, kf, first_slot);
        __syncthreads();
    };
    // prologue: the eight rows of kernel rows 0 and 1, four at a time
    {
        u32x4 pre[4][CPR];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_row(r0 + half * 4 + j, pre[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = half * 4 + j;           // row r0 + row: family row & 1, m = row >> 1
                store_row((row & 1) * 4 + (row >> 1), pre[j]);
            }
        }
    }
    __syncthreads();
    static_assert(KF % 2 == 1, "the last kernel row is an even one");
#pragma unroll 1
    for (int kf = 0; kf + 1 < KF; kf += 2) {
        kernel_row(kf, std::integral_constant<int, 0>{});
        kernel_row(kf + 1, std::integral_constant<int, 1>{});
    }
    kernel_row(KF - 1, std::integral_constant<int, 0>{});

    if (f >= p.fo) return;
    // ---- epilogue: D[i][j] of a 16 x 16 tile: j = lane & 15 = time, i = 4 (lane >> 4) + register = output channel
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int t = t0 + tt * 16 + l15;
        if (t >= p.to) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = ct * 32 + c * 16 + 4 * l4 + q;
                float x = (acc[c][tt][q] * (1.f / kConvWScale) + p.bias[co]) * p.bn_a[co] + p.bn_b[co];
                x = fminf(fmaxf(x, 0.f), 20.f);
                v[q] = t < olen ? x : 0.f;
            }
            if (SPLIT_OUT) {
                store_split4(p.y_sp, (size_t)b * p.fo + f, p.to, t, c * 16 + 4 * l4, v);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p.y[(((size_t)b * CO + ct * 32 + c * 16 + 4 * l4 + q) * p.fo + f) * p.ys + t] = v[q];
            }
        }
    }
}

inline uint16_t c_f16_bits(_Float16 h) {
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

}  // namespace

// w [co][32][21][11] fp32 -> [kf][kt][32-channel tile][16-channel half][plane][lane][8] fp16 terms (hi, lo) of w * 2^6; lane l
// (row = l & 15 of the 16-channel half, k chunk l >> 4) element e holds input channel 8 * (l >> 4) + e.
std::vector<uint16_t> pack_conv_w_split(const float* w, int co_total) {
    const int nco = co_total / 32;
    std::vector<uint16_t> out((size_t)KF * KT * nco * 2 * NPL * 64 * 8, 0);
    for (int kf = 0; kf < KF; ++kf)
        for (int kt = 0; kt < KT; ++kt)
            for (int ct = 0; ct < nco; ++ct)
                for (int c = 0; c < 2; ++c)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = ct * 32 + c * 16 + (lane & 15), ci = 8 * (lane >> 4) + e;
                            const float x = w[(((size_t)co * CI + ci) * KF + kf) * KT + kt] * kConvWScale;
                            const _Float16 h1 = (_Float16)x;
                            const _Float16 h2 = (_Float16)(x - (float)h1);
                            const size_t base = ((((((size_t)kf * KT + kt) * nco + ct) * 2 + c) * NPL) * 64 + lane) * 8 + e;
                            out[base] = c_f16_bits(h1); out[base + 512] = c_f16_bits(h2);
                        }
    return out;
}

void launch_conv_split(const ConvSplitLaunch& c, hipStream_t s) {
    ConvSplitArgs a{c.x_sp, c.wp_sp, c.bias, c.bn_a, c.bn_b, c.out_lens_dev, c.y, c.y_sp, c.B, c.fi, c.fo, c.ti, c.to, c.ys, c.co / 32};
    const dim3 grid(ceil_div(c.to, BTT), ceil_div(c.fo, BNF), c.B * a.nco);
    const size_t lds = (size_t)NSLOT * NPL * ROWPLANE * 2;   // 75,776 B: two workgroups per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_f16x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_f16x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef DSMI_EXPERIMENTS       // the form that reads a tap's fragments in front of its own MFMAs (rounds 3-5), for A/B runs
    static const bool plain = [] { const char* e = exp_env("DSMI_DEBUG_CONV_XPIPE"); return e && std::atoi(e) == 0; }();
    if (plain) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_f16x3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_f16x3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (c.y_sp) DSMI_LAUNCH((conv_f16x3_kernel<true, false>), grid, dim3(256), lds, s, c.ev, a);
        else DSMI_LAUNCH((conv_f16x3_kernel<false, false>), grid, dim3(256), lds, s, c.ev, a);
        return;
    }
#endif
    if (c.y_sp) DSMI_LAUNCH((conv_f16x3_kernel<true>), grid, dim3(256), lds, s, c.ev, a);
    else DSMI_LAUNCH((conv_f16x3_kernel<false>), grid, dim3(256), lds, s, c.ev, a);
}

}  // namespace dsmi
