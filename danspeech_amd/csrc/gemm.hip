// fp32 input-projection GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   C[m][n] = sum_k A[m][k] * W[n][k] + bias[n]
//
// Replaces the `linear(input, w_ih, b_ih)` half of torch.nn.{GRU,LSTM,RNN} that the
// reference runs inside BatchRNN.forward (danspeech/deepspeech/model.py:114-122) for all
// time steps at once, with two producer fusions:
//   GEMM_A_SUM_BN : A = (h_fwd + h_bwd) * bn_a + bn_b    -- the direction sum of the
//                   previous layer (model.py:121) and SequenceWise(BatchNorm1d) (model.py:115-116)
//   GEMM_A_CONV   : A[(b,t)][c*F+f] = conv_out[b][c][f][t] -- the view/transpose/contiguous of
//                   model.py:501-503 folded into the operand load.
//
// Tile 128x128x32, 256 threads = 4 waves in 2x2, each wave 2x2 MFMA tiles of 32x32.
// Both operands are K-contiguous, so a lane's float4 (k = 8q+4h .. +3, h = lane>>5) feeds
// four consecutive MFMAs: the k -> (instruction, lane-half) assignment is a free
// permutation as long as A and W use the same one.
// LDS rows are padded to 36 floats so that ds_read_b128 of 16 consecutive rows covers
// all 64 banks (stride 36 dwords = 4 mod 64 * 9).
#include "common.h"
#include <cstdlib>

namespace dsmi {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;      // padded row length (floats) of a [rows][BK] LDS tile
constexpr int LDC = BM + 4;      // padded row length of the k-major A tile (GEMM_A_CONV)

struct GemmArgs {
    const float* a; const float* a2; const float* alpha; const float* beta;
    const float* w; const float* bias; float* c;
    int M, N, K, lda, ldw, ldc, B, T, ys, tiles_per_b;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_TILE = (MODE == GEMM_A_CONV) ? BK * LDC : BM * LDT;
    constexpr int W_TILE = BN * LDT;
    constexpr int STAGE = A_TILE + W_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, hk = lane >> 5;
    const int n0 = blockIdx.x * BN;
    const int mt = blockIdx.y;

    // row-tile origin
    int m0 = mt * BM, bb = 0, t0 = 0;
    if (MODE == GEMM_A_CONV) { bb = mt / p.tiles_per_b; t0 = (mt % p.tiles_per_b) * BM; }

    f32x4 ra[4], rw[4];   // register-staged next tile

    auto load_global = [&](int k0) {
        if (MODE == GEMM_A_CONV) {
            // 32 k-rows x 128 t ; thread: k = tid/32 + 8*pass, t = 4*(tid%32)
            const int tq = (tid & 31) * 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int k = k0 + (tid >> 5) + 8 * ps;
                const int t = t0 + tq;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K && t + 3 < p.ys)
                    v = *reinterpret_cast<const f32x4*>(p.a + ((size_t)bb * p.K + k) * p.ys + t);
                ra[ps] = v;
            }
        } else {
            const int kc = k0 + (tid & 7) * 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = m0 + (tid >> 3) + 32 * ps;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M && kc < p.K) {
                    v = *reinterpret_cast<const f32x4*>(p.a + (size_t)m * p.lda + kc);
                    if (MODE == GEMM_A_SUM_BN) {
                        if (p.a2) v += *reinterpret_cast<const f32x4*>(p.a2 + (size_t)m * p.lda + kc);
                        const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + kc);
                        const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + kc);
                        v = v * al + be;
                    }
                }
                ra[ps] = v;
            }
        }
        const int kc = k0 + (tid & 7) * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int n = n0 + (tid >> 3) + 32 * ps;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N && kc < p.K) v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.ldw + kc);
            rw[ps] = v;
        }
    };

    auto store_lds = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Ws = As + A_TILE;
        if (MODE == GEMM_A_CONV) {
            const int tq = (tid & 31) * 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                *reinterpret_cast<f32x4*>(As + ((tid >> 5) + 8 * ps) * LDC + tq) = ra[ps];
        } else {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * ps) * LDT + (tid & 7) * 4) = ra[ps];
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
            *reinterpret_cast<f32x4*>(Ws + ((tid >> 3) + 32 * ps) * LDT + (tid & 7) * 4) = rw[ps];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_global((kt + 1) * BK);
        const float* As = smem + buf * STAGE;
        const float* Ws = As + A_TILE;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[2], wf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wr * 64 + mi * 32 + li;
                if (MODE == GEMM_A_CONV) {
                    const float* s = As + (8 * q + 4 * hk) * LDC + row;
                    af[mi][0] = s[0]; af[mi][1] = s[LDC]; af[mi][2] = s[2 * LDC]; af[mi][3] = s[3 * LDC];
                } else {
                    af[mi] = *reinterpret_cast<const f32x4*>(As + row * LDT + 8 * q + 4 * hk);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                wf[ni] = *reinterpret_cast<const f32x4*>(Ws + (wc * 64 + ni * 32 + li) * LDT + 8 * q + 4 * hk);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][c], wf[ni][c], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[i][j] -> col j = lane&31 (n), row i = (r&3) + 8*(r>>2) + 4*hk (m)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wc * 64 + ni * 32 + li;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk;
                size_t crow;
                if (MODE == GEMM_A_CONV) {
                    const int t = t0 + i;
                    if (t >= p.T) continue;
                    crow = (size_t)t * p.B + bb;
                } else {
                    if (m0 + i >= p.M) continue;
                    crow = (size_t)(m0 + i);
                }
                p.c[crow * p.ldc + n] = acc[mi][ni][r] + bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-grade GEMM on the fp16 MFMA: both operands are split into two fp16 terms,
// x = hi + lo * 2^-11 (hi = fp16(x), lo = fp16((x - hi) * 2^11), 22 mantissa bits; lo is stored
// scaled so that it stays a normal number), and three products are formed per fragment pair:
// hi.hi into one fp32 accumulator, hi.lo + lo.hi into a second one that is folded in with 2^-11
// in the epilogue (lo.lo < 2^-22 is dropped).  The fp16 MFMA runs at 16x the fp32 MFMA
// rate, so the product costs 3/16 of the fp32 one; measured error vs fp64 at K = 800: 2.6e-7
// against 1.0e-6 for the fp32 MFMA chain (tools/exp/split_mfma_accuracy.hip).  Operand ranges must stay
// below fp16's 65504: the caller checks weights and BatchNorm bounds at load time (api.hip).
// W is split and tiled once on the host (pack_gemm_w_split: [n-tile][k-tile][plane][128][32] fp16, so
// the operand loads are fully coalesced 16-byte copies); the activations are split ONCE per GEMM
// by split_a_kernel, fused with the same producer transforms as the fp32 kernel (direction sum +
// BatchNorm1d, conv transpose), into the same tiled form.  The kernel itself: see gemm_f16x3_kernel below.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int XT_STRIDE = 129;         // f32 transpose tile [32 k][129] (GEMM_A_CONV)
constexpr float kGemmWScale = 64.f;      // see GemmSplitArgs

// The split-fp16 GEMM's operand format (round 3): x = hi + lo with hi = fp16(x), lo = fp16(x - hi), the lo term UNSCALED, so
// that all three products of a pair -- hi.hi, hi.lo, lo.hi -- are true-valued and can run into ONE fp32 accumulator.  An
// unscaled lo term is a normal fp16 number only for |x| >= 2^-3 (its 11 bits then complete the 22 of the pair); below that it
// keeps an absolute precision of 2^-25, which is what a dot product needs.  The weights are small (|w| ~ K^-1/2), so they are
// multiplied by 2^6 when they are packed (exact; |w| < 937 is checked at load time) and the result is divided in the epilogue.
struct GemmSplitArgs {
    const float* a; const float* a2; const float* alpha; const float* beta;
    const uint16_t* w_sp; const float* bias; float* c;
    int M, N, K, lda, ldc, B, T, ys, tiles_per_b, ktiles;
    int ntiles, mtiles;       // output tiles (gemm kernel's XCD-aware tile order)
    int pn;                   // n-tiles of a W panel: what an XCD keeps in its L2 while it walks the m-tiles (gemm_panel_width)
    int pn2;                  // the same in PAIRS of n-tiles (the 128 x 256 form)
};

// Pass 1: form the A operand ONCE (producer transforms + two-term split) as tiled fp16 planes
// [m-tile][k-tile][plane][128][32]; every one of the N/128 column tiles then reads it as is.
template <int MODE>
__global__ __launch_bounds__(256) void split_a_kernel(GemmSplitArgs p, uint16_t* a_sp) {
    __shared__ float Tf[MODE == GEMM_A_CONV ? 32 * XT_STRIDE : 1];
    const int tid = threadIdx.x;
    const int kt = blockIdx.x, mt = blockIdx.y;
    const int k0 = kt * BK;
    _Float16* tile = reinterpret_cast<_Float16*>(a_sp) + ((size_t)mt * p.ktiles + kt) * (2 * 4096);
    f32x4 v[4];
    if (MODE == GEMM_A_CONV) {
        const int bb = mt / p.tiles_per_b, t0 = (mt % p.tiles_per_b) * BM;
        const int tq = (tid & 31) * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int k = k0 + (tid >> 5) + 8 * ps, t = t0 + tq;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (k < p.K && t + 3 < p.ys) x = *reinterpret_cast<const f32x4*>(p.a + ((size_t)bb * p.K + k) * p.ys + t);
            float* d = Tf + ((tid >> 5) + 8 * ps) * XT_STRIDE + tq;
            d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; d[3] = x[3];
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = (tid >> 3) + 32 * ps, kc = (tid & 7) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[ps][c] = Tf[(kc + c) * XT_STRIDE + row];
        }
    } else {
        const int m0 = mt * BM, kc = k0 + (tid & 7) * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = m0 + (tid >> 3) + 32 * ps;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (m < p.M && kc < p.K) {
                x = *reinterpret_cast<const f32x4*>(p.a + (size_t)m * p.lda + kc);
                if (MODE == GEMM_A_SUM_BN) {
                    if (p.a2) x += *reinterpret_cast<const f32x4*>(p.a2 + (size_t)m * p.lda + kc);
                    x = x * *reinterpret_cast<const f32x4*>(p.alpha + kc) + *reinterpret_cast<const f32x4*>(p.beta + kc);
                }
            }
            v[ps] = x;
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = (tid >> 3) + 32 * ps, kc = (tid & 7) * 4;
        f16x4 h, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const _Float16 hi = (_Float16)v[ps][c];
            h[c] = hi; l[c] = (_Float16)(v[ps][c] - (float)hi);      // UNSCALED: see GemmSplitArgs
        }
        *reinterpret_cast<f16x4*>(tile + 0 * 4096 + row * 32 + kc) = h;
        *reinterpret_cast<f16x4*>(tile + 1 * 4096 + row * 32 + kc) = l;
    }
}

// Pass 2: C = A2 . W2^T + bias on v_mfma_f32_16x16x32_f16, three products per (A, W) fragment pair into one accumulator.
// Tile 128 x 128, four waves as 2 x 2, each wave 4 x 4 MFMA tiles of 16 x 16; ONE LDS stage of a 32-deep k-tile (32 KB) filled
// by direct global -> LDS loads, and the registers as the second buffer: read all sixteen fragments, barrier, request the next
// tile into the same stage, multiply (48 MFMAs), wait, barrier.  Three workgroups per CU cover each other's waits.
// Why this shape (round 3, profiles/r03_gemm_bounds.txt): the chip holds its clock at 1.9 GHz under this kernel (MI355X_MICROARCH.md
// "DVFS give-back"), so every tile shape, stage count and prefetch depth tried on v_mfma_f32_32x32x16_f16 -- 128 x 128 with
// two to four stages, 256 x 128, 256 x 256 with register double buffering -- cost the same 0.38-0.40 ms; the 16 x 16 x 32 MFMA
// does the same products at a higher sustained clock: 0.340 against 0.398 ms for this very loop on 32 x 32 x 16, 8.15 against
// 8.45 ms per bench step (tools/exp/retired/gemm_32x32x16_forms.hip.inc).
// STAGES = 2 (round 5): two LDS stages (64 KB, two workgroups per CU); the k-tile TWO ahead is requested into the stage whose
// fragments everybody has just taken into registers, so a request has two k-tiles' MFMAs (1500 cycles) to land instead of one --
// the one-stage form waits at the end of every k-tile for a request issued 48 MFMAs earlier (MFMA busy 0.6).  A/B:
// DSMI_DEBUG_GEMM_STAGES, tools/exp/gemm_stages_time.py.
template <bool CONV_ROWS, int STAGES = 1>
__global__ __launch_bounds__(256, STAGES == 1 ? 3 : 2) void gemm_f16x3_kernel(GemmSplitArgs p, const uint16_t* a_sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    int nt, mt;
    {
        const int ntiles = p.ntiles, mtiles = p.mtiles, total = ntiles * mtiles;
        const int share = (total + 7) / 8;
        const int idx = (blockIdx.x & 7) * share + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= share || idx >= total) return;
        const int PN = p.pn;
        const int panel = idx / (PN * mtiles), rem = idx - panel * (PN * mtiles);
        const int pw = min(PN, ntiles - panel * PN);
        mt = rem / pw; nt = panel * PN + (rem - mt * pw);
    }
    // rows of 64 bytes (32 k); a row's four 16-byte chunks are stored at chunk ^ ((row >> 1) & 3) ^ ((row >> 2) & 3): conflict-free
    // for the ds_read_b128 lane groups of both MFMA shapes
    const int drow = lane >> 2;
    const int doff = (16 * wid + drow) * 64 + (((lane & 3) ^ ((drow >> 1) & 3) ^ ((drow >> 2) & 3)) * 16);
    const unsigned char* asrc = reinterpret_cast<const unsigned char*>(a_sp) + (size_t)mt * p.ktiles * 16384 + doff;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w_sp) + (size_t)nt * p.ktiles * 16384 + doff;
    unsigned char* lds = smem3;                 // [stage][A hi, A lo, W hi, W lo][128 rows][64 B]
    auto dma = [&](int kt) {
        unsigned char* st = lds + (STAGES == 2 ? (kt & 1) * 32768 : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)             // i: operand i >> 2, plane (i >> 1) & 1, 64-row half i & 1; this wave's 16 rows of it
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(((i >> 2) ? wsrc : asrc) + (size_t)kt * 16384 +
                                                                                               ((i >> 1) & 1) * 8192 + (i & 1) * 4096),
                                             (__attribute__((address_space(3))) void*)(st + (i >> 1) * 8192 + (i & 1) * 4096 + wid * 1024), 16, 0, 0);
    };
    int stage_off = 0;                          // byte offset of the stage the fragments are read from
    auto frag = [&](int plane_base, int row, int chunk) {
        return *reinterpret_cast<const f16x8*>(lds + stage_off + plane_base + row * 64 + ((chunk ^ ((row >> 1) & 3) ^ ((row >> 2) & 3)) * 16));
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    dma(0);
    if (STAGES == 2 && p.ktiles > 1) { dma(1); asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); }     // (loads complete in issue order: all but the newest eight)
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < p.ktiles; ++kt) {
        if (STAGES == 2) stage_off = (kt & 1) * 32768;
        f16x8 af[4][2], wf[4][2];        // [tile][plane]: lane l holds row l & 15 of the tile, k = 8 (l >> 4) ..
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[q][pl] = frag(pl * 8192, wr * 64 + q * 16 + (lane & 15), lane >> 4);
                wf[q][pl] = frag(16384 + pl * 8192, wc * 64 + q * 16 + (lane & 15), lane >> 4);
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // everybody holds its fragments: the stage is free
        if (kt + STAGES < p.ktiles) dma(kt + STAGES);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)          // hi.hi, hi.lo, lo.hi: plane pair by plane pair across the sixteen tiles
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][pp == 2 ? 1 : 0], wf[ni][pp == 1 ? 1 : 0], acc[mi][ni], 0, 0, 0);
        // the next tile has landed, for everybody (two stages: all but the eight requests just issued -- unless none were)
        if (STAGES == 2 && kt + 2 < p.ktiles) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue.  D[i][j] of a 16 x 16 tile: j = lane & 15 (column n), i = 4 (lane >> 4) + register: stored as it stands, a wave
    // instruction writes sixteen 64-byte pieces.  Each wave turns its tile through its quarter of the (now idle) LDS stage, sixteen
    // rows at a time, so that a lane holds four consecutive columns of a row and an instruction writes four 256-byte row pieces.
    const int n0 = nt * BN;
    int m0 = mt * BM, bb = 0, t0 = 0;
    if (CONV_ROWS) { bb = mt / p.tiles_per_b; t0 = (mt % p.tiles_per_b) * BM; }
    constexpr int TP = 68;                      // words per row of the turn buffer: rows 4 apart land 16 banks apart
    float* turn = reinterpret_cast<float*>(lds) + wid * (16 * TP);
    const int cn = (lane & 15) * 4, cr = lane >> 4;             // this lane's four columns and its row (+ 4 j) when writing out
    const int ncol = n0 + wc * 64 + cn;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && ncol + 3 < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + ncol);
    else if (p.bias)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = ncol + q < p.N ? p.bias[ncol + q] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) turn[((lane >> 4) * 4 + r) * TP + ni * 16 + (lane & 15)] = acc[mi][ni][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (wave-private region: no barrier)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wr * 64 + mi * 16 + cr + 4 * j;           // row of the workgroup tile
            f32x4 v = *reinterpret_cast<const f32x4*>(turn + (cr + 4 * j) * TP + cn);
            v = v * (1.f / kGemmWScale) + bv;
            size_t crow;
            bool ok;
            if (CONV_ROWS) { ok = t0 + i < p.T; crow = (size_t)(t0 + i) * p.B + bb; }
            else { ok = m0 + i < p.M; crow = (size_t)(m0 + i); }
            if (!ok) continue;
            float* dst = p.c + crow * p.ldc + ncol;
            if (ncol + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
            else
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ncol + q < p.N) dst[q] = v[q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next sixteen rows overwrite them
    }
}

// The same GEMM on a 128 x 256 workgroup tile (round 5): one A tile against TWO adjacent n-tiles of W.  What bounds the 128 x 128
// form is not the requests' latency (two LDS stages: -4 %, DSMI_DEBUG_GEMM_STAGES) but the fragments' way from LDS to the
// registers: 16 ds_read_b128 per wave and 48 MFMAs = 64 KB per workgroup and k-tile against the LDS's 128 B per clock -- 512 of the
// 768 cycles the MFMAs take.  Here a wave's four A fragments (per plane) serve eight W tiles instead of four: 24 reads per 96 MFMAs
// (-25 % per MFMA), 48 KB requested per k-tile for twice the products (-25 %), half the barriers per MFMA.  One LDS stage (48 KB)
// and the registers as the second buffer, as before; 224 registers of fragments and accumulators: two workgroups per CU.
template <bool CONV_ROWS, int GAP = 6>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_wide_kernel(GemmSplitArgs p, const uint16_t* a_sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    int nt2, mt;
    const int ntiles2 = (p.ntiles + 1) >> 1;
    {
        const int mtiles = p.mtiles, total = ntiles2 * mtiles;
        const int share = (total + 7) / 8;
        const int idx = (blockIdx.x & 7) * share + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= share || idx >= total) return;
        const int PN = p.pn2;
        const int panel = idx / (PN * mtiles), rem = idx - panel * (PN * mtiles);
        const int pw = min(PN, ntiles2 - panel * PN);
        mt = rem / pw; nt2 = panel * PN + (rem - mt * pw);
    }
    const int nta = 2 * nt2, ntb = min(2 * nt2 + 1, p.ntiles - 1);      // (an odd number of n-tiles: the last pair multiplies its tile twice, stores it once)
    const int drow = lane >> 2;
    const int doff = (16 * wid + drow) * 64 + (((lane & 3) ^ ((drow >> 1) & 3) ^ ((drow >> 2) & 3)) * 16);
    const unsigned char* asrc = reinterpret_cast<const unsigned char*>(a_sp) + (size_t)mt * p.ktiles * 16384 + doff;
    const unsigned char* wsa = reinterpret_cast<const unsigned char*>(p.w_sp) + (size_t)nta * p.ktiles * 16384 + doff;
    const unsigned char* wsb = reinterpret_cast<const unsigned char*>(p.w_sp) + (size_t)ntb * p.ktiles * 16384 + doff;
    unsigned char* lds = smem3;                 // [A, W tile a, W tile b][hi, lo][128 rows][64 B]
    // request i of a k-tile (12 per wave): operand i >> 2 (A, Wa, Wb), plane (i >> 1) & 1, 64-row half i & 1; this wave's 16 rows of it
    auto dma_piece = [&](int kt, int i) {
        const unsigned char* src = (i >> 2) == 0 ? asrc : ((i >> 2) == 1 ? wsa : wsb);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)kt * 16384 + ((i >> 1) & 1) * 8192 + (i & 1) * 4096),
                                         (__attribute__((address_space(3))) void*)(lds + (i >> 1) * 8192 + (i & 1) * 4096 + wid * 1024), 16, 0, 0);
    };
    auto dma = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 12; ++i) dma_piece(kt, i);
    };
    auto frag = [&](int plane_base, int row, int chunk) {
        return *reinterpret_cast<const f16x8*>(lds + plane_base + row * 64 + ((chunk ^ ((row >> 1) & 3) ^ ((row >> 2) & 3)) * 16));
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    dma(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < p.ktiles; ++kt) {
        f16x8 af[4][2], wf[8][2];        // [tile][plane]: lane l holds row l & 15 of the tile, k = 8 (l >> 4) ..; W tiles 0-3 of n-tile a, 4-7 of n-tile b
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[q][pl] = frag(pl * 8192, wr * 64 + q * 16 + (lane & 15), lane >> 4);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wf[q][pl] = frag(16384 + (q >> 2) * 16384 + pl * 8192, wc * 64 + (q & 3) * 16 + (lane & 15), lane >> 4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // everybody holds its fragments: the stage is free
        // The next k-tile's twelve requests go out BETWEEN the MFMAs, one per GAP of them from the first: an LDS-DMA piece holds the wave's
        // issue port for 60 cycles and more -- twelve of them in front of the MFMAs are 700 cycles in which this wave's matrix pipe stands
        // (the 96 MFMAs take 1536) --, and the last request has the remaining MFMAs to land under.  The last k-tile requests itself again
        // (into the stage nobody reads any more): no branch in the block.  Per layer GEMM, one box (profiles/r05_gemm_forms.txt; commit
        // 292020c holds the sweep build): all twelve in front 613 us; one per 3 MFMAs 637, 4: 626, 5: 612, 6: 594, 7: 587, 8: 580 - 596.
        const int ktn = kt + 1 < p.ktiles ? kt + 1 : kt;
#pragma unroll
        for (int e = 0; e < 96; ++e) {          // hi.hi, hi.lo, lo.hi: plane pair by plane pair across the thirty-two tiles
            const int pp = e >> 5, mi = (e >> 3) & 3, ni = e & 7;
            if (e < 12 * GAP && e % GAP == 0) dma_piece(ktn, e / GAP);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi][pp == 2 ? 1 : 0], wf[ni][pp == 1 ? 1 : 0], acc[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 12; ++g) {          // ... and the scheduler is told to keep it that way
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);     // one vector-memory instruction
            __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);   // GAP MFMAs
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 96 - 12 * GAP, 0);
        // (the MFMAs behind the last request are NOT pinned in front of the wait by a sched_barrier: the compiler then moves them behind
        // the barrier, where they run beside the next k-tile's fragment reads -- 558 against 573 us per layer GEMM with them pinned,
        // on the box that gave both)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");           // the next tile has landed, for everybody
    }

    // ---- epilogue: as the 128 x 128 form's, once per n-tile of the pair
    int m0 = mt * BM, bb = 0, t0 = 0;
    if (CONV_ROWS) { bb = mt / p.tiles_per_b; t0 = (mt % p.tiles_per_b) * BM; }
    constexpr int TP = 68;
    float* turn = reinterpret_cast<float*>(lds) + wid * (16 * TP);
    const int cn = (lane & 15) * 4, cr = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (2 * nt2 + h >= p.ntiles) break;
        const int n0 = (2 * nt2 + h) * BN;
        const int ncol = n0 + wc * 64 + cn;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && ncol + 3 < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + ncol);
        else if (p.bias)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = ncol + q < p.N ? p.bias[ncol + q] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) turn[((lane >> 4) * 4 + r) * TP + ni * 16 + (lane & 15)] = acc[mi][4 * h + ni][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (wave-private region: no barrier)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = wr * 64 + mi * 16 + cr + 4 * j;           // row of the workgroup tile
                f32x4 v = *reinterpret_cast<const f32x4*>(turn + (cr + 4 * j) * TP + cn);
                v = v * (1.f / kGemmWScale) + bv;
                size_t crow;
                bool ok;
                if (CONV_ROWS) { ok = t0 + i < p.T; crow = (size_t)(t0 + i) * p.B + bb; }
                else { ok = m0 + i < p.M; crow = (size_t)(m0 + i); }
                if (!ok) continue;
                float* dst = p.c + crow * p.ldc + ncol;
                if (ncol + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
                else
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (ncol + q < p.N) dst[q] = v[q];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next sixteen rows overwrite them
        }
    }
}

static inline uint16_t g_f16_bits(_Float16 h) {
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// W [N][ldw] fp32 (first K columns valid) -> [n-tile][k-tile][plane][128][32] fp16 terms (hi, lo) of W * kGemmWScale.
std::vector<uint16_t> pack_gemm_w_split(const float* w, int N, int K, int ldw) {
    const int ntl = ceil_div(N, BN), ktl = ceil_div(K, BK);
    std::vector<uint16_t> out((size_t)ntl * ktl * 2 * 128 * 32, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float x = w[(size_t)n * ldw + k] * kGemmWScale;
            const _Float16 h1 = (_Float16)x;
            const _Float16 h2 = (_Float16)(x - (float)h1);
            const size_t base = (((size_t)(n / BN) * ktl + k / BK) * 2) * 4096 + (size_t)(n % BN) * 32 + (k % BK);
            out[base] = g_f16_bits(h1); out[base + 4096] = g_f16_bits(h2);
        }
    return out;
}

// The W panel an XCD walks the m-tiles with should stay in that XCD's 4-MiB L2 beside the A tiles streaming through it (a k-tile of
// one n-tile is 16 KiB: two planes).  Measured by FETCH_SIZE on 64-clip launches (tools/exp/gemm_panel_fetch.sh,
// profiles/r04_gemm_panel_fetch.txt): eight n-tiles 1.46 GB (K = 800) / 3.75 GB (K = 1312), five 1.29 / 2.81, four 1.35 / 2.78, three
// 1.58 / 3.16 -- five; the time does not move, alone or in the pipeline.  DSMI_DEBUG_GEMM_PN overrides.
static int gemm_panel_width(int ktiles) {
    static const int forced = [] { const char* e = exp_env("DSMI_DEBUG_GEMM_PN"); return e ? std::atoi(e) : 0; }();
    (void)ktiles;
    return forced > 0 ? forced : 5;
}

void launch_gemm(const GemmLaunch& g, hipStream_t s) {
    if (g.w_sp && g.a_sp) {
        GemmSplitArgs a;
        a.a = g.a; a.a2 = g.a2; a.alpha = g.alpha; a.beta = g.beta; a.w_sp = g.w_sp; a.bias = g.bias; a.c = g.c;
        a.M = g.M; a.N = g.N; a.K = g.K; a.lda = g.lda; a.ldc = g.ldc; a.B = g.B; a.T = g.T; a.ys = g.ys;
        a.tiles_per_b = 0; a.ktiles = ceil_div(g.K, BK);
        int mtiles3;
        if (g.mode == GEMM_A_CONV) { a.tiles_per_b = ceil_div(g.T, BM); mtiles3 = a.tiles_per_b * g.B; }
        else mtiles3 = ceil_div(g.M, BM);
        const dim3 sgrid(a.ktiles, mtiles3);
        switch (g.mode) {
            case GEMM_A_ROWMAJOR: hipLaunchKernelGGL(split_a_kernel<GEMM_A_ROWMAJOR>, sgrid, dim3(256), 0, s, a, g.a_sp); break;
            case GEMM_A_SUM_BN: hipLaunchKernelGGL(split_a_kernel<GEMM_A_SUM_BN>, sgrid, dim3(256), 0, s, a, g.a_sp); break;
            default: hipLaunchKernelGGL(split_a_kernel<GEMM_A_CONV>, sgrid, dim3(256), 0, s, a, g.a_sp); break;
        }
        a.ntiles = ceil_div(g.N, BN); a.mtiles = mtiles3;
        a.pn = gemm_panel_width(a.ktiles);
        const dim3 grid3(8 * ceil_div(a.ntiles * a.mtiles, 8));
        // the 128 x 256 form (default); DSMI_DEBUG_GEMM_WIDE=0: the 128 x 128 form (A/B runs)
        static const int wide = [] { const char* e = exp_env("DSMI_DEBUG_GEMM_WIDE"); return e ? std::atoi(e) : 1; }();
        // pairs of n-tiles per W panel: three (fetch per 64-clip launch, tools/exp/gemm_wide_fetch.sh: two 1.21 / 2.01 GB (K = 800 / 1312),
        // three 1.03 / 1.99, four 0.91 / 1.84 -- but 637 against 613 us alone; the 128 x 128 form at its best width, five n-tiles:
        // 1.29 / 2.90 GB, 640 us); DSMI_DEBUG_GEMM_PN (n-tiles) overrides
        static const bool pn_forced = exp_env("DSMI_DEBUG_GEMM_PN") != nullptr;
        a.pn2 = pn_forced ? std::max(1, (a.pn + 1) / 2) : 3;
#ifndef DSMI_EXPERIMENTS
        (void)wide; (void)grid3;
        {
#else
        if (wide) {
#endif
            const int ntiles2 = (a.ntiles + 1) / 2;
            const dim3 gridw(8 * ceil_div(ntiles2 * a.mtiles, 8));
            const size_t ldsw = 49152;                  // one stage: A and two W tiles, two 8-KiB planes each
            if (g.mode == GEMM_A_CONV) DSMI_LAUNCH(gemm_f16x3_wide_kernel<true>, gridw, dim3(256), ldsw, s, g.ev, a, (const uint16_t*)g.a_sp);
            else DSMI_LAUNCH(gemm_f16x3_wide_kernel<false>, gridw, dim3(256), ldsw, s, g.ev, a, (const uint16_t*)g.a_sp);
            return;
        }
#ifdef DSMI_EXPERIMENTS       // the 128 x 128 forms: rounds 3-4, kept for A/B runs
        static const int stages = [] { const char* e = exp_env("DSMI_DEBUG_GEMM_STAGES"); return e && std::atoi(e) == 2 ? 2 : 1; }();
        if (stages == 2) {
            const size_t lds2 = 65536;                  // two stages
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16x3_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                attr = true;
            }
            if (g.mode == GEMM_A_CONV) DSMI_LAUNCH((gemm_f16x3_kernel<true, 2>), grid3, dim3(256), lds2, s, g.ev, a, (const uint16_t*)g.a_sp);
            else DSMI_LAUNCH((gemm_f16x3_kernel<false, 2>), grid3, dim3(256), lds2, s, g.ev, a, (const uint16_t*)g.a_sp);
            return;
        }
        const size_t lds3 = 32768;                      // one stage: four 8-KiB operand planes of a 32-deep k-tile
        if (g.mode == GEMM_A_CONV) DSMI_LAUNCH((gemm_f16x3_kernel<true, 1>), grid3, dim3(256), lds3, s, g.ev, a, (const uint16_t*)g.a_sp);
        else DSMI_LAUNCH((gemm_f16x3_kernel<false, 1>), grid3, dim3(256), lds3, s, g.ev, a, (const uint16_t*)g.a_sp);
        return;
#endif
    }
    GemmArgs a;
    a.a = g.a; a.a2 = g.a2; a.alpha = g.alpha; a.beta = g.beta; a.w = g.w; a.bias = g.bias; a.c = g.c;
    a.M = g.M; a.N = g.N; a.K = g.K; a.lda = g.lda; a.ldw = g.ldw; a.ldc = g.ldc;
    a.B = g.B; a.T = g.T; a.ys = g.ys; a.tiles_per_b = 0;
    int mtiles;
    if (g.mode == GEMM_A_CONV) {
        a.tiles_per_b = ceil_div(g.T, BM);
        mtiles = a.tiles_per_b * g.B;
    } else {
        mtiles = ceil_div(g.M, BM);
    }
    dim3 grid(ceil_div(g.N, BN), mtiles);
    const size_t lds_rm = 2 * (BM * LDT + BN * LDT) * sizeof(float);
    const size_t lds_cv = 2 * (BK * LDC + BN * LDT) * sizeof(float);
    switch (g.mode) {
        case GEMM_A_ROWMAJOR:
            DSMI_LAUNCH(gemm_f32_kernel<GEMM_A_ROWMAJOR>, grid, dim3(256), lds_rm, s, g.ev, a); break;
        case GEMM_A_SUM_BN:
            DSMI_LAUNCH(gemm_f32_kernel<GEMM_A_SUM_BN>, grid, dim3(256), lds_rm, s, g.ev, a); break;
        default:
            DSMI_LAUNCH(gemm_f32_kernel<GEMM_A_CONV>, grid, dim3(256), lds_cv, s, g.ev, a); break;
    }
}

}  // namespace dsmi
