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

void launch_gemm(const GemmLaunch& g, hipStream_t s) {
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
