// One time step of one recurrent layer, both directions in one launch (gfx950).
//
// Replaces the per-step cell loop of torch.nn.{GRU,LSTM,RNN} that the reference runs
// through BatchRNN.forward (danspeech/deepspeech/model.py:114-122) on a packed batch:
//   hg   = h_{t-1} @ W_hh^T + b_hh                (fp32 MFMA, v_mfma_f32_32x32x2_f32)
//   cell = gate math on (x-projection[t], hg)     (torch gate order: GRU r,z,n / LSTM i,f,g,o)
//   out[t][b] = cell if t < len[b] else 0         (pad_packed_sequence zero padding)
// The forward direction handles t = step, the reverse direction t = T-1-step; a sequence
// shorter than T simply stays at h = 0 in the reverse direction until t = len-1, which is
// pack_padded_sequence's semantics without any gather.  h_{t-1} is read back from the
// layer's own output buffer (row t-1 / t+1), so the output doubles as the state.
//
// Work split: one workgroup owns U = 8 hidden units of one direction, i.e. G*U <= 32
// gate rows = one MFMA tile of rows; its 8 waves split K = H eight ways and the partial
// tiles meet in LDS (fixed summation order -> run-to-run deterministic).  W_hh is
// pre-packed on the host into the exact lane order of the A operand (1 KiB per wave
// instruction, fully coalesced); a workgroup always reads the same slice, so it stays
// resident in the L2 of the XCD that blockIdx.x % 8 maps to.
#include "common.h"
#include <cmath>

namespace dsmi {

RnnGeom make_rnn_geom(int kind, int H, int D) {
    RnnGeom g;
    g.kind = kind;
    g.G = kind == DSMI_RNN_GRU ? 3 : (kind == DSMI_RNN_LSTM ? 4 : 1);
    g.H = H;
    g.U = 8;
    g.nwg = ceil_div(H, g.U);
    g.Kp = round_up(H, 8);
    g.nq = g.Kp / 8;
    g.D = D;
    g.Np = D * g.nwg * g.G * g.U;
    return g;
}

int rnn_src_row(const RnnGeom& g, int col, int* dir_out) {
    const int GU = g.G * g.U;
    const int per_dir = g.nwg * GU;
    const int d = col / per_dir;
    const int r = col % per_dir;
    const int w = r / GU, i = r % GU;
    const int gate = i / g.U, u = i % g.U;
    const int unit = w * g.U + u;
    if (dir_out) *dir_out = d;
    if (unit >= g.H) return -1;
    return gate * g.H + unit;
}

std::vector<float> pack_whh(const RnnGeom& g, const float* w_hh) {
    std::vector<float> out((size_t)g.nwg * g.nq * 64 * 4, 0.f);
    const int GU = g.G * g.U;
    for (int w = 0; w < g.nwg; ++w)
        for (int q = 0; q < g.nq; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hk = lane >> 5;
                if (i >= GU) continue;
                const int gate = i / g.U, u = i % g.U;
                const int unit = w * g.U + u;
                if (unit >= g.H) continue;
                for (int c = 0; c < 4; ++c) {
                    const int k = 8 * q + 4 * hk + c;
                    if (k >= g.H) continue;
                    out[(((size_t)w * g.nq + q) * 64 + lane) * 4 + c] = w_hh[(size_t)(gate * g.H + unit) * g.H + k];
                }
            }
    return out;
}

struct StepArgs {
    const float* whh[2]; const float* bhh[2]; const float* xp; float* out[2]; float* cst[2];
    const int32_t* lens;
    float* hpack;              // [2 parity][D][ceil(B/32)][nq][64][4]: h in the B operand's lane order
    int B, T, step, G, H, U, Hs, nq, Np, nwg;
    const float* hcarry;       // streaming (D = 1): [B][Hs] state carried in from the previous chunk, read at step 0
    int pbase;                 // streaming: parity offset of the packed state so that chunks continue each other
    unsigned long long* dbg;   // diagnostics build only (STAMP = true)
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

#define STAMP_AT(k)                                                                                   \
    do {                                                                                              \
        if (STAMP && lane == 0) {                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            p.dbg[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + v) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    } while (0)

// KIND: cell type.  NQW: compile-time upper bound of the 8-wide k-blocks one wave owns (all of
// its W_hh and h operands are requested in one burst before the first MFMA, so the whole
// weight slice is in flight at once).  NW: waves per workgroup = K-split factor.
// One workgroup handles 32 batch rows (blockIdx.z selects the batch tile).
template <int KIND, int NQW, int NW, bool STAMP = false>
__global__ __launch_bounds__(NW * 64) void rnn_step_kernel(StepArgs p) {
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NT = NW * 64;
    constexpr int PP = ((32 / NG) * 32 + NT - 1) / NT;   // (unit, batch) pairs per thread, upper bound
    __shared__ __attribute__((aligned(16))) float red[NW * 32 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id, provably uniform -> scalar branches
    const int li = lane & 31, hk = lane >> 5;
    const int w = blockIdx.x, d = blockIdx.y;
    const int b0 = blockIdx.z * 32;
    const int t = d == 0 ? p.step : p.T - 1 - p.step;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool carry0 = p.step == 0 && p.hcarry != nullptr;       // BatchRNNStream: rnn(x, hx=previous_hidden), model.py:229
    const bool has_prev = (tprev >= 0 && tprev < p.T) || carry0;
    float* outd = p.out[d];
    const float* hprev = carry0 ? p.hcarry : outd + (size_t)(has_prev ? tprev : 0) * p.B * p.Hs;
    const int par = (p.step + p.pbase) & 1;
    const int nb = min(32, p.B - b0);
    const int GU = p.G * p.U;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    STAMP_AT(0);

    // ---- 1. request this wave's whole operand slice: W_hh (packed, 1 KiB per instruction) and h
    const int q0 = (v * p.nq) / NW, q1 = ((v + 1) * p.nq) / NW;
    f32x4 wv[NQW], hv[NQW];
    if (has_prev) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(p.whh[d]) + ((size_t)w * p.nq) * 64 + lane;
        const f32x4* hq = reinterpret_cast<const f32x4*>(p.hpack) +
                          ((((size_t)(par ^ 1) * gridDim.y + d) * gridDim.z + blockIdx.z) * p.nq) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            const int q = min(q0 + i, max(q1 - 1, q0));      // clamp: duplicates (and a wave without k-blocks) are skipped below
            wv[i] = wp[(size_t)q * 64];
            hv[i] = hq[(size_t)q * 64];
        }
    }
    STAMP_AT(1);

    // ---- 2. request the epilogue's operands (independent of the MFMA phase, consumed after it)
    float xg[PP][NG], bh[PP][NG], hp[PP];
    int slen[PP];   // raw length: compared only in the epilogue so that no wait lands here
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int pr = tid + i * NT;
        const int bl = pr & 31, u = pr >> 5;
        const int unit = w * p.U + u;
        const bool ok = pr < p.U * 32 && bl < nb && unit < p.H;
        const int b = b0 + bl;
        hp[i] = 0.f; slen[i] = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) { xg[i][g] = 0.f; bh[i][g] = 0.f; }
        if (ok) {
            const float* xr = p.xp + ((size_t)t * p.B + b) * p.Np + xcol + u;
#pragma unroll
            for (int g = 0; g < NG; ++g) { xg[i][g] = xr[g * p.U]; bh[i][g] = p.bhh[d][g * p.H + unit]; }
            if (has_prev) hp[i] = hprev[(size_t)b * p.Hs + unit];
            slen[i] = p.lens[b];
        }
    }

    // ---- 3. MFMA: D[gate row][batch] += W[gate row][k] * h[batch][k] over this wave's k range
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (has_prev) {
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            if (q0 + i < q1) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i][c], hv[i][c], acc, 0, 0, 0);
            }
            if (i == 3) STAMP_AT(2);
        }
    }
    STAMP_AT(3);
    // partial tiles -> LDS: D[i][j], col j = lane&31 (batch), row i = (r&3)+8(r>>2)+4hk (gate row)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hk;
        red[(v * 32 + i) * 32 + li] = acc[r];
    }
    __syncthreads();
    STAMP_AT(4);

    // ---- 4. K-split reduction in fixed order + cell.  The new state goes to the layer output
    // (natural layout) and to the packed copy the next step's B operand reads.
    float* hw = p.hpack + ((((size_t)par * gridDim.y + d) * gridDim.z + blockIdx.z) * p.nq) * 256;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int pr = tid + i * NT;
        if (pr >= p.U * 32) break;
        const int bl = pr & 31, u = pr >> 5;
        const int b = b0 + bl;
        const int unit = w * p.U + u;
        if (unit >= p.Hs) continue;
        float* hwp = hw + ((size_t)(unit >> 3) * 64 + ((unit >> 2) & 1) * 32 + bl) * 4 + (unit & 3);
        if (bl >= nb) { *hwp = 0.f; continue; }      // batch padding of the tile: zero operand rows
        float* o = outd + ((size_t)t * p.B + b) * p.Hs + unit;
        if (unit >= p.H) { *o = 0.f; *hwp = 0.f; continue; }
        float hg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int row = g * p.U + u;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s += red[(k * 32 + row) * 32 + bl];
            hg[g] = s + bh[i][g];
        }
        float hn;
        if (KIND == DSMI_RNN_GRU) {
            const float r = sigmoidf_(xg[i][0] + hg[0]);
            const float z = sigmoidf_(xg[i][1] + hg[1]);
            const float n = tanhf(xg[i][2] + r * hg[2]);
            hn = (1.f - z) * n + z * hp[i];
        } else if (KIND == DSMI_RNN_LSTM) {
            float* cp = p.cst[d] + (size_t)b * p.Hs + unit;
            const float c0 = (p.step == 0 && !carry0) ? 0.f : *cp;
            const float ig = sigmoidf_(xg[i][0] + hg[0]);
            const float fg = sigmoidf_(xg[i][1] + hg[1]);
            const float gg = tanhf(xg[i][2] + hg[2]);
            const float og = sigmoidf_(xg[i][3] + hg[3]);
            const float cn = fg * c0 + ig * gg;
            hn = og * tanhf(cn);
            *cp = t < slen[i] ? cn : c0;
        } else {
            hn = tanhf(xg[i][0] + hg[0]);
        }
        hn = t < slen[i] ? hn : 0.f;
        *o = hn;
        *hwp = hn;
    }
    STAMP_AT(5);
}

template <int KIND, int NW>
static void launch_nw(const StepArgs& a, int D, hipStream_t s, const EvPair& ev) {
    const int nqw = ceil_div(a.nq, NW);
    const dim3 grid(a.nwg, D, ceil_div(a.B, 32)), block(NW * 64);
    if (a.dbg) { hipLaunchKernelGGL((rnn_step_kernel<KIND, 13, NW, true>), grid, block, 0, s, a); return; }
    if (nqw <= 4) DSMI_LAUNCH((rnn_step_kernel<KIND, 4, NW>), grid, block, 0, s, ev, a);
    else if (nqw <= 7) DSMI_LAUNCH((rnn_step_kernel<KIND, 7, NW>), grid, block, 0, s, ev, a);
    else if (nqw <= 10) DSMI_LAUNCH((rnn_step_kernel<KIND, 10, NW>), grid, block, 0, s, ev, a);
    else if (nqw <= 13) DSMI_LAUNCH((rnn_step_kernel<KIND, 13, NW>), grid, block, 0, s, ev, a);
    else if (nqw <= 16) DSMI_LAUNCH((rnn_step_kernel<KIND, 16, NW>), grid, block, 0, s, ev, a);
    else DSMI_LAUNCH((rnn_step_kernel<KIND, 19, NW>), grid, block, 0, s, ev, a);
}

template <int KIND>
static void launch_kind(const StepArgs& a, int D, hipStream_t s, const EvPair& ev) {
    // 8 waves split K up to nq = 152 k-blocks (H <= 1216); wider layers use 16 waves (H <= 2432)
    if (ceil_div(a.nq, 8) <= 19) launch_nw<KIND, 8>(a, D, s, ev);
    else launch_nw<KIND, 16>(a, D, s, ev);
}

void launch_rnn_step(const RnnStepLaunch& p, hipStream_t s) {
    StepArgs a;
    a.dbg = p.dbg;
    a.hpack = p.hpack;
    for (int d = 0; d < 2; ++d) {
        a.whh[d] = p.whh_packed[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; a.cst[d] = p.cstate[d];
    }
    a.xp = p.xp; a.lens = p.lens_dev; a.B = p.B; a.T = p.T; a.step = p.step;
    a.hcarry = p.hcarry; a.pbase = p.pbase;
    a.G = p.g.G; a.H = p.g.H; a.U = p.g.U; a.Hs = p.g.Kp; a.nq = p.g.nq; a.Np = p.g.Np; a.nwg = p.g.nwg;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: launch_kind<DSMI_RNN_GRU>(a, p.g.D, s, p.ev); break;
        case DSMI_RNN_LSTM: launch_kind<DSMI_RNN_LSTM>(a, p.g.D, s, p.ev); break;
        default: launch_kind<DSMI_RNN_TANH>(a, p.g.D, s, p.ev); break;
    }
}

}  // namespace dsmi
