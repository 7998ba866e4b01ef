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
// Work split: one workgroup owns U = floor(32/G) hidden units of one direction, i.e. G*U <= 32
// gate rows = one MFMA tile of rows; its 4 waves split K = H four ways and the partial
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
    g.U = 32 / g.G;
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
    int B, T, step, G, H, U, Hs, nq, Np, nwg;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

template <int KIND, int NBT>
__global__ __launch_bounds__(256) void rnn_step_kernel(StepArgs p) {
    __shared__ __attribute__((aligned(16))) float red[4 * NBT * 32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;
    const int li = lane & 31, hk = lane >> 5;
    const int w = blockIdx.x, d = blockIdx.y;
    const int b0 = blockIdx.z * (32 * NBT);
    const int t = d == 0 ? p.step : p.T - 1 - p.step;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const bool has_prev = tprev >= 0 && tprev < p.T;
    float* outd = p.out[d];
    const float* hprev = outd + (size_t)(has_prev ? tprev : 0) * p.B * p.Hs;

    f32x16 acc[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bt][r] = 0.f;

    if (has_prev) {
        const int q0 = (v * p.nq) / 4, q1 = ((v + 1) * p.nq) / 4;
        const f32x4* wp = reinterpret_cast<const f32x4*>(p.whh[d]) + ((size_t)w * p.nq) * 64 + lane;

        for (int q = q0; q < q1; ++q) {
            const f32x4 wv = wp[(size_t)q * 64];
            f32x4 hv[NBT];
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) {
                const int j = b0 + bt * 32 + li;
                hv[bt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < p.B) hv[bt] = *reinterpret_cast<const f32x4*>(hprev + (size_t)j * p.Hs + 8 * q + 4 * hk);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt)
                    acc[bt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c], hv[bt][c], acc[bt], 0, 0, 0);
        }
    }
    // partial tiles -> LDS: D[i][j], col j = lane&31 (batch), row i = (r&3)+8(r>>2)+4hk (gate row)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hk;
            red[((v * NBT + bt) * 32 + i) * 32 + li] = acc[bt][r];
        }
    __syncthreads();

    const int nb = min(32 * NBT, p.B - b0);
    const int GU = p.G * p.U;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    for (int pr = tid; pr < p.U * nb; pr += 256) {
        const int bl = pr % nb, u = pr / nb;
        const int b = b0 + bl;
        const int unit = w * p.U + u;
        if (unit >= p.Hs) continue;
        float* o = outd + ((size_t)t * p.B + b) * p.Hs + unit;
        if (unit >= p.H) { *o = 0.f; continue; }
        const int bt = bl >> 5, bj = bl & 31;
        float hg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < (KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1))) {
                const int row = g * p.U + u;
                float s = red[((0 * NBT + bt) * 32 + row) * 32 + bj];
                s += red[((1 * NBT + bt) * 32 + row) * 32 + bj];
                s += red[((2 * NBT + bt) * 32 + row) * 32 + bj];
                s += red[((3 * NBT + bt) * 32 + row) * 32 + bj];
                hg[g] = s + p.bhh[d][g * p.H + unit];
            }
        }
        const float* xr = p.xp + ((size_t)t * p.B + b) * p.Np + xcol + u;
        const float hp = has_prev ? hprev[(size_t)b * p.Hs + unit] : 0.f;
        const bool active = t < p.lens[b];
        float hn;
        if (KIND == DSMI_RNN_GRU) {
            const float r = sigmoidf_(xr[0] + hg[0]);
            const float z = sigmoidf_(xr[p.U] + hg[1]);
            const float n = tanhf(xr[2 * p.U] + r * hg[2]);
            hn = (1.f - z) * n + z * hp;
        } else if (KIND == DSMI_RNN_LSTM) {
            float* cp = p.cst[d] + (size_t)b * p.Hs + unit;
            const float c0 = p.step == 0 ? 0.f : *cp;
            const float ig = sigmoidf_(xr[0] + hg[0]);
            const float fg = sigmoidf_(xr[p.U] + hg[1]);
            const float gg = tanhf(xr[2 * p.U] + hg[2]);
            const float og = sigmoidf_(xr[3 * p.U] + hg[3]);
            const float cn = fg * c0 + ig * gg;
            hn = og * tanhf(cn);
            *cp = active ? cn : c0;
        } else {
            hn = tanhf(xr[0] + hg[0]);
        }
        *o = active ? hn : 0.f;
    }
}

template <int KIND>
static void launch_kind(const StepArgs& a, int D, hipStream_t s, const EvPair& ev) {
    const int B = a.B;
    if (B <= 32) {
        DSMI_LAUNCH((rnn_step_kernel<KIND, 1>), dim3(a.nwg, D, 1), dim3(256), 0, s, ev, a);
    } else if (B <= 64) {
        DSMI_LAUNCH((rnn_step_kernel<KIND, 2>), dim3(a.nwg, D, 1), dim3(256), 0, s, ev, a);
    } else {
        DSMI_LAUNCH((rnn_step_kernel<KIND, 4>), dim3(a.nwg, D, ceil_div(B, 128)), dim3(256), 0, s, ev, a);
    }
}

void launch_rnn_step(const RnnStepLaunch& p, hipStream_t s) {
    StepArgs a;
    for (int d = 0; d < 2; ++d) {
        a.whh[d] = p.whh_packed[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; a.cst[d] = p.cstate[d];
    }
    a.xp = p.xp; a.lens = p.lens_dev; a.B = p.B; a.T = p.T; a.step = p.step;
    a.G = p.g.G; a.H = p.g.H; a.U = p.g.U; a.Hs = p.g.Kp; a.nq = p.g.nq; a.Np = p.g.Np; a.nwg = p.g.nwg;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: launch_kind<DSMI_RNN_GRU>(a, p.g.D, s, p.ev); break;
        case DSMI_RNN_LSTM: launch_kind<DSMI_RNN_LSTM>(a, p.g.D, s, p.ev); break;
        default: launch_kind<DSMI_RNN_TANH>(a, p.g.D, s, p.ev); break;
    }
}

}  // namespace dsmi
