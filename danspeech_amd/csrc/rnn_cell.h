// The recurrent cell shared by the persistent layer kernels: one (hidden unit, clip) of torch's GRU / LSTM / RNN(tanh) step,
// reference danspeech/deepspeech/model.py:114-122 (nn.GRU / nn.LSTM / nn.RNN inside BatchRNN), gate order as in torch
// (GRU r, z, n; LSTM i, f, g, o).  xg = the x-projection's gate pre-activations (b_ih included), hg = W_hh h + b_hh.
// rnn_step.hip keeps the plain expf / tanhf statement; these use the hardware exp and reciprocal (parity tests hold both
// to the same goldens).
#pragma once
#include "common.h"

namespace dsmi {

__device__ __forceinline__ float cell_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }
__device__ __forceinline__ float cell_tanh(float v) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * v)); }

// the same with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the correctly rounded division: 27 vector instructions fewer
// per GRU cell, which is what the ring kernel's cell slot is made of (rnn_persist_ring.hip); 1e-7 against the statement above
__device__ __forceinline__ float cell_sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float cell_tanh_fast(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }

// -> h_t of this (unit, clip); `live` = the step lies inside the clip (pad_packed_sequence zero otherwise: the state of a
// clip that has ended is 0, and a reverse chain stays at 0 until it reaches the clip's last frame).  cstate: LSTM cell state.
template <int KIND, bool FAST = false>
__device__ __forceinline__ float rnn_cell(const float* xg, const float* hg, float hprev, float& cstate, bool live) {
    auto sg = [](float v) { return FAST ? cell_sigmoid_fast(v) : cell_sigmoid(v); };
    auto th = [](float v) { return FAST ? cell_tanh_fast(v) : cell_tanh(v); };
    float hn;
    if constexpr (KIND == DSMI_RNN_GRU) {
        const float r = sg(xg[0] + hg[0]);
        const float zz = sg(xg[1] + hg[1]);
        const float n = th(xg[2] + r * hg[2]);
        hn = (1.f - zz) * n + zz * hprev;
    } else if constexpr (KIND == DSMI_RNN_LSTM) {
        const float ig = sg(xg[0] + hg[0]);
        const float fg = sg(xg[1] + hg[1]);
        const float gg = th(xg[2] + hg[2]);
        const float og = sg(xg[3] + hg[3]);
        const float cn = fg * cstate + ig * gg;
        hn = og * th(cn);
        if (live) cstate = cn;
    } else {
        hn = th(xg[0] + hg[0]);
    }
    return live ? hn : 0.f;
}

}  // namespace dsmi
