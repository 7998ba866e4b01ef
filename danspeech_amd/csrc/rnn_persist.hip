// Persistent recurrent layer: ALL time steps of both directions of one BatchRNN in ONE launch.
//
// Same arithmetic as rnn_step.hip (which stays as the general fallback and as the reference the
// parity tests compare this kernel with), different machine mapping:
//   * one workgroup per CU owns 8 hidden units of one direction for the whole sequence; every
//     wave keeps its K-slice of W_hh in REGISTERS for all T steps, so the weights cross the
//     fabric once per layer instead of once per step (the per-step launch re-fetches all of
//     W_hh from Infinity Cache/HBM every step: profiles/r01_pmc_rnn_step.md);
//   * the per-step all-to-all of h goes through the packed state buffer with the counter form
//     of the hand-off protocol of cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
//     "Valid forms" row 3: producers store their h granules write-through (sc1, one wave
//     instruction = whole 128-B lines), every storing wave drains vmcnt, workgroup barrier, ONE
//     lane adds to an agent-scope counter; consumers poll that counter with an sc1 load, pass a
//     workgroup barrier, and read h with sc1 loads only.  One counter per (direction, batch
//     tile, step), zeroed by a memset node before the launch; no flag is ever reused.
//   * every spin is bounded: on timeout the workgroup raises an error word and stops waiting,
//     so a lost workgroup can never hang the GPU (the host then reports DSMI_ERR_HIP).
// Requires all workgroups co-resident: the launcher checks grid <= number of CUs (the kernel
// reserves > 80 KiB of LDS so that at most one workgroup fits a CU) and otherwise falls back
// to the per-step path.
#include "common.h"

namespace dsmi {

namespace {

constexpr int PNW = 8;                 // waves per workgroup
constexpr int PNT = PNW * 64;
constexpr int PU = 8;                  // hidden units per workgroup (rnn geometry U)
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr size_t PERSIST_LDS = 82 * 1024;   // > half of the CU's 160 KiB: at most one workgroup per CU

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

struct PersistArgs {
    const float* whh[2]; const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; float* hpack; unsigned* cnt; unsigned* err;
    int B, T, G, H, Hs, nq, Np, nwg;
    unsigned long long* dbg;   // diagnostics build only: per-wave accumulated phase times [wg][wave][8]
};

__device__ __forceinline__ float psigmoid(float v) { return 1.f / (1.f + expf(-v)); }

#define PSTAMP(k)                                                                         \
    do {                                                                                  \
        if (STAMP) {                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();             \
            tacc[k] += now_ - tlast; tlast = now_;                                        \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    } while (0)

template <int KIND, int NQW, bool STAMP = false>
__global__ __launch_bounds__(PNT) void rnn_persist_kernel(PersistArgs p) {
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    // reduce buffer + new-state staging; padded past 80 KiB so that two workgroups never share a CU
    // (dynamic LDS, requested as PERSIST_LDS bytes at launch: a static pad would be optimised away)
    extern __shared__ __attribute__((aligned(16))) float plds[];
    float* red = plds;                               // [PNW][32][32]
    float* hstage = red + PNW * 32 * 32;             // [PU][32]
    int& s_dead = *reinterpret_cast<int*>(hstage + PU * 32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hk = lane >> 5;
    const int w = blockIdx.x, d = blockIdx.y, z = blockIdx.z;
    const int b0 = z * 32;
    const int nb = min(32, p.B - b0);
    const int GU = p.G * PU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    const int chain = d * gridDim.z + z;
    unsigned* cnt = p.cnt + (size_t)chain * p.T;
    if (tid == 0) s_dead = 0;

    // ---- resident operand: this wave's K-slice of the packed W_hh (1 KiB per instruction)
    const int q0 = (v * p.nq) / PNW, q1 = ((v + 1) * p.nq) / PNW;
    f32x4 wv[NQW];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(p.whh[d]) + ((size_t)w * p.nq) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NQW; ++i) wv[i] = wp[(size_t)min(q0 + i, q1 - 1) * 64];
    }
    // packed state, [parity][chain][nq][64 lanes][4]; accessed ONLY through sc1 buffer ops
    const size_t hp_par = (size_t)gridDim.y * gridDim.z * p.nq * 256;    // floats per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.hpack, 0, (int)(2 * hp_par * sizeof(float)), 0x00020000);
    const unsigned hchain = (unsigned)((size_t)chain * p.nq * 256 * sizeof(float));

    // epilogue role: threads 0..255 own (unit u = tid>>5, batch bl = tid&31)
    const int eu = tid >> 5, ebl = tid & 31;
    const int eunit = w * PU + eu, eb = b0 + ebl;
    const bool eact = tid < PU * 32 && ebl < nb && eunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = eact ? p.bhh[d][g * p.H + eunit] : 0.f;
    const int mylen = eact ? p.lens[eb] : 0;
    float hprev_own = 0.f, cprev_own = 0.f;   // this thread's h_{t-1}, c_{t-1}
    __syncthreads();

    for (int s = 0; s < p.T; ++s) {
        const int t = d == 0 ? s : p.T - 1 - s;
        // x-projection operands of this step do not depend on other workgroups: request them first
        float xg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) xg[g] = 0.f;
        if (eact) {
            const float* xr = p.xp + ((size_t)t * p.B + eb) * p.Np + xcol + eu;
#pragma unroll
            for (int g = 0; g < NG; ++g) xg[g] = xr[g * PU];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        PSTAMP(0);   // loop head + x-projection request
        if (s > 0) {
            // ---- wait until every workgroup of this chain has published h_{s-1}
            // (bounded: after a timeout here or anywhere else on the chip nobody waits any more)
            if (v == 0 && !s_dead) {
                unsigned spins = 0;
                while (__hip_atomic_load(&cnt[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.nwg) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { s_dead = 1; break; }
                    if (spins > SPIN_LIMIT) { atomicExch(p.err, 1u); s_dead = 1; break; }
                }
            }
            __syncthreads();
            PSTAMP(1);   // waiting for the other workgroups
            // ---- B operand: h_{s-1} of all units, this wave's K-slice, sc1 loads only
            const unsigned hbase = (unsigned)(((s - 1) & 1) * hp_par * sizeof(float)) + hchain + (unsigned)lane * 16u;
            f32x4 hv[NQW];
#pragma unroll
            for (int i = 0; i < NQW; ++i) {
                const int q = min(q0 + i, q1 - 1);
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(hrs, hbase + (unsigned)q * 1024u, 0, 16);
                hv[i] = __builtin_bit_cast(f32x4, raw);
            }
#pragma unroll
            for (int i = 0; i < NQW; ++i) {
                if (q0 + i < q1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i][c], hv[i][c], acc, 0, 0, 0);
                }
            }
        }
        PSTAMP(2);   // h load + MFMA chain
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hk;
            red[(v * 32 + i) * 32 + li] = acc[r];
        }
        __syncthreads();
        PSTAMP(3);   // partial tiles to LDS + barrier (wave skew)
        // ---- K-split reduction (fixed order) + cell, one (unit, batch) pair per thread
        if (tid < PU * 32) {
            float hn = 0.f;
            if (eact) {
                float hg[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int row = g * PU + eu;
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < PNW; ++k) sum += red[(k * 32 + row) * 32 + ebl];
                    hg[g] = sum + bh[g];
                }
                if (KIND == DSMI_RNN_GRU) {
                    const float r = psigmoid(xg[0] + hg[0]);
                    const float zz = psigmoid(xg[1] + hg[1]);
                    const float n = tanhf(xg[2] + r * hg[2]);
                    hn = (1.f - zz) * n + zz * hprev_own;
                } else if (KIND == DSMI_RNN_LSTM) {
                    const float ig = psigmoid(xg[0] + hg[0]);
                    const float fg = psigmoid(xg[1] + hg[1]);
                    const float gg = tanhf(xg[2] + hg[2]);
                    const float og = psigmoid(xg[3] + hg[3]);
                    const float cn = fg * cprev_own + ig * gg;
                    hn = og * tanhf(cn);
                    if (t < mylen) cprev_own = cn;
                } else {
                    hn = tanhf(xg[0] + hg[0]);
                }
                if (t >= mylen) hn = 0.f;         // pad_packed_sequence zero; the reverse chain stays at 0 until len-1
                hprev_own = hn;
                p.out[d][((size_t)t * p.B + eb) * p.Hs + eunit] = hn;
            } else if (ebl < nb && eunit < p.Hs) {
                p.out[d][((size_t)t * p.B + eb) * p.Hs + eunit] = 0.f;     // padding units of the last workgroup
            }
            hstage[eu * 32 + ebl] = hn;
        }
        __syncthreads();
        PSTAMP(4);   // reduction + cell
        // ---- publish: wave 0 writes this workgroup's 8 units x 32 batch rows as one 1-KiB sc1 store
        if (v == 0) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = hstage[(4 * hk + c) * 32 + li];
            const unsigned off = (unsigned)((s & 1) * hp_par * sizeof(float)) + hchain + (unsigned)w * 1024u + (unsigned)lane * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), hrs, off, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&cnt[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PSTAMP(5);   // publish + drain + signal
    }
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * PNW + v) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    }
}

template <int KIND>
bool launch_kind(const PersistArgs& a, int D, int nz, hipStream_t s, const EvPair& ev) {
    const int nqw = ceil_div(a.nq, PNW);
    const dim3 grid(a.nwg, D, nz), block(PNT);
    if (a.dbg) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_kernel<KIND, 13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PERSIST_LDS);
        hipLaunchKernelGGL((rnn_persist_kernel<KIND, 13, true>), grid, block, PERSIST_LDS, s, a);
        return true;
    }
#define LAUNCH_P(N)                                                                                                  \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_kernel<KIND, N>),                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)PERSIST_LDS);                      \
        DSMI_LAUNCH((rnn_persist_kernel<KIND, N>), grid, block, PERSIST_LDS, s, ev, a);                               \
    } while (0)
    if (nqw <= 4) LAUNCH_P(4);
    else if (nqw <= 7) LAUNCH_P(7);
    else if (nqw <= 10) LAUNCH_P(10);
    else if (nqw <= 13) LAUNCH_P(13);
    else if (nqw <= 16) LAUNCH_P(16);
    else if (nqw <= 19) LAUNCH_P(19);
    else return false;
    return true;
}

}  // namespace

bool rnn_persist_eligible(const RnnGeom& g, int B, int n_cus) {
    if (g.U != PU || (g.H % 8) != 0) return false;
    if (ceil_div(g.nq, PNW) > 19) return false;
    return g.nwg * g.D * ceil_div(B, 32) <= n_cus;
}

bool launch_rnn_persist(const RnnPersistLaunch& p, hipStream_t s) {
    PersistArgs a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh_packed[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack; a.cnt = p.counters; a.err = p.err; a.dbg = p.dbg;
    a.B = p.B; a.T = p.T; a.G = p.g.G; a.H = p.g.H; a.Hs = p.g.Kp; a.nq = p.g.nq; a.Np = p.g.Np; a.nwg = p.g.nwg;
    const int nz = ceil_div(p.B, 32);
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_kind<DSMI_RNN_GRU>(a, p.g.D, nz, s, p.ev);
        case DSMI_RNN_LSTM: return launch_kind<DSMI_RNN_LSTM>(a, p.g.D, nz, s, p.ev);
        default: return launch_kind<DSMI_RNN_TANH>(a, p.g.D, nz, s, p.ev);
    }
}

}  // namespace dsmi
