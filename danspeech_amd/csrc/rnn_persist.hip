// Persistent recurrent layer: ALL time steps of both directions of one BatchRNN in ONE launch.
//
// Same cell arithmetic as rnn_step.hip (which stays as the general fallback and as the plain
// fp32-MFMA statement of the step that the parity tests also run), different machine mapping:
//   * one workgroup per CU owns 8 hidden units of one direction for the whole sequence; every
//     wave keeps its K-slice of W_hh in REGISTERS for all T steps, so the weights cross the
//     fabric once per layer instead of once per step (the per-step launch re-fetches all of
//     W_hh from Infinity Cache/HBM every step: profiles/r01_pmc_rnn_step.md);
//   * the h . W_hh^T product runs on the fp16 MFMA (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA
//     rate) with BOTH operands split into two fp16 terms, x = hi + lo * 2^-11 (hi = fp16(x),
//     lo = fp16((x - hi) * 2^11): 22 mantissa bits, lo kept scaled so that it stays a normal
//     number), three products: hi.hi in one fp32 accumulator, hi.lo + lo.hi in a second one that
//     is folded in with 2^-11 at the end (lo.lo < 2^-22 is dropped).  Measured error vs an fp64
//     reference at K = 800: 2.6e-7, against 1.0e-6 for the fp32 MFMA chain and 8.0e-7 for the
//     earlier three-term bf16 split with six products (tools/exp/split_mfma_accuracy.hip): better than
//     fp32-MFMA accuracy at 3/16 of its time.  Ranges are safe by construction: |h| <= 1 for
//     every cell type, |W_hh| is checked at load time (fp16 max 65504).
//     W_hh is split once on the host; the 256 cell threads of a workgroup split the state values
//     they produce and store them straight into the packed buffer, already in B-operand lane order
//     (2-byte stores, 128 contiguous bytes per wave instruction);
//   * the per-step all-to-all of h goes through the packed state buffer with the counter form
//     of the hand-off protocol of cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
//     "Valid forms" row 3: producers store their h granules write-through (sc1), every storing
//     wave drains vmcnt, workgroup barrier, ONE lane adds to an agent-scope counter; consumers poll that counter with an sc1 load, pass a
//     workgroup barrier, and read h with sc1 loads only.  One counter per (direction, batch
//     tile, step), zeroed by a memset node before the launch; no flag is ever reused.
//   * batches above 32 clips: every workgroup walks the 32-clip batch tiles one after the other each
//     step with the same resident weights (MULTI); layers too wide for both directions to be
//     co-resident run one launch per direction;
//   * every spin is bounded: on timeout the workgroup raises an error word and stops waiting,
//     so a lost workgroup can never hang the GPU (the host then reports DSMI_ERR_HIP).
// Requires all workgroups co-resident: the launcher checks grid <= number of CUs (the kernel
// requests > 80 KiB of LDS so that at most one workgroup fits a CU) and otherwise falls back
// to the per-step path.
#include "common.h"
#include "rnn_cell.h"
#include <cstring>

namespace dsmi {

namespace {

constexpr int PNW = 8;                 // waves per workgroup
constexpr int PNT = PNW * 64;
constexpr int PU = 8;                  // hidden units per workgroup (rnn geometry U)
constexpr int RP = 40;                 // row pitch of the reduce buffer in words
constexpr size_t PERSIST_LDS = 82 * 1024;   // > half of the CU's 160 KiB: at most one workgroup per CU

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;   // 2^11: the lo term is stored scaled

struct PersistArgs {
    const uint16_t* whh_sp[2];   // [wg][pair][plane 2][lane 64][8 fp16]: split W_hh (hi, lo * 2^11) in A-operand lane order
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack_sp; unsigned* cnt; unsigned* err;
    int B, T, G, H, Hs, npair, Np, nwg;
    int nz;                    // batch tiles of 32 clips, all walked by every workgroup each step
    int d0, nd;                // first direction of this launch, directions in the layer (chains are numbered over the layer)
    unsigned spin_limit;       // polls of one wait before the workgroup raises *err and stops waiting
    int drop_wg, drop_step;    // test hook (DSMI_DEBUG_DROP_SIGNAL): this workgroup of direction 0 never signals that step (-1: off)
    unsigned long long* dbg;   // diagnostics build only: per-wave accumulated phase times [wg][wave][8]
};

// Cell non-linearities on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm's
// branchy expf / tanhf: the cell sits on the critical path of every step (-0.3 us of 4.56 per step).  Absolute
// error ~1e-7 on outputs in [-1, 1] -- tanh(x) = 1 - 2 / (1 + e^2x) loses RELATIVE accuracy near 0 but h is consumed
// at absolute scale -- measured end to end: same parity margins as libm (tests/test_gpu_parity.py).

#define PSTAMP(k)                                                                         \
    do {                                                                                  \
        if (STAMP) {                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();             \
            tacc[k] += now_ - tlast; tlast = now_;                                        \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    } while (0)

constexpr int PMAXZ = 8;               // batch tiles per launch (B <= 256)

template <int KIND, int NPW, bool MULTI, bool STAMP = false>
__global__ __launch_bounds__(PNT) void rnn_persist_kernel(PersistArgs p) {
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    // reduce buffer + new-state staging; padded past 80 KiB so that two workgroups never share a CU
    // (dynamic LDS, requested as PERSIST_LDS bytes at launch: a static pad would be optimised away)
    extern __shared__ __attribute__((aligned(16))) float plds[];
    float* red = plds;                               // [PNW][32 gate rows][RP]: row pitch 40 words keeps both the MFMA-layout
                                                     // writes (lanes along the batch) and the cell's reads (lanes along units) conflict-free
    int& s_dead = *reinterpret_cast<int*>(red + PNW * 32 * RP);
    // MULTI: per-tile recurrent state of the epilogue threads (a single tile keeps it in registers)
    float* st_h = red + PNW * 32 * RP + 32;          // [PMAXZ][PU*32]
    float* st_c = st_h + PMAXZ * PU * 32;            // [PMAXZ][PU*32]
    int* st_len = reinterpret_cast<int*>(st_c + PMAXZ * PU * 32);   // [PMAXZ][PU*32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hk = lane >> 5;
    const int w = blockIdx.x, d = p.d0 + blockIdx.y;
    const int nz = MULTI ? p.nz : 1;
    const int GU = p.G * PU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    if (tid == 0) s_dead = 0;

    // ---- resident operand: this wave's pairs of the split W_hh (two 1-KiB planes per 16 k)
    const int p0 = (v * p.npair) / PNW, p1 = ((v + 1) * p.npair) / PNW;
    f16x8 wv[NPW][2];
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh_sp[d]) + ((size_t)w * p.npair) * 128 + lane;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pq = min(p0 + i, max(p1 - 1, p0));     // a wave with no pair of its own (small H) reads pair p0 and skips the MFMAs
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[i][pl] = __builtin_bit_cast(f16x8, wp[((size_t)pq * 2 + pl) * 64]);
        }
    }
    // packed split state, [parity][chain][pair][plane][hk][batch j][8 fp16]; accessed ONLY through sc1 buffer ops
    const size_t hp_par = (size_t)p.nd * p.nz * p.npair * 2048;    // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack_sp, 0, (int)(2 * hp_par), 0x00020000);

    // epilogue role: threads 0..255 own (batch bl = tid>>3, unit u = tid&7) of every tile: the 8 units of a batch row sit
    // in 8 adjacent lanes, so a wave's 2-byte stores of the new state are 128 contiguous bytes of packed granules
    const int eu = tid & 7, ebl = tid >> 3;
    const int eunit = w * PU + eu;
    const bool eunit_ok = tid < PU * 32 && eunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = eunit_ok ? p.bhh[d][g * p.H + eunit] : 0.f;
    int mylen = (eunit_ok && ebl < p.B) ? p.lens[ebl] : 0;
    float hprev_own = 0.f, cprev_own = 0.f;   // this thread's h_{t-1}, c_{t-1}
    if (MULTI && tid < PU * 32) {
        for (int z = 0; z < nz; ++z) {
            const int eb = z * 32 + ebl;
            st_h[z * PU * 32 + tid] = 0.f; st_c[z * PU * 32 + tid] = 0.f;
            st_len[z * PU * 32 + tid] = (eunit_ok && eb < p.B) ? p.lens[eb] : 0;
        }
    }
    __syncthreads();

    unsigned* pend = nullptr;      // MULTI: counter of the tile just published, not yet signalled
    for (int s = 0; s < p.T; ++s) {
        const int t = d == 0 ? s : p.T - 1 - s;
      for (int z = 0; z < nz; ++z) {
        const int b0 = z * 32;
        const int nb = min(32, p.B - b0);
        const int eb = b0 + ebl;
        const bool eact = eunit_ok && ebl < nb;
        const int chain = d * p.nz + z;
        unsigned* cnt = p.cnt + (size_t)chain * p.T;
        const unsigned hchain = (unsigned)((size_t)chain * p.npair * 2048);
        if (MULTI && tid < PU * 32) { mylen = st_len[z * PU * 32 + tid]; hprev_own = st_h[z * PU * 32 + tid]; cprev_own = st_c[z * PU * 32 + tid]; }
        // x-projection operands of this step do not depend on other workgroups: request them first
        float xg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) xg[g] = 0.f;
        if (eact) {
            const float* xr = p.xp + ((size_t)t * p.B + eb) * p.Np + xcol + eu;
#pragma unroll
            for (int g = 0; g < NG; ++g) xg[g] = xr[g * PU];
        }
        f32x16 acc, acl;        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acl[r] = 0.f; }
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
                    if (spins > p.spin_limit) { atomicExch(p.err, 1u); s_dead = 1; break; }
                }
            }
            __syncthreads();
            PSTAMP(1);   // waiting for the other workgroups
            // ---- B operand: split h_{s-1}, this wave's pairs, sc1 loads only (lane = hk*32 + j)
            const unsigned hbase = (unsigned)(((s - 1) & 1) * hp_par) + hchain + (unsigned)lane * 16u;
            f16x8 hv[NPW][2];
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int pq = min(p0 + i, max(p1 - 1, p0));     // a wave with no pair of its own (small H) reads pair p0 and skips the MFMAs
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    hv[i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        hrs, hbase + (unsigned)(pq * 2 + pl) * 1024u, 0, 16));
            }
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                if (p0 + i < p1) {
                    acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[i][1], hv[i][0], acl, 0, 0, 0);
                    acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[i][0], hv[i][1], acl, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[i][0], hv[i][0], acc, 0, 0, 0);
                }
            }
        }
        PSTAMP(2);   // h load + MFMA chain
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hk;
            red[(v * 32 + i) * RP + li] = acc[r] + acl[r] * kLoInv;
        }
        if (MULTI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile's state stores are acknowledged
        __syncthreads();
        // MULTI: the previous tile is signalled only now -- its write-through stores drained while this tile waited,
        // loaded and multiplied, so no wave ever sits in a drain; consumers need that tile a whole round of tiles later
        if (MULTI && tid == 0 && pend) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    for (int k = 0; k < PNW; ++k) sum += red[(k * 32 + row) * RP + ebl];
                    hg[g] = sum + bh[g];
                }
                hn = rnn_cell<KIND>(xg, hg, hprev_own, cprev_own, t < mylen);
                hprev_own = hn;
                if (MULTI) { st_h[z * PU * 32 + tid] = hn; if (KIND == DSMI_RNN_LSTM) st_c[z * PU * 32 + tid] = cprev_own; }
                p.out[d][((size_t)t * p.B + eb) * p.Hs + eunit] = hn;
            } else if (ebl < nb && eunit < p.Hs) {
                p.out[d][((size_t)t * p.B + eb) * p.Hs + eunit] = 0.f;     // padding units of the last workgroup
            }
            // ---- publish straight from the cell threads: hi / lo fp16 terms of (batch row ebl, unit eu) at byte
            // 16 * ebl + 2 * eu of this workgroup's granule block = 2 * tid: contiguous per wave, write-through (sc1)
            const _Float16 h1 = (_Float16)hn;
            const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
            const unsigned off = (unsigned)((s & 1) * hp_par) + hchain + (unsigned)(w >> 1) * 2048u +
                                 (unsigned)(w & 1) * 512u + (unsigned)tid * 2u;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h1), hrs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h2), hrs, off + 1024u, 0, 16);
        }
        PSTAMP(4);   // reduction + cell + publish
        if (MULTI) {
            pend = &cnt[s];                                        // signalled after the next tile's MFMAs (see above)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own stores
            __syncthreads();
            if (tid == 0 && !(d == 0 && w == p.drop_wg && s == p.drop_step)) __hip_atomic_fetch_add(&cnt[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PSTAMP(5);   // publish + drain + signal
      }
    }
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * PNW + v) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    }
}

template <int KIND>
bool launch_kind(const PersistArgs& a, int ny, hipStream_t s, const EvPair& ev) {
    const int npw = ceil_div(a.npair, PNW);
    const dim3 grid(a.nwg, ny, 1), block(PNT);
#define LAUNCH_P(N, MU, ST)                                                                                          \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_kernel<KIND, N, MU, ST>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)PERSIST_LDS);                      \
        DSMI_LAUNCH((rnn_persist_kernel<KIND, N, MU, ST>), grid, block, PERSIST_LDS, s, ev, a);                       \
    } while (0)
    if (a.dbg) { if (npw > 7 || a.nz > 1) return false; LAUNCH_P(7, false, true); return true; }
    if (a.nz > 1) {
        if (npw <= 4) LAUNCH_P(4, true, false);
        else if (npw <= 7) LAUNCH_P(7, true, false);
        else if (npw <= 10) LAUNCH_P(10, true, false);
        else return false;
        return true;
    }
    if (npw <= 2) LAUNCH_P(2, false, false);
    else if (npw <= 4) LAUNCH_P(4, false, false);
    else if (npw <= 7) LAUNCH_P(7, false, false);
    else if (npw <= 10) LAUNCH_P(10, false, false);
    else return false;
    return true;
}

}  // namespace

// H up to 1280 (10 pairs of 16 k per wave: 80 + 80 operand VGPRs); every workgroup of a launch must own a CU, so a layer whose
// two directions do not fit together (H > 1024 on 256 CUs) runs them as two launches, one after the other.
bool rnn_persist_eligible(const RnnGeom& g, int B, int n_cus) {
    if (g.U != PU || (g.H % 8) != 0) return false;
    if (ceil_div(ceil_div(g.nq, 2), PNW) > 10) return false;
    if (ceil_div(B, 32) > PMAXZ) return false;
    return g.nwg <= n_cus;
}

static inline uint16_t f16_bits(_Float16 h) {
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

// w_hh [G*H][H] (torch layout) of one direction -> [wg][pair][plane][lane][8] fp16 terms (hi, lo * 2^11); lane
// (i = gate row, hk) element e holds k = 16*pair + 8*hk + e, the same k the producer of units 8*(2*pair+hk).. publishes.
std::vector<uint16_t> pack_whh_split(const RnnGeom& g, const float* w_hh) {
    const int npair = ceil_div(g.nq, 2);
    std::vector<uint16_t> out((size_t)g.nwg * npair * 2 * 64 * 8, 0);
    const int GU = g.G * g.U;
    for (int w = 0; w < g.nwg; ++w)
        for (int pq = 0; pq < npair; ++pq)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hk = lane >> 5;
                if (i >= GU) continue;
                const int gate = i / g.U, u = i % g.U, unit = w * g.U + u;
                if (unit >= g.H) continue;
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * pq + 8 * hk + e;
                    if (k >= g.H) continue;
                    const float x = w_hh[(size_t)(gate * g.H + unit) * g.H + k];
                    const _Float16 h1 = (_Float16)x;
                    const _Float16 h2 = (_Float16)((x - (float)h1) * kLoScale);
                    const size_t base = (((size_t)w * npair + pq) * 2) * 64 * 8 + (size_t)lane * 8 + e;
                    out[base] = f16_bits(h1); out[base + 512] = f16_bits(h2);
                }
            }
    return out;
}

bool launch_rnn_persist(const RnnPersistLaunch& p, hipStream_t s) {
    PersistArgs a;
    for (int d = 0; d < 2; ++d) { a.whh_sp[d] = p.whh_sp[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack_sp = p.hpack_sp; a.cnt = p.counters; a.err = p.err; a.dbg = p.dbg;
    a.B = p.B; a.T = p.T; a.G = p.g.G; a.H = p.g.H; a.Hs = p.g.Kp; a.npair = ceil_div(p.g.nq, 2); a.Np = p.g.Np; a.nwg = p.g.nwg;
    a.nz = ceil_div(p.B, 32); a.d0 = p.d0; a.nd = p.g.D;
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_kind<DSMI_RNN_GRU>(a, p.ny, s, p.ev);
        case DSMI_RNN_LSTM: return launch_kind<DSMI_RNN_LSTM>(a, p.ny, s, p.ev);
        default: return launch_kind<DSMI_RNN_TANH>(a, p.ny, s, p.ev);
    }
}

}  // namespace dsmi
