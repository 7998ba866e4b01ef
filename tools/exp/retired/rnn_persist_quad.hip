// Persistent recurrent layer, four chains per workgroup (OPT-IN, DSMI_PERSIST_QUAD=1; kept as the record of the experiment):
// one workgroup carries BOTH directions of TWO 16-clip tiles for its 16 hidden units, so that a 32-clip batch of a
// bidirectional layer needs a quarter of the CUs the one-chain kernel takes (cfgA: 50 workgroups) and four batches'
// recurrent layers fit on the chip side by side.
//
// Measured (cfgA, 32 x 10 s, MI355X): 5.8 us per step = 2.89 ms per layer on 50 CUs -- the fewest CU-microseconds per step
// of the four variants (290, against 380 for the paired-tile kernel on 100 CUs and 540 for the one-chain kernel on 200) --
// but the whole pipeline is SLOWER with it: 11.7 ms per batch with four batches in flight against 8.5 ms with two batches
// on the paired-tile kernel.  Every interval below ends up as long as its MFMA side (state-load latency + MFMAs + nine LDS
// operand reads, ~1.4 us), and with four streams the dense kernels are left ~60-100 CUs (GEMM 0.39 -> 1.0 ms) while a
// persistent workgroup, which needs a WHOLE free CU, waits behind their many small ones.
//
// Same arithmetic, state layout, counters and hand-off protocol as rnn_persist16.hip / rnn_persist_duo.hip (split-fp16
// products on v_mfma_f32_16x16x32_f16, sc1 stores / sc1 loads, sharded agent-scope counter per (chain, step), bounded
// spins); what changes is who waits for whom.  A chain's step is ~1 us of work for the CU (state ingest + MFMAs, K-split
// reduction + cell) and ~2 us of waiting (stores drain, signal -> everybody's signal visible, load latency).  Here a half
// of the workgroup (waves 0-3: forward direction, waves 4-7: backward; each holds its direction's W_hh) alternates between
// its two tiles, so one tile's waiting is the other tile's work, and the second half runs the same program one interval
// behind, so that the MFMA interval of one half coincides with the cell interval of the other on every CU:
//
//      interval     half A (forward)                               half B (backward)
//      4s + 0       M(tile 0, s)                                   C(tile 1, s-1)
//      4s + 1       C(tile 0, s)                                   M(tile 0, s)
//      4s + 2       M(tile 1, s)                                   C(tile 0, s)
//      4s + 3       C(tile 1, s)                                   M(tile 1, s)
//
//   M(c, s): the state loads of (c, s) have landed and -- vmcnt retires in order -- the other tile's stores of the interval
//            before have drained: one wave signals that step; MFMAs; partial tiles -> LDS; the other tile's x-projection
//            operands for its next cell are requested (consumed two intervals later).
//   C(c, s): K-split reduction, cell, publish stores of (c, s); then wait (bounded) until the OTHER tile's previous step is
//            complete everywhere -- it was signalled an interval ago -- and request its state loads for the next interval.
//
// A workgroup barrier ends every interval: all CUs run the same phase pattern, which is what keeps 50-workgroup chains
// from running at the sum of everybody's busy parts (rnn_persist_duo.hip, header).
#include "common.h"
#include "rnn_cell.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace dsmi {

namespace {

constexpr int QNW = 8;                 // waves per workgroup: two halves of four (K-split inside a half)
constexpr int QNT = QNW * 64;
constexpr int QU = 16;                 // hidden units per workgroup
constexpr int QB = 16;                 // clips per batch tile
constexpr int QRP = 20;                // row pitch (words) of the reduce buffers
constexpr int QNKR = 6;                // k-blocks of W_hh a wave keeps in registers; a seventh sits in LDS
constexpr size_t Q_LDS = 132 * 1024;   // > half of the CU's LDS (the 8 x 256 registers say "one per CU" as well)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

struct QuadArgs {
    const uint16_t* whh[2];    // pack_whh16 per direction (the 16-unit image of rnn_persist16.hip)
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack; unsigned* cnt; unsigned* err;
    int B, T, H, Hs, Np, nwg, nkb;
    int ntiles;
    unsigned spin_limit;
    int drop_wg, drop_step;
};


// TAIL: see rnn_persist_duo.hip -- the k-blocks left over by the four-way split are dealt out gate by gate.
template <int KIND, int NKW, bool TAIL>
__global__ __launch_bounds__(QNT, 2) void rnn_persist_quad_kernel(QuadArgs p) {
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NKR = NKW < QNKR ? NKW : QNKR;
    extern __shared__ __attribute__((aligned(16))) float qlds[];
    float* red_all = qlds;                                           // [2 halves][4 waves][4 gate slots][16 units][QRP]
    int* sync = reinterpret_cast<int*>(red_all + 2 * 4 * 4 * 16 * QRP);   // [0] dead flag, [8 + half] drained waves, [16 + half] polls passed
    u32x4* wlds = reinterpret_cast<u32x4*>(sync + 32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hx = v >> 2, vh = v & 3;                               // half = direction, wave within the half
    const int d = hx;
    const int tidh = tid & 255;
    const int ln = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x, pair = blockIdx.y;
    const int GU = NG * QU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    float* red = red_all + hx * (4 * 4 * 16 * QRP);
    if (tid < 32) sync[tid] = 0;

    // ---- resident operand: this wave's k-blocks of its direction's split W_hh, all gates
    const int kq = p.nkb >> 2;
    const int kb0 = TAIL ? vh * kq : (vh * p.nkb) / 4, kb1 = TAIL ? kb0 + kq : ((vh + 1) * p.nkb) / 4;
    const int xkb = 4 * kq + vh / NG, xgate = vh % NG;               // TAIL: this wave's left-over (block, gate)
    const bool has_x = TAIL && vh < NG * (p.nkb & 3);
    f16x8 wv[NKR][NG][2];
    // register budget (two tiles' cell state per thread on top of rnn_persist_duo.hip's): with six full blocks the LOW planes of
    // the last NLO of them live in LDS, each read one block ahead of its MFMAs by the lane that wrote it
    constexpr int NLO = TAIL && NKW == 6 ? 3 : 0;
    constexpr int LO0 = NKW - NLO;                                   // first block whose low planes are in LDS
    u32x4* wxl = wlds + (size_t)v * (2 + 3 * NG) * 64 + lane;        // [0..1] the left-over item, [2 + (i - LO0) * NG + gate] low planes
    u32x4* wl = wlds + (size_t)v * (NKW - NKR) * NG * 2 * 64 + lane;
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh[d]) + ((size_t)w * p.nkb) * (NG * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int kb = min(kb0 + i, max(kb1 - 1, kb0));
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const u32x4 frag = wp[(((size_t)kb * NG + g) * 2 + pl) * 64];
                    if (NLO && i >= LO0 && pl == 1) wxl[(2 + (i - LO0) * NG + g) * 64] = frag;
                    else if (i < NKR) wv[i < NKR ? i : 0][g][pl] = __builtin_bit_cast(f16x8, frag);
                    else wl[(((i - NKR) * NG + g) * 2 + pl) * 64] = frag;       // read back by this same lane only
                }
        }
        if (has_x)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wxl[pl * 64] = wp[(((size_t)xkb * NG + xgate) * 2 + pl) * 64];
    }
    const size_t hp_par = (size_t)2 * p.ntiles * p.nkb * 2048;       // bytes per parity (both directions)
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);

    // cell role inside the half: thread -> (unit cu = 8 * (tidh >> 7) + (tidh & 7), clip cj = (tidh >> 3) & 15), both tiles
    const int cuh = tidh >> 7, ce = tidh & 7, cj = (tidh >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w * QU + cu;
    const bool cunit_ok = cunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    const unsigned shard = (unsigned)(w & (kPersist16Shards - 1)) * 64u;
    const bool ok1 = 2 * pair + 1 < p.ntiles;                        // an odd tile count leaves the last pair's second tile empty
    int eb[2], mylen[2];
    bool eact[2], epad[2];
    float hprev[2] = {0.f, 0.f}, cprev[2] = {0.f, 0.f};
    float xg[2][NG];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int tile = 2 * pair + c;
        const int nb = tile < p.ntiles ? min(QB, p.B - tile * QB) : 0;
        eb[c] = tile * QB + cj;
        eact[c] = cunit_ok && cj < nb;
        epad[c] = !eact[c] && cj < nb && cunit < p.Hs;               // padding units of the last workgroup: zeros
        mylen[c] = eact[c] ? p.lens[eb[c]] : 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) xg[c][g] = 0.f;
    }
    if (eact[0]) {                                       // x-projection of (tile 0, step 0); tile 1's is requested in M(0, 0)
        const float* xr = p.xp + ((size_t)(d == 0 ? 0 : p.T - 1) * p.B + eb[0]) * p.Np + xcol + cu;
#pragma unroll
        for (int g = 0; g < NG; ++g) xg[0][g] = xr[g * QU];
    }
    // chain ids / offsets are wave-uniform: kept as functions of c so that they live in scalar registers
    auto chain_of = [&](int c) { return d * p.ntiles + min(2 * pair + c, p.ntiles - 1); };
    auto cnt_of = [&](int c) { return p.cnt + (size_t)chain_of(c) * p.T * kPersist16CntWords; };
    auto hchain_of = [&](int c) { return (unsigned)((size_t)chain_of(c) * p.nkb * 2048); };

    f16x8 hv[NKW][2];
    f16x8 hxv[2] = {};
    int ndrain = 0, npoll = 0;                                       // LDS tickets handed out so far (per half)
    // The barrier that ends an interval must not drain the vector-memory queue (the state request of the next interval and
    // the x-projection operands are meant to be in flight across it): __syncthreads() carries a release fence that hipcc
    // lowers to s_waitcnt vmcnt(0) in front of s_barrier; only the LDS traffic has to be ordered here.
    auto interval_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    __syncthreads();
    if (hx == 1) interval_barrier();                                 // half B runs one interval behind

    // ---- M(c, s)
    auto phase_m = [&](auto cc, int s) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const bool ok = c == 0 || ok1;
        // the state loads requested in the interval before have landed, and so have the other tile's publish stores, which
        // were issued ahead of them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ++ndrain;
        const int sp = c == 0 ? s - 1 : s;                           // the step the other tile stored in the interval before
        const bool sig = sp >= 0 && sp + 1 < p.T && (c == 1 || ok1);
        if (lane == 0) __hip_atomic_fetch_add(&sync[8 + hx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (vh == 0 && sig) {
            while (__hip_atomic_load(&sync[8 + hx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * ndrain) __builtin_amdgcn_s_sleep(1);
            const bool drop = chain_of(c ^ 1) == 0 && w == p.drop_wg && sp == p.drop_step;
            if (lane == 0 && !drop)
                __hip_atomic_fetch_add(&cnt_of(c ^ 1)[(size_t)sp * kPersist16CntWords + shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ok) {
            f32x4 acc[NG], acl[NG];        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
            for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (s > 0) {
                f16x8 wlo[NLO ? NLO : 1][NG];
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    if (NLO && i + 1 >= LO0 && i + 1 < NKW)
#pragma unroll
                        for (int g = 0; g < NG; ++g) wlo[NLO ? i + 1 - LO0 : 0][g] = __builtin_bit_cast(f16x8, wxl[(2 + (i + 1 - LO0) * NG + g) * 64]);
                    if (kb0 + i < kb1) {
                        f16x8 wa[NG][2];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl)
                                wa[g][pl] = (NLO && i >= LO0 && pl == 1) ? wlo[NLO ? i - LO0 : 0][g] : i < NKR ? wv[i < NKR ? i : 0][g][pl]
                                                    : __builtin_bit_cast(f16x8, wl[(((i - NKR) * NG + g) * 2 + pl) * 64]);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], hv[i][0], acl[g], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][0], acc[g], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][1], acl[g], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // blocks in order: the LDS-resident operands are read when registers have been released
                }
                if (has_x) {
                    f16x8 wx[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wx[pl] = __builtin_bit_cast(f16x8, wxl[pl * 64]);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        if (g == xgate) {
                            acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[1], hxv[0], acl[g], 0, 0, 0);
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[0], hxv[0], acc[g], 0, 0, 0);
                            acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[0], hxv[1], acl[g], 0, 0, 0);
                        }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((vh * 4 + g) * 16 + 4 * lg + r) * QRP + ln] = acc[g][r] + acl[g][r] * kLoInv;
        }
        // x-projection operands of the OTHER tile's next cell (two intervals from now): they come from HBM, and requested
        // any later they would stand in front of that interval's state loads in the in-order return queue
        const int sn = c == 0 ? s : s + 1;
        if (sn < p.T && eact[c ^ 1]) {
            const int tn = d == 0 ? sn : p.T - 1 - sn;
            const float* xr = p.xp + ((size_t)tn * p.B + eb[c ^ 1]) * p.Np + xcol + cu;
#pragma unroll
            for (int g = 0; g < NG; ++g) xg[c ^ 1][g] = xr[g * QU];
        }
        interval_barrier();
    };

    // ---- C(c, s)
    auto phase_c = [&](auto cc, int s) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const bool ok = c == 0 || ok1;
        // ---- the other tile's turn comes next: its previous step must be complete everywhere (signalled one interval ago)
        const int sn = c == 0 ? s : s + 1;                           // the step M(c ^ 1, sn) will compute
        const bool okn = (c == 1 || ok1) && sn < p.T && sn > 0;
        if (okn) {
            ++npoll;
            if (vh == 0) {
                if (!sync[0]) {
                    unsigned spins = 0;
                    const unsigned* cp = &cnt_of(c ^ 1)[(size_t)(sn - 1) * kPersist16CntWords + (lane & (kPersist16Shards - 1)) * 64];
                    const unsigned need = (unsigned)((p.nwg + kPersist16Shards - 1 - (lane & (kPersist16Shards - 1))) / kPersist16Shards);
                    while (true) {
                        const unsigned got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                        if (__builtin_amdgcn_ballot_w64(got < need) == 0) break;
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { sync[0] = 1; break; }
                        if (spins > p.spin_limit) { atomicExch(p.err, 1u); sync[0] = 1; break; }
                    }
                }
                if (lane == 0) __hip_atomic_store(&sync[16 + hx], npoll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (__hip_atomic_load(&sync[16 + hx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < npoll) __builtin_amdgcn_s_sleep(1);
            }
            // ... and its state is requested now, an interval ahead of the MFMAs that consume it
            const unsigned hbase = (unsigned)(((sn - 1) & 1) * hp_par) + hchain_of(c ^ 1) + (unsigned)lane * 16u;
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                const int kb = min(kb0 + i, max(kb1 - 1, kb0));
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    hv[i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, hbase + (unsigned)(kb * 2 + pl) * 1024u, 0, 16));
            }
            if (has_x)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    hxv[pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, hbase + (unsigned)(xkb * 2 + pl) * 1024u, 0, 16));
        }
        if (ok) {
            const int t = d == 0 ? s : p.T - 1 - s;
            float hn = 0.f;
            if (eact[c]) {
                float hg[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sum += red[((q * 4 + g) * 16 + cu) * QRP + cj];
                    hg[g] = sum + bh[g];
                }
                hn = rnn_cell<KIND>(xg[c], hg, hprev[c], cprev[c], t < mylen[c]);
                hprev[c] = hn;
            }
            if (eact[c] || epad[c]) p.out[d][((size_t)t * p.B + eb[c]) * p.Hs + cunit] = hn;
            const _Float16 h1 = (_Float16)hn;
            const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
            const unsigned off = (unsigned)((s & 1) * hp_par) + hchain_of(c) + (unsigned)(w >> 1) * 2048u +
                                 (unsigned)(2 * (w & 1) + cuh) * 256u + (unsigned)cj * 16u + (unsigned)ce * 2u;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h1), hrs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h2), hrs, off + 1024u, 0, 16);
        }
        interval_barrier();
    };

    for (int s = 0; s < p.T; ++s) {
        phase_m(std::integral_constant<int, 0>{}, s);
        phase_c(std::integral_constant<int, 0>{}, s);
        phase_m(std::integral_constant<int, 1>{}, s);
        phase_c(std::integral_constant<int, 1>{}, s);
    }
    if (hx == 0) interval_barrier();                                 // half A's share of the barrier half B still owes
}

template <int KIND>
bool launch_quad(const QuadArgs& a, hipStream_t s, const EvPair& ev) {
    const int nkw = ceil_div(a.nkb, 4);
    constexpr int NGk = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    const int kq = a.nkb / 4, kr = a.nkb % 4;
    const bool tail = kr > 0 && NGk * kr <= 4 && kq >= 1 && kq <= (KIND == DSMI_RNN_LSTM ? 4 : 6);
    const dim3 grid(a.nwg, (a.ntiles + 1) / 2, 1), block(QNT);
#define LAUNCH_Q(N, TL)                                                                                              \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_quad_kernel<KIND, N, TL>),                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);                            \
        DSMI_LAUNCH((rnn_persist_quad_kernel<KIND, N, TL>), grid, block, Q_LDS, s, ev, a);                            \
    } while (0)
    if (tail) {
        if (kq <= 2) LAUNCH_Q(2, true);
        else if (kq <= 4) LAUNCH_Q(4, true);
        else if constexpr (KIND != DSMI_RNN_LSTM) LAUNCH_Q(6, true);
        return true;
    }
    if (KIND == DSMI_RNN_LSTM) {
        if (nkw <= 2) LAUNCH_Q(2, false);
        else if (nkw <= 4) LAUNCH_Q(4, false);
        else return false;
        return true;
    }
    if (nkw <= 2) LAUNCH_Q(2, false);
    else if (nkw <= 4) LAUNCH_Q(4, false);
    else if (nkw <= 6) LAUNCH_Q(6, false);
    else if (nkw <= 7) LAUNCH_Q(7, false);
    else return false;
#undef LAUNCH_Q
    return true;
}

}  // namespace

// Bidirectional layers with at least two tiles (17+ clips), the half-CU register budget (GRU / RNN: H <= 896, LSTM:
// H <= 512), every tile pair co-resident on `n_cus` CUs (the caller passes one gate slot's CUs).
bool rnn_persist_quad_eligible(const RnnGeom& g16, int B, int n_cus) {
    if (g16.U != QU || (g16.H % QU) != 0 || g16.D != 2) return false;
    const int nkw = ceil_div(ceil_div(g16.H, 32), 4);
    if (nkw > (g16.kind == DSMI_RNN_LSTM ? 4 : 7)) return false;
    const int ntiles = ceil_div(B, QB);
    if (ntiles < 2) return false;
    return g16.nwg * ((ntiles + 1) / 2) <= n_cus;
}

bool launch_rnn_persist_quad(const RnnPersist16Launch& p, hipStream_t s) {
    QuadArgs a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh16[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack16; a.cnt = p.counters; a.err = p.err;
    a.B = p.B; a.T = p.T; a.H = p.g.H; a.Hs = p.g.Kp; a.Np = p.g.Np; a.nwg = p.g.nwg; a.nkb = ceil_div(p.g.H, 32);
    a.ntiles = ceil_div(p.B, QB);
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_quad<DSMI_RNN_GRU>(a, s, p.ev);
        case DSMI_RNN_LSTM: return launch_quad<DSMI_RNN_LSTM>(a, s, p.ev);
        default: return launch_quad<DSMI_RNN_TANH>(a, s, p.ev);
    }
}

}  // namespace dsmi
