// Persistent recurrent layer, paired-tile variant: one workgroup carries TWO chains -- the two 16-clip tiles of a
// 32-clip batch for its 16 hidden units of one direction -- in a fixed software pipeline, so that a whole batch needs
// half the CUs (cfgA: 2 directions x 50 workgroups = 100) and the other half of the chip is free for a second batch.
//
// Same arithmetic, state layout, counters and hand-off protocol as rnn_persist16.hip (split-fp16 products on
// v_mfma_f32_16x16x32_f16, sc1 stores / sc1 loads, sharded agent-scope counter per (chain, step), bounded spins).
//
// Why a pipeline and not two workgroups per CU: a step of one chain is ~1.5 us in which the CU's pipes are busy (state
// ingest, MFMAs, K-split reduction, cell) and ~1.6 us of waiting (store drain, signal -> everybody's signal visible, load
// latency).  Two independent half-CU workgroups (rnn_persist16 NWV = 4, one per batch in flight) do not interleave those:
// somewhere in a 50-workgroup chain the two busy parts coincide in every step, and the chain runs at the sum (4.7 us per
// step, 2.3 ms per layer with two batches in flight).  Here the interleaving is the program: waves 0-3 (half A,
// tile 0) and waves 4-7 (half B, tile 1) run the same four-slot step, B two slots behind A, with a workgroup barrier after
// every slot:
//
//      slot        half A                                  half B
//      4s + 0      load h(s-1), MFMAs, partials -> LDS     drain stores(s-1); one wave signals step s-1 as soon as
//                                                          the four waves have drained (LDS counter, no barrier)
//      4s + 1      K-split reduce, cell, publish stores    poll "step s-1 of my chain is complete everywhere"
//      4s + 2      drain stores(s); signal step s          load h(s-1), MFMAs, partials -> LDS
//      4s + 3      poll step s complete                    reduce, cell, publish stores
//
// Every busy slot of one half runs beside a waiting slot of the other, on every CU in the same order, so both chains
// advance in lockstep anti-phase chip-wide.  A signal leaves ~0.45 us into a drain slot and its poll starts ~0.5 us later,
// when most of the hand-off latency has passed.
#include "common.h"
#include "rnn_cell.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace dsmi {

namespace {

constexpr int DNW = 8;                 // waves per workgroup: two halves of four (K-split inside a half)
constexpr int DNT = DNW * 64;
constexpr int DU = 16;                 // hidden units per workgroup
constexpr int DB = 16;                 // clips per batch tile
constexpr int DRP = 20;                // row pitch (words) of the reduce buffers
constexpr int DNKR = 6;                // k-blocks of W_hh a wave keeps in registers; a seventh sits in LDS
constexpr size_t D_LDS = 100 * 1024;   // > half of the CU's LDS (the 8 x 256 registers say "one per CU" as well)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

struct DuoArgs {
    const uint16_t* whh[2];    // pack_whh16 per direction (the 16-unit image of rnn_persist16.hip)
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack; unsigned* cnt; unsigned* err;
    int B, T, H, Hs, Np, nwg, nkb;
    int ntiles, D;
    int pair0, npairs;         // this launch carries tile pairs pair0 .. pair0 + npairs - 1 (gridDim.y = D * npairs)
    unsigned spin_limit;
    int drop_wg, drop_step;
    unsigned long long* dbg;   // diagnostics build only: per wave, time in each of its four slots [0..3] and at the barrier behind it [4..7]
};


// TAIL: the k-blocks do not divide by the half's four waves (H = 800: 25).  Instead of one wave carrying a whole extra block
// (9 more MFMAs and its W_hh fragments in LDS, the slot as long as that wave), every wave takes nkb / 4 full blocks and the
// 1-3 blocks left over are dealt out gate by gate: wave vh takes gate vh % NG of block 4 * (nkb / 4) + vh / NG -- three more
// MFMAs and 8 more registers, no LDS operand.  Offered when that is at most one (block, gate) item per wave.
template <int KIND, int NKW, bool STAMP = false, bool TAIL = false>
__global__ __launch_bounds__(DNT, 2) void rnn_persist_duo_kernel(DuoArgs p) {
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NKR = NKW < DNKR ? NKW : DNKR;
    extern __shared__ __attribute__((aligned(16))) float dlds[];
    float* red_all = dlds;                                           // [2 halves][4 waves][4 gate slots][16 units][DRP]
    int* sync = reinterpret_cast<int*>(red_all + 2 * 4 * 4 * 16 * DRP);   // [0] dead flag, [8 + half] drained-waves counter
    u32x4* wlds = reinterpret_cast<u32x4*>(sync + 32);              // [8 waves][NKW - NKR][NG][2 planes][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hx = v >> 2, vh = v & 3;                               // half (tile of the pair), wave within the half
    const int tidh = tid & 255;
    const int ln = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x;
    const int d = blockIdx.y / p.npairs, pair = p.pair0 + (blockIdx.y - d * p.npairs);
    const int tile = 2 * pair + hx;
    const bool tile_ok = tile < p.ntiles;                            // an odd tile count leaves the last half B idle (barriers only)
    const int GU = NG * DU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    float* red = red_all + hx * (4 * 4 * 16 * DRP);
    if (tid < 32) sync[tid] = 0;

    // ---- resident operand: this wave's k-blocks of the split W_hh, all gates (both halves hold the same image)
    const int kq = p.nkb >> 2;
    const int kb0 = TAIL ? vh * kq : (vh * p.nkb) / 4, kb1 = TAIL ? kb0 + kq : ((vh + 1) * p.nkb) / 4;
    const int xkb = 4 * kq + vh / NG, xgate = vh % NG;               // TAIL: this wave's left-over (block, gate)
    const bool has_x = TAIL && vh < NG * (p.nkb & 3);
    f16x8 wv[NKR][NG][2];
    // TAIL: that item's W_hh fragments [0..1] and, with six full blocks, the low planes of the sixth [2 + gate] live in LDS
    // and are read (by the lane that wrote them) once earlier blocks have released registers: held from the start they push
    // the kernel past 256 registers and the compiler parks a W_hh fragment in scratch, behind the state loads.
    constexpr bool LO5 = TAIL && NKW == 6;
    u32x4* wxl = wlds + (size_t)v * (2 + NG) * 64 + lane;
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
                    if (LO5 && i == 5 && pl == 1) wxl[(2 + g) * 64] = frag;
                    else if (i < NKR) wv[i < NKR ? i : 0][g][pl] = __builtin_bit_cast(f16x8, frag);
                    else wl[(((i - NKR) * NG + g) * 2 + pl) * 64] = frag;       // read back by this same lane only
                }
        }
        if (has_x)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wxl[pl * 64] = wp[(((size_t)xkb * NG + xgate) * 2 + pl) * 64];
    }
    const size_t hp_par = (size_t)p.D * p.ntiles * p.nkb * 2048;     // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);

    // cell role inside the half: thread -> (unit cu = 8 * (tidh >> 7) + (tidh & 7), clip cj = (tidh >> 3) & 15)
    const int cuh = tidh >> 7, ce = tidh & 7, cj = (tidh >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w * DU + cu;
    const int nb = tile_ok ? min(DB, p.B - tile * DB) : 0;
    const int eb = tile * DB + cj;
    const bool cunit_ok = cunit < p.H;
    const bool eact = cunit_ok && cj < nb;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    const int mylen = eact ? p.lens[eb] : 0;
    float hprev_own = 0.f, cprev_own = 0.f;
    const int chain = d * p.ntiles + (tile_ok ? tile : 0);
    unsigned* cnt = p.cnt + (size_t)chain * p.T * kPersist16CntWords;
    const unsigned hchain = (unsigned)((size_t)chain * p.nkb * 2048);
    const unsigned shard = (unsigned)(w & (kPersist16Shards - 1)) * 64u;
    float xg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) xg[g] = 0.f;
    if (eact) {                                          // x-projection of step 0
        const float* xr = p.xp + ((size_t)(d == 0 ? 0 : p.T - 1) * p.B + eb) * p.Np + xcol + cu;
#pragma unroll
        for (int g = 0; g < NG; ++g) xg[g] = xr[g * DU];
    }
    __syncthreads();

    const int nslots = 4 * p.T + 2;
    for (int gs = 0; gs < nslots; ++gs) {
        const int ls = gs - 2 * hx;                                  // this half's own slot counter
        const int s = ls >> 2, k = ls & 3;
        unsigned long long t0_ = 0, c0_ = 0;
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); t0_ = __builtin_amdgcn_s_memrealtime(); c0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if (tile_ok && ls >= 0 && s < p.T) {
            const int t = d == 0 ? s : p.T - 1 - s;
            if (k == 0) {
                // ---- state of the previous step, MFMAs, partial tiles.  (The x-projection operands of this step were requested
                // in slot 2 of the previous one: they come from HBM, and vmcnt retires in order -- requested here they would
                // stand between the state loads and the first MFMA for a full HBM latency: 1.8 us instead of 1.0 in this slot.)
                f32x4 acc[NG], acl[NG];        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (s > 0) {
                    const unsigned hbase = (unsigned)(((s - 1) & 1) * hp_par) + hchain + (unsigned)lane * 16u;
                    f16x8 hv[NKW][2];
#pragma unroll
                    for (int i = 0; i < NKW; ++i) {
                        const int kb = min(kb0 + i, max(kb1 - 1, kb0));
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            hv[i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                hrs, hbase + (unsigned)(kb * 2 + pl) * 1024u, 0, 16));
                    }
                    f16x8 hx_[2] = {};
                    if (has_x)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            hx_[pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                hrs, hbase + (unsigned)(xkb * 2 + pl) * 1024u, 0, 16));
                    __builtin_amdgcn_sched_barrier(0);     // every state load is issued, in k order, before the first MFMA
                    f16x8 wlo5[NG];
#pragma unroll
                    for (int i = 0; i < NKW; ++i) {
                        if (LO5 && i == 3)
#pragma unroll
                            for (int g = 0; g < NG; ++g) wlo5[g] = __builtin_bit_cast(f16x8, wxl[(2 + g) * 64]);
                        if (kb0 + i < kb1) {
                            f16x8 wa[NG][2];
#pragma unroll
                            for (int g = 0; g < NG; ++g)
#pragma unroll
                                for (int pl = 0; pl < 2; ++pl)
                                    wa[g][pl] = (LO5 && i == 5 && pl == 1) ? wlo5[g] : i < NKR ? wv[i < NKR ? i : 0][g][pl]
                                                        : __builtin_bit_cast(f16x8, wl[(((i - NKR) * NG + g) * 2 + pl) * 64]);
#pragma unroll
                            for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], hv[i][0], acl[g], 0, 0, 0);
#pragma unroll
                            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][0], acc[g], 0, 0, 0);
#pragma unroll
                            for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][1], acl[g], 0, 0, 0);
                        }
                        // keep the k-blocks in load order: the scheduler otherwise starts with the LDS-resident block, whose
                        // state operand was requested LAST, and the first MFMA then waits for every load
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (has_x) {
                        // operand read here, into registers the finished blocks have released (the kernel sits at the 256-register
                        // limit: held from the start, these eight spill a W_hh fragment into scratch, behind the state loads)
                        f16x8 wx[2];
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) wx[pl] = __builtin_bit_cast(f16x8, wxl[pl * 64]);
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            if (g == xgate) {
                                acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[1], hx_[0], acl[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[0], hx_[0], acc[g], 0, 0, 0);
                                acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[0], hx_[1], acl[g], 0, 0, 0);
                            }
                    }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[((vh * 4 + g) * 16 + 4 * lg + r) * DRP + ln] = acc[g][r] + acl[g][r] * kLoInv;
            } else if (k == 1) {
                // ---- K-split reduction (fixed order) + cell + publish, one (unit, clip) pair per thread
                float hn = 0.f;
                if (eact) {
                    float hg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) sum += red[((q * 4 + g) * 16 + cu) * DRP + cj];
                        hg[g] = sum + bh[g];
                    }
                    hn = rnn_cell<KIND>(xg, hg, hprev_own, cprev_own, t < mylen);
                    hprev_own = hn;
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = hn;
                } else if (cj < nb && cunit < p.Hs) {
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = 0.f;     // padding units of the last workgroup
                }
                const _Float16 h1 = (_Float16)hn;
                const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
                const unsigned off = (unsigned)((s & 1) * hp_par) + hchain + (unsigned)(w >> 1) * 2048u +
                                     (unsigned)(2 * (w & 1) + cuh) * 256u + (unsigned)cj * 16u + (unsigned)ce * 2u;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h1), hrs, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h2), hrs, off + 1024u, 0, 16);
            } else if (k == 2) {
                // ---- every wave drains its own stores; the half's wave 0 signals as soon as all four have
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&sync[8 + hx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (eact && s + 1 < p.T) {               // x-projection of the NEXT step (consumed in its slot 1), behind the drain
                    const int tn = d == 0 ? s + 1 : p.T - 2 - s;
                    const float* xr = p.xp + ((size_t)tn * p.B + eb) * p.Np + xcol + cu;
#pragma unroll
                    for (int g = 0; g < NG; ++g) xg[g] = xr[g * DU];
                }
                if (vh == 0) {
                    const int want = 4 * (s + 1);
                    while (__hip_atomic_load(&sync[8 + hx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
                    const bool drop = chain == 0 && w == p.drop_wg && s == p.drop_step;
                    if (lane == 0 && !drop)
                        __hip_atomic_fetch_add(&cnt[(size_t)s * kPersist16CntWords + shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (s + 1 < p.T) {
                // ---- wait until every workgroup of this chain has published h_s (bounded); the barrier that ends the slot
                // releases the half's other waves into their loads
                if (vh == 0 && !sync[0]) {
                    unsigned spins = 0;
                    const unsigned* cp = &cnt[(size_t)s * kPersist16CntWords + (lane & (kPersist16Shards - 1)) * 64];
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
            }
        }
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime();
            const unsigned long long c1_ = __builtin_amdgcn_s_memtime();
            if (ls >= 0 && k == 0) tacc[7] += c1_ - c0_;          // shader-clock cycles spent in slot 0 (overwrites the slot-3 barrier column)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t2_ = __builtin_amdgcn_s_memrealtime();
            if (ls >= 0) { tacc[k] += t1_ - t0_; if (k != 3) tacc[4 + k] += t2_ - t1_; }
        } else {
            __syncthreads();
        }
    }
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * DNW + v) * 8;
        for (int q = 0; q < 8; ++q) o[q] = tacc[q];
    }
}

template <int KIND>
bool launch_duo(const DuoArgs& a, hipStream_t s, const EvPair& ev) {
    const int nkw = ceil_div(a.nkb, 4);
    constexpr int NGk = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    const int kq = a.nkb / 4, kr = a.nkb % 4;
    const bool tail = kr > 0 && NGk * kr <= 4 && kq >= 1 && kq <= (KIND == DSMI_RNN_LSTM ? 4 : 6);
    const dim3 grid(a.nwg, a.D * a.npairs, 1), block(DNT);
    if (a.dbg) {
        if (KIND != DSMI_RNN_GRU || nkw != 7) return false;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_duo_kernel<DSMI_RNN_GRU, 7, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_LDS);
        hipLaunchKernelGGL((rnn_persist_duo_kernel<DSMI_RNN_GRU, 7, true>), grid, block, D_LDS, s, a);
        return true;
    }
#define LAUNCH_K(KERNEL, ...)                                                                                        \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<KIND, __VA_ARGS__>),                           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_LDS);                            \
        DSMI_LAUNCH((KERNEL<KIND, __VA_ARGS__>), grid, block, D_LDS, s, ev, a);                                       \
    } while (0)
#define LAUNCH_T(N) LAUNCH_K(rnn_persist_duo_kernel, N, false, true)
#define LAUNCH_D(N) LAUNCH_K(rnn_persist_duo_kernel, N, false, false)
    if (tail) {
        if (kq <= 2) LAUNCH_T(2);
        else if (kq <= 4) LAUNCH_T(4);
        else if constexpr (KIND != DSMI_RNN_LSTM) LAUNCH_T(6);
        return true;
    }
    if (KIND == DSMI_RNN_LSTM) {
        if (nkw <= 2) LAUNCH_D(2);
        else if (nkw <= 4) LAUNCH_D(4);
        else return false;
        return true;
    }
    if (nkw <= 2) LAUNCH_D(2);
    else if (nkw <= 4) LAUNCH_D(4);
    else if (nkw <= 6) LAUNCH_D(6);
    else if (nkw <= 7) LAUNCH_D(7);
    else return false;
#undef LAUNCH_T
#undef LAUNCH_D
#undef LAUNCH_K
    return true;
}

}  // namespace

// At least two tiles (17+ clips), the half-CU register budget (GRU / RNN: H <= 896, LSTM: H <= 512), every tile pair of both
// directions co-resident on `n_cus` CUs (the caller passes one gate lane's CUs first, then the whole device).
int rnn_persist_duo_pairs(const RnnGeom& g16, int B, int n_cus) {
    if (g16.U != DU || (g16.H % DU) != 0) return 0;
    const int nkw = ceil_div(ceil_div(g16.H, 32), 4);
    if (nkw > (g16.kind == DSMI_RNN_LSTM ? 4 : 7)) return 0;
    const int ntiles = ceil_div(B, DB);
    if (ntiles < 2) return 0;
    return std::min((ntiles + 1) / 2, n_cus / (g16.nwg * g16.D));
}

bool rnn_persist_duo_eligible(const RnnGeom& g16, int B, int n_cus) {
    return rnn_persist_duo_pairs(g16, B, n_cus) >= std::max(1, (ceil_div(B, DB) + 1) / 2);
}

bool launch_rnn_persist_duo(const RnnPersist16Launch& p, hipStream_t s) {
    DuoArgs a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh16[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack16; a.cnt = p.counters; a.err = p.err;
    a.B = p.B; a.T = p.T; a.H = p.g.H; a.Hs = p.g.Kp; a.Np = p.g.Np; a.nwg = p.g.nwg; a.nkb = ceil_div(p.g.H, 32);
    a.ntiles = ceil_div(p.B, DB); a.D = p.g.D;
    a.pair0 = p.pair0; a.npairs = p.npairs > 0 ? p.npairs : (a.ntiles + 1) / 2;
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step; a.dbg = p.dbg;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_duo<DSMI_RNN_GRU>(a, s, p.ev);
        case DSMI_RNN_LSTM: return launch_duo<DSMI_RNN_LSTM>(a, s, p.ev);
        default: return launch_duo<DSMI_RNN_TANH>(a, s, p.ev);
    }
}

}  // namespace dsmi
