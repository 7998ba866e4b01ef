// Persistent recurrent layer, second generation: 16 hidden units per workgroup, 16-clip batch tiles.
//
// Same contract and hand-off protocol as rnn_persist.hip (all T steps of a BatchRNN in one launch, W_hh
// resident in registers, split-fp16 products, counter form of the agent-scope hand-off, bounded spins), with
// the work cut the other way.  The step time of the first-generation kernel grows with the width of a chain:
// measured 2.4 us/step at 13 workgroups per direction, 3.0 at 50, 4.2 at 100, 5.1 at 125
// (tools/exp/persist_vs_width.py) -- every workgroup ingests the whole state of its chain from L2 each step
// and waits for every producer of the chain.  So a chain is made half as wide and half as deep:
//   * a workgroup owns 16 hidden units (G x 16 gate rows = G MFMA row tiles of 16) -> H/16 workgroups per chain;
//   * a chain carries a tile of 16 clips, the MFMA is v_mfma_f32_16x16x32_f16 (16 gate rows x 16 clips x 32 k);
//   * the batch tiles of a 32-clip batch run as separate chains on separate CUs, side by side with the two
//     directions: cfgA, B = 32 -> 2 directions x 2 tiles x 50 workgroups = 200 CUs, each ingesting 50 KB per step
//     from 50 producers instead of 100 KB from 100.
// Batches with more tiles than fit side by side are walked tile after tile by every workgroup (signal of a tile
// deferred behind the next tile's MFMAs, as in rnn_persist.hip).
//
// Layouts (k = hidden unit index, kb = k / 32, kg = (k / 8) % 4, e = k % 8):
//   W_hh split, per direction:  [workgroup][kb][gate][plane 2][lane 64][8 fp16], lane = (unit u = lane & 15, kg = lane >> 4)
//   state, per parity and chain: [kb][plane 2][kg 4][clip j 16][8 fp16]   (one 1-KiB MFMA B operand per (kb, plane))
//   x-projection: geometry U = 16 (make_rnn_geom_u), one workgroup's G x 16 gate columns contiguous.
#include "common.h"
#include "rnn_cell.h"
#include <cstring>
#include <cstdlib>
#include <type_traits>

namespace dsmi {

namespace {

constexpr int QNW = 8;                 // waves per workgroup (K-split) of the full-CU variants
constexpr int QNT = QNW * 64;
constexpr int QNKR4 = 6;               // four-wave variant: k-blocks of W_hh a wave keeps in registers; a seventh sits in LDS
constexpr int QU = 16;                 // hidden units per workgroup
constexpr int QB = 16;                 // clips per batch tile
constexpr int QRP = 20;                // row pitch (words) of the reduce buffer: conflict-free for the MFMA-layout writes
constexpr int QMAXZ = 8;               // batch tiles one workgroup can walk
constexpr size_t Q_LDS = 82 * 1024;    // > half of the CU's LDS: one workgroup per CU
constexpr size_t Q_LDS_HALF = 78 * 1024;   // four-wave variant: at most two workgroups per CU (its 256 registers per wave allow no third)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

// A word of LDS as it is now (a volatile access through a generic pointer is a flat load: vmcnt(0) with it).
__device__ inline int lds_peek(const int* q) {
    int r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(size_t)q) : "memory");
    return r;
}

// Tile-walking kernel: a feeder wave keeps two sets of state operands in registers, a cell wave one set and the cell, so with
// three gates and five k-blocks per wave the last eight of the thirty W_hh fragments live in LDS (1 KiB each per wave, read per
// instance): with all thirty in registers the compiler spills one and reloads it in every instance -- a vector-memory load whose
// in-order return waits for every state request before it.
constexpr int pipe_lds_frags(int ng, int nkw) { return ng == 3 && nkw == 5 ? 8 : 0; }

struct P16Args {
    const uint16_t* whh[2];    // pack_whh16 per direction
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack; unsigned* cnt; unsigned* err;
    int B, T, H, Hs, Np, nwg, nkb;
    int ntiles;                // 16-clip batch tiles
    int pgroups;               // tile groups running side by side (gridDim.y = D * pgroups); a workgroup walks tiles pg, pg + pgroups, ...
    int D;
    unsigned spin_limit;       // polls of one wait before the workgroup raises *err and stops waiting
    int drop_wg, drop_step;    // test hook (DSMI_DEBUG_DROP_SIGNAL): workgroup drop_wg of chain 0 never signals step drop_step (-1: off)
    unsigned long long* dbg;   // diagnostics build only: per-wave accumulated phase times [workgroup][wave][8]
    int skip;                  // timing experiments (DSMI_DEBUG_PIPE_SKIP, tile-walking kernel only; results are garbage):
                               // 1 no state loads, 2 no MFMAs, 4 polls taken as answered, 8 no stores, 16 no cell
};

#define QSTAMP(k)                                                                         \
    do {                                                                                  \
        if (STAMP) {                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                            \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();             \
            tacc[k] += now_ - tlast; tlast = now_;                                        \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
    } while (0)


// KIND: cell type; NKW: compile-time bound of the 32-wide k-blocks one wave owns; NWV: waves per workgroup.
// NWV = 8: one workgroup per CU (two waves per SIMD, <= 256 registers each).
// NWV = 4: HALF a CU per workgroup -- one wave per SIMD, <= 256 registers, so that the recurrent layers of TWO batches in
// flight (two handles, two streams, one gate lane each: api.hip) share every CU: a step is ~3 us of hand-off latency around
// ~1 us of work, and a second, independent chain on the same CU runs in the other one's shadow.  A wave then owns up to 7
// k-blocks of W_hh: six in registers, the seventh in LDS (read per MFMA, 1 KiB per wave instruction).
template <int KIND, int NKW, int NWV, bool STAMP = false>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void rnn_persist16_kernel(P16Args p) {
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    extern __shared__ __attribute__((aligned(16))) float qlds[];
    constexpr int NKR = NWV == 4 ? (NKW < QNKR4 ? NKW : QNKR4) : NKW;   // k-blocks of W_hh in registers; the rest in LDS
    float* red = qlds;                                               // [NWV][NG][16 units][QRP]
    int& s_dead = *reinterpret_cast<int*>(red + NWV * 4 * 16 * QRP);
    float* st_h = red + NWV * 4 * 16 * QRP + 32;                    // [QMAXZ][256] carried state when a workgroup walks several tiles
    float* st_c = st_h + QMAXZ * 256;
    int* st_len = reinterpret_cast<int*>(st_c + QMAXZ * 256);
    u32x4* wlds = reinterpret_cast<u32x4*>(st_len + QMAXZ * 256);    // [NWV][NKW - NKR][NG][2 planes][64 lanes] 16-byte fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;                        // MFMA lane roles: column / row index, k-group / row-group
    const int w = blockIdx.x;
    const int d = blockIdx.y / p.pgroups, pg = blockIdx.y - d * p.pgroups;
    const int nz = (p.ntiles - pg + p.pgroups - 1) / p.pgroups;      // tiles this workgroup walks
    const bool multi = nz > 1;
    const int GU = NG * QU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    if (tid == 0) s_dead = 0;

    // ---- resident operand: this wave's k-blocks of the split W_hh, all gates
    const int kb0 = (v * p.nkb) / NWV, kb1 = ((v + 1) * p.nkb) / NWV;
    f16x8 wv[NKR][NG][2];
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
                    if (i < NKR) wv[i < NKR ? i : 0][g][pl] = __builtin_bit_cast(f16x8, frag);
                    else wl[(((i - NKR) * NG + g) * 2 + pl) * 64] = frag;       // read back by this same lane only
                }
        }
    }
    const size_t hp_par = (size_t)p.D * p.ntiles * p.nkb * 2048;     // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);

    // cell role: threads 0..255 own (unit u = 8 * (tid >> 7) + (tid & 7), clip j = (tid >> 3) & 15): a wave's 2-byte
    // stores of the new state are 128 contiguous bytes of one k-group block
    const int cuh = tid >> 7, ce = tid & 7, cj = (tid >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w * QU + cu;
    const bool cunit_ok = tid < 256 && cunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    int mylen = 0;
    float hprev_own = 0.f, cprev_own = 0.f;
    if (tid < 256) {
        if (multi) {
            for (int z = 0; z < nz; ++z) {
                const int eb = (pg + z * p.pgroups) * QB + cj;
                st_h[z * 256 + tid] = 0.f; st_c[z * 256 + tid] = 0.f;
                st_len[z * 256 + tid] = (cunit_ok && eb < p.B) ? p.lens[eb] : 0;
            }
        } else {
            const int eb = pg * QB + cj;
            mylen = (cunit_ok && eb < p.B) ? p.lens[eb] : 0;
        }
    }
    __syncthreads();

    unsigned* pend = nullptr;          // several tiles: counter of the tile just published, signalled behind the next tile's MFMAs
    bool pend_drop = false;
    for (int s = 0; s < p.T; ++s) {
        const int t = d == 0 ? s : p.T - 1 - s;
        for (int z = 0; z < nz; ++z) {
            const int tile = pg + z * p.pgroups;
            const int b0 = tile * QB;
            const int nb = min(QB, p.B - b0);
            const int eb = b0 + cj;
            const bool eact = cunit_ok && cj < nb;
            const int chain = d * p.ntiles + tile;
            unsigned* cnt = p.cnt + (size_t)chain * p.T * kPersist16CntWords;
            const unsigned hchain = (unsigned)((size_t)chain * p.nkb * 2048);
            if (multi && tid < 256) { mylen = st_len[z * 256 + tid]; hprev_own = st_h[z * 256 + tid]; cprev_own = st_c[z * 256 + tid]; }
            // x-projection operands of this step do not depend on other workgroups: request them first
            float xg[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) xg[g] = 0.f;
            if (eact) {
                const float* xr = p.xp + ((size_t)t * p.B + eb) * p.Np + xcol + cu;
#pragma unroll
                for (int g = 0; g < NG; ++g) xg[g] = xr[g * QU];
            }
            f32x4 acc[NG], acl[NG];        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
            for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            QSTAMP(0);   // loop head + x-projection request
            if (s > 0) {
                // ---- wait until every workgroup of this chain has published h_{s-1} (bounded)
                // the counter of a (chain, step) is sharded (workgroup id % kPersist16Shards), each shard on its own 256-byte
                // line: atomics and polls on one line are served one at a time, ~10 ns each
                if (v == 0 && !s_dead) {
                    unsigned spins = 0;
                    const unsigned* cp = &cnt[(size_t)(s - 1) * kPersist16CntWords + (lane & (kPersist16Shards - 1)) * 64];
                    const unsigned need = (unsigned)((p.nwg + kPersist16Shards - 1 - (lane & (kPersist16Shards - 1))) / kPersist16Shards);   // workgroups w with w % shards == lane
                    while (true) {
                        const unsigned got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                        if (__builtin_amdgcn_ballot_w64(got < need) == 0) break;
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { s_dead = 1; break; }
                        if (spins > p.spin_limit) { atomicExch(p.err, 1u); s_dead = 1; break; }
                    }
                }
                __syncthreads();
                QSTAMP(1);   // waiting for the other workgroups
                // ---- B operand: split h_{s-1} of this wave's k-blocks, sc1 loads only (lane = kg * 16 + clip)
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
                // every state load is issued, in k order, before the first MFMA: without this the compiler sinks the loads of
                // the first k-block into its conditional block, behind all the others, and the first MFMA waits for every load
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    if (kb0 + i < kb1) {
                        f16x8 wa[NG][2];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl)
                                wa[g][pl] = i < NKR ? wv[i < NKR ? i : 0][g][pl]
                                                    : __builtin_bit_cast(f16x8, wl[(((i - NKR) * NG + g) * 2 + pl) * 64]);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][1], hv[i][0], acl[g], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][0], acc[g], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[g][0], hv[i][1], acl[g], 0, 0, 0);
                    }
                }
            }
            QSTAMP(2);   // h load + MFMA chain
            // partial tiles -> LDS: D[row = unit 4 * lg + r][col = clip ln]
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((v * 4 + g) * 16 + 4 * lg + r) * QRP + ln] = acc[g][r] + acl[g][r] * kLoInv;
            if (multi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile's state stores are acknowledged
            __syncthreads();
            if (multi && tid == 0 && pend && !pend_drop) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            QSTAMP(3);   // partial tiles to LDS + barrier (wave skew)
            // ---- K-split reduction (fixed order) + cell + publish, one (unit, clip) pair per thread
            if (tid < 256) {
                float hn = 0.f;
                if (eact) {
                    float hg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < NWV; ++k) sum += red[((k * 4 + g) * 16 + cu) * QRP + cj];
                        hg[g] = sum + bh[g];
                    }
                    hn = rnn_cell<KIND>(xg, hg, hprev_own, cprev_own, t < mylen);
                    hprev_own = hn;
                    if (multi) { st_h[z * 256 + tid] = hn; if (KIND == DSMI_RNN_LSTM) st_c[z * 256 + tid] = cprev_own; }
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
            }
            QSTAMP(4);   // reduction + cell + publish stores issued
            const bool drop = chain == 0 && w == p.drop_wg && s == p.drop_step;
            if (multi) {
                pend = &cnt[(size_t)s * kPersist16CntWords + (w & (kPersist16Shards - 1)) * 64];
                pend_drop = drop;
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own stores
                __syncthreads();
                if (tid == 0 && !drop) __hip_atomic_fetch_add(&cnt[(size_t)s * kPersist16CntWords + (w & (kPersist16Shards - 1)) * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            QSTAMP(5);   // drain + barrier + signal
        }
    }
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * QNW + v) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Several batch tiles per workgroup (more 16-clip tiles than fit side by side): the tile-walking kernel.  A workgroup walks the
// tile instances i = s * nz + z (step s, its z-th tile) with ONE barrier per instance: MFMAs of instance i and partial tiles
// -> barrier -> signal of instance i - 1 (its stores were drained in front of the barrier) -> cell of instance i, and the
// reduce buffer is double-buffered because a fast wave writes instance i + 1's partial tiles while a cell wave still reads
// instance i's.  From four tiles per workgroup on (and where the registers allow: AHEAD) the eight K-split waves have ROLES
// (round 4; DESIGN.md 4 "The tile-walking kernel"): waves 0-3 reduce, run the cell, publish and signal and never poll; waves
// 4-7, the feeders, poll for instance i + 2 behind their MFMAs of instance i and request their part of its state two
// instances ahead.  Workgroups of two or three tiles (the short groups of a launch) run a one-role form: every wave polls for
// instance i + 1 in front of its MFMAs of instance i (a two-tile group: behind its own signal).
template <int KIND, int NKW, bool SKIPS = false>
__global__ __launch_bounds__(QNT) void rnn_persist16_pipe_kernel(P16Args p) {
    const int skip = SKIPS ? p.skip : 0;                             // (timing experiments: a build of its own)
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NF = NKW * NG * 2, NFL = pipe_lds_frags(NG, NKW);  // W_hh fragments (k-block, gate, plane) of a wave; the last NFL in LDS
    constexpr int AHEAD = (NG == 4 && NKW >= 4) ? 1 : 2;             // (an LSTM of four k-blocks per wave has no registers for a second set)
    constexpr int RED = QNW * NG * 16 * QRP;                         // words per reduce buffer
    extern __shared__ __attribute__((aligned(16))) float qlds[];
    float* red0 = qlds;                                              // [2][QNW][NG][16 units][QRP]
    int& s_dead = *reinterpret_cast<int*>(red0 + 2 * RED);
    float* st_h = red0 + 2 * RED + 32;                               // [QMAXZ][256] carried state of the tiles
    float* st_c = st_h + QMAXZ * 256;
    int* st_len = reinterpret_cast<int*>(st_c + QMAXZ * 256);
    u32x4* wlds = reinterpret_cast<u32x4*>(st_len + QMAXZ * 256);    // [QNW][NFL][64 lanes] 16-byte fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x;
    const int d = blockIdx.y / p.pgroups, pg = blockIdx.y - d * p.pgroups;
    const int nz = (p.ntiles - pg + p.pgroups - 1) / p.pgroups;
    const int GU = NG * QU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    if (tid == 0) s_dead = 0;

    const int kb0 = (v * p.nkb) / QNW, kb1 = ((v + 1) * p.nkb) / QNW;
    f16x8 wv[NKW][NG][2];                                            // (the entries that live in LDS are never touched)
    u32x4* wl = wlds + (size_t)v * (NFL > 0 ? NFL : 1) * 64 + lane;  // written and read by this same lane only
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
                    const int f = (i * NG + g) * 2 + pl;
                    if (f < NF - NFL) wv[i][g][pl] = __builtin_bit_cast(f16x8, frag);
                    else wl[(f - (NF - NFL)) * 64] = frag;
                }
        }
    }
    const size_t hp_par = (size_t)p.D * p.ntiles * p.nkb * 2048;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);

    const int cuh = tid >> 7, ce = tid & 7, cj = (tid >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w * QU + cu;
    const bool cunit_ok = tid < 256 && cunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    if (tid < 256)
        for (int z = 0; z < nz; ++z) {
            const int eb = (pg + z * p.pgroups) * QB + cj;
            st_h[z * 256 + tid] = 0.f; st_c[z * 256 + tid] = 0.f;
            st_len[z * 256 + tid] = (cunit_ok && eb < p.B) ? p.lens[eb] : 0;
        }
    __syncthreads();

    const int NI = p.T * nz;                                         // tile instances i = s * nz + z
    auto tile_of = [&](int z) { return pg + z * p.pgroups; };
    auto xp_of = [&](int s, int z) {
        const int t = d == 0 ? s : p.T - 1 - s;
        return p.xp + ((size_t)t * p.B + tile_of(z) * QB + cj) * p.Np + xcol + cu;
    };
    auto active = [&](int z) { return cunit_ok && cj < min(QB, p.B - tile_of(z) * QB); };

    f16x8 hva[NKW][2], hvb[NKW][2];       // the state operands of even / odd instances: requested TWO instances ahead
#pragma unroll
    for (int i = 0; i < NKW; ++i) { hva[i][0] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; hva[i][1] = hva[i][0]; hvb[i][0] = hva[i][0]; hvb[i][1] = hva[i][0]; }
    float xa[NG], xb[NG];                  // x-projection of even / odd instances (AHEAD == 1: of this / the next instance)
#pragma unroll
    for (int g = 0; g < NG; ++g) { xa[g] = 0.f; xb[g] = 0.f; }
    if (active(0)) {
        const float* xr = xp_of(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) xa[g] = xr[g * QU];
    }
    unsigned* pend = nullptr;
    bool pend_drop = false;
    const int shard = lane & (kPersist16Shards - 1);
    const unsigned need = (unsigned)((p.nwg + kPersist16Shards - 1 - shard) / kPersist16Shards);

    // Roles (AHEAD == 2).  Waves 0-3 are the CELL waves: after the barrier of instance i they reduce, run the cell and publish,
    // while waves 4-7, the FEEDERS, are already in instance i + 1.  A feeder polls for instance i + 2 behind its MFMAs of
    // instance i (it would wait for the cell waves at the barrier anyway) and requests its own part of that state; it does not
    // pass the barrier before the poll has answered, so a cell wave knows without polling that the state of instance i + 1 is
    // there when it starts instance i, and requests its part behind its MFMAs of instance i (it has the cell of instance i to
    // cover the landing; a feeder, which would go straight on, has two sets of operands and a whole instance).
    auto state_base = [&](const int si, const int zi) {
        const int ch = d * p.ntiles + tile_of(zi);
        return (unsigned)(((si - 1) & 1) * hp_par) + (unsigned)((size_t)ch * p.nkb * 2048) + (unsigned)lane * 16u;
    };
    auto load_state_k = [&](const unsigned hbase, const int k, f16x8 (&dst)[NKW][2]) __attribute__((always_inline)) {
        if (skip & 1) return;
        const int kb = min(kb0 + k, max(kb1 - 1, kb0));
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            dst[k][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, hbase + (unsigned)(kb * 2 + pl) * 1024u, 0, 16));
    };
    auto load_state = [&](const int si, const int zi, f16x8 (&dst)[NKW][2]) __attribute__((always_inline)) {
        const unsigned hbase = state_base(si, zi);
#pragma unroll
        for (int k = 0; k < NKW; ++k) load_state_k(hbase, k, dst);
    };

    // (the two roles run SEPARATE copies of the loop: in one copy the compiler orders a cell wave's request into a register set
    // behind the feeder branch's request into the same set -- a wait for all but the newest operation, i.e. for the cell's
    // stores, at the top of every instance)
    // STEADY instances (every instance from the second step on, in the two-role form) issue the SAME memory operations on every
    // path -- requests beyond the last instance are clamped to it and never used -- because the compiler's wait for an
    // operand counts the operations issued since on the path that issued fewest: one path without requests turns every wait
    // for an old operand into a wait for everything.
    int sz[2] = {0, 0};
    auto instance = [&](auto role_tag, auto steady_tag, const int i, f16x8 (&hv)[NKW][2], float (&xg)[NG], float (&xn)[NG])
                        __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr int ROLE = decltype(role_tag)::value;              // 0: every wave does everything (AHEAD == 1); 1: cell wave; 2: feeder
        constexpr bool feeder = ROLE == 2;
        const int s = sz[0], z = sz[1];                              // i = s * nz + z (carried: no division per instance)
        const int t = d == 0 ? s : p.T - 1 - s;
        const int tile = tile_of(z);
        const int chain = d * p.ntiles + tile;
        unsigned* cnt = p.cnt + (size_t)chain * p.T * kPersist16CntWords;
        float* red = red0 + (i & 1) * RED;
        const int z1 = z + 1 < nz ? z + 1 : 0, s1 = z + 1 < nz ? s : s + 1;
        constexpr int AH = ROLE == 0 ? 1 : 2;                        // how many instances ahead a polling wave polls and requests
        const int z2 = AH == 1 ? z1 : (z1 + 1 < nz ? z1 + 1 : 0), s2 = AH == 1 ? s1 : (z1 + 1 < nz ? s1 : s1 + 1);
        const bool next_state = i + AH < NI && (STEADY || s2 > 0);
        const int chain2 = d * p.ntiles + tile_of(z2);
        const unsigned* cp = p.cnt + (size_t)chain2 * p.T * kPersist16CntWords + (size_t)(next_state ? s2 - 1 : 0) * kPersist16CntWords + shard * 64;
        unsigned got = need;
        // (a workgroup that walks TWO tiles -- the short group of a launch whose other groups walk three -- polls for what its own
        // previous instance wrote: only behind the barrier and the signal of this instance)
        const bool late_poll = ROLE == 0 && nz < 3;
        // a cell wave has ONE set of state operands (the cell's registers take the place of the second): it requests the k-blocks
        // of instance i + 1 one by one behind the MFMAs that read those of instance i
        const bool cell_ahead = ROLE == 1 && (STEADY || (i + 1 < NI && s1 > 0));
        const bool last = i + 1 >= NI;                               // (requests beyond the last instance are clamped to it)
        const int sx = STEADY && last ? s : s1, zx = STEADY && last ? z : z1;
        const unsigned hbase1 = cell_ahead ? state_base(sx, zx) : 0u;
        if (ROLE != 0) {
            if (ROLE == 1 && (STEADY || i + 1 < NI)) {
                asm volatile("" ::: "memory");                 // (behind the cell's stores: the wait below counts on the order)
                const float* xr = p.xp + ((size_t)(d == 0 ? sx : p.T - 1 - sx) * p.B + min(tile_of(zx) * QB + cj, p.B - 1)) * p.Np + xcol + cu;
#pragma unroll
                for (int g = 0; g < NG; ++g) xn[g] = xr[g * QU];
            }
        } else {
            // one set of state operands: the poll of instance i + 1 goes out in front of the MFMAs of instance i (its producers
            // signalled an iteration or more ago); every wave polls for itself: no barrier between the answer and its loads
            if (next_state && !late_poll && lane < kPersist16Shards) got = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- B(i): multiply (the state of instance i was requested during the previous iteration)
        f32x4 acc[NG], acl[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {
#pragma unroll
            for (int k = 0; k < NKW; ++k) {
                if ((STEADY || s > 0) && !(skip & 2) && kb0 + k < kb1) {
                    f16x8 wf[NG][2];
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            wf[g][pl] = (k * NG + g) * 2 + pl < NF - NFL ? wv[k][g][pl]
                                                                         : __builtin_bit_cast(f16x8, wl[max((k * NG + g) * 2 + pl - (NF - NFL), 0) * 64]);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[g][1], hv[k][0], acl[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[g][0], hv[k][0], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[g][0], hv[k][1], acl[g], 0, 0, 0);
                }
                if (cell_ahead) load_state_k(hbase1, k, hv);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((v * NG + g) * 16 + 4 * lg + r) * QRP + ln] = acc[g][r] + acl[g][r] * kLoInv;
        // ---- the cell waves' stores of instance i-1 are acknowledged (in-order return: everything older than the requests of
        // this instance's top has come back)
        if (cell_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NKW + NG) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- the producers of instance i + AHEAD have signalled -> request its state, this wave's part
        auto poll_and_request = [&](const bool fresh) __attribute__((always_inline)) {
            if (next_state && !lds_peek(&s_dead) && !(skip & 4)) {
                unsigned spins = 0;
                if (fresh) got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                while (__builtin_amdgcn_ballot_w64(got < need) != 0) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { s_dead = 1; break; }
                    if (spins > p.spin_limit) { atomicExch(p.err, 1u); s_dead = 1; break; }
                    got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                }
            }
            const bool beyond = STEADY && i + AH >= NI;
            load_state(beyond ? s : s2, beyond ? z : z2, hv);
        };
        if ((next_state || STEADY) && ROLE != 1 && !late_poll) poll_and_request(ROLE == 2);
        if (ROLE == 0 && i + 1 < NI && active(z1)) {
            const float* xr = xp_of(s1, z1);
#pragma unroll
            for (int g = 0; g < NG; ++g) xn[g] = xr[g * QU];
        }
        __syncthreads();                                      // partial tiles written, everybody's stores of instance i-1 acknowledged
        if (feeder) return;
        if (tid == 0 && pend && !pend_drop) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ROLE == 0 && late_poll && next_state) poll_and_request(true);
        // ---- C(i): reduce + cell + publish
        float hn = 0.f;
        const bool eact = active(z);
        const int eb = tile * QB + cj;
        if (tid < 256) {
            const int mylen = st_len[z * 256 + tid];
            const float hprev_own = st_h[z * 256 + tid];
            float cprev_own = st_c[z * 256 + tid];
            if (eact && !(skip & 16)) {
                float hg[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < QNW; ++k) sum += red[((k * NG + g) * 16 + cu) * QRP + cj];
                    hg[g] = sum + bh[g];
                }
                hn = rnn_cell<KIND, true>(xg, hg, hprev_own, cprev_own, t < mylen);
                st_h[z * 256 + tid] = hn;
                if (KIND == DSMI_RNN_LSTM) st_c[z * 256 + tid] = cprev_own;
                if (!(skip & 8)) p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = hn;
            } else if (cj < min(QB, p.B - tile * QB) && cunit < p.Hs) {
                p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = 0.f;
            }
        }
        if (tid < 256 && !(skip & 8)) {
            const _Float16 h1 = (_Float16)hn;
            const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
            const unsigned off = (unsigned)((s & 1) * hp_par) + (unsigned)((size_t)chain * p.nkb * 2048) + (unsigned)(w >> 1) * 2048u +
                                 (unsigned)(2 * (w & 1) + cuh) * 256u + (unsigned)cj * 16u + (unsigned)ce * 2u;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h1), hrs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h2), hrs, off + 1024u, 0, 16);
        }
        pend = &cnt[(size_t)s * kPersist16CntWords + (w & (kPersist16Shards - 1)) * 64];
        pend_drop = chain == 0 && w == p.drop_wg && s == p.drop_step;
        if (ROLE == 0)
#pragma unroll
            for (int g = 0; g < NG; ++g) xg[g] = xn[g];
    };
    auto step = [&]() { sz[0] = sz[1] + 1 < nz ? sz[0] : sz[0] + 1; sz[1] = sz[1] + 1 < nz ? sz[1] + 1 : 0; };
    // Two instances ahead is the state instance i + 2 - nz wrote, signalled behind the barrier of instance i + 3 - nz: a feeder
    // polls for it IN FRONT of the barrier of instance i, so the two-role form needs four tiles or more (with three it waits
    // for its own workgroup's signal).  Workgroups of one launch may walk different numbers of tiles.
    if (AHEAD == 2 && nz >= 4) {
        auto walk = [&](auto role_tag) __attribute__((always_inline)) {
            const std::false_type first{};
            const std::true_type steady{};
            const int P = min((nz + 1) & ~1, NI);        // the instances of step 0 (and one more where nz is odd) in the general form
            int i = 0;
            constexpr bool one_set = decltype(role_tag)::value == 1;
            auto& hodd = *(one_set ? &hva : &hvb);
            for (; i + 1 < P; i += 2) { instance(role_tag, first, i, hva, xa, xb); step(); instance(role_tag, first, i + 1, hodd, xb, xa); step(); }
            if (i < P) { instance(role_tag, first, i, hva, xa, xb); return; }      // (one step of an odd number of tiles)
            for (; i + 1 < NI; i += 2) { instance(role_tag, steady, i, hva, xa, xb); step(); instance(role_tag, steady, i + 1, hodd, xb, xa); step(); }
            if (i < NI) instance(role_tag, steady, i, hva, xa, xb);
        };
        if (v >= 4) walk(std::integral_constant<int, 2>{});
        else walk(std::integral_constant<int, 1>{});
    } else {
        for (int i = 0; i < NI; ++i) { instance(std::integral_constant<int, 0>{}, std::false_type{}, i, hva, xa, xb); step(); }
    }
}

inline uint16_t q_f16_bits(_Float16 h) {
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

template <int KIND>
bool launch16(const P16Args& a, hipStream_t s, const EvPair& ev, int waves) {
    const dim3 grid(a.nwg, a.D * a.pgroups, 1);
#define LAUNCH_Q(N, W, LDS)                                                                                          \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist16_kernel<KIND, N, W>),                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));                            \
        DSMI_LAUNCH((rnn_persist16_kernel<KIND, N, W>), grid, dim3(W * 64), LDS, s, ev, a);                           \
    } while (0)
    if (waves == 4) {              // half-CU workgroups (see the kernel): one or two tiles per workgroup, no diagnostics build
        const int nkw = ceil_div(a.nkb, 4);
        if (a.dbg || ceil_div(a.ntiles, a.pgroups) > 2) return false;
        if (nkw <= 2) LAUNCH_Q(2, 4, Q_LDS_HALF);
        else if (nkw <= 4) LAUNCH_Q(4, 4, Q_LDS_HALF);
        else if (nkw <= 6 && KIND != DSMI_RNN_LSTM) LAUNCH_Q(6, 4, Q_LDS_HALF);
        else if (nkw <= 7 && KIND != DSMI_RNN_LSTM) LAUNCH_Q(7, 4, Q_LDS_HALF);
        else return false;
        return true;
    }
    const int nkw = ceil_div(a.nkb, QNW);
    const dim3 block(QNT);
    if (a.dbg) {
        if (nkw > 4) return false;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist16_kernel<KIND, 4, QNW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
        hipLaunchKernelGGL((rnn_persist16_kernel<KIND, 4, QNW, true>), grid, block, Q_LDS, s, a);
        return true;
    }
    if (ceil_div(a.ntiles, a.pgroups) > 2) {      // three or more tiles per workgroup: the software-pipelined kernel (with two
                                                   // tiles the next instance's producers were signalled a moment ago: nothing to overlap)
        constexpr int NGK = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
        auto lds_pipe = [](int nkw) {            // two reduce buffers, the dead flag, carried state of up to QMAXZ tiles, W_hh fragments
            return (size_t)2 * QNW * NGK * 16 * QRP * 4 + 128 + (size_t)3 * QMAXZ * 256 * 4 + (size_t)QNW * pipe_lds_frags(NGK, nkw) * 1024;
        };
#define LAUNCH_QP(N)                                                                                                 \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist16_pipe_kernel<KIND, N>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe(N));                      \
        DSMI_LAUNCH((rnn_persist16_pipe_kernel<KIND, N>), grid, block, lds_pipe(N), s, ev, a);                        \
    } while (0)
#ifdef DSMI_EXPERIMENTS
        if (a.skip && KIND == DSMI_RNN_GRU && nkw == 5) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist16_pipe_kernel<DSMI_RNN_GRU, 5, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe(5));
            DSMI_LAUNCH((rnn_persist16_pipe_kernel<DSMI_RNN_GRU, 5, true>), grid, block, lds_pipe(5), s, ev, a);
        } else
#endif
        if (nkw <= 2) LAUNCH_QP(2);
        else if (nkw <= 4) LAUNCH_QP(4);
        else if (nkw <= 5 && KIND != DSMI_RNN_LSTM) LAUNCH_QP(5);
        else return false;
#undef LAUNCH_QP
        return true;
    }
    if (nkw <= 2) LAUNCH_Q(2, QNW, Q_LDS);
    else if (nkw <= 4) LAUNCH_Q(4, QNW, Q_LDS);
    else if (nkw <= 5 && KIND != DSMI_RNN_LSTM) LAUNCH_Q(5, QNW, Q_LDS);
    else return false;
#undef LAUNCH_Q
    return true;
}

}  // namespace

RnnGeom make_rnn_geom_u(int kind, int H, int D, int U) {
    RnnGeom g = make_rnn_geom(kind, H, D);
    g.U = U;
    g.nwg = ceil_div(H, U);
    g.Np = D * g.nwg * g.G * U;
    return g;
}

// H a multiple of 16, at most 4 k-blocks per wave (H <= 1024; 5 for GRU / RNN: H <= 1280), both directions of a
// tile group co-resident, no more tiles per workgroup than the carried-state arrays hold.
bool rnn_persist16_eligible(const RnnGeom& g16, int B, int n_cus, int* pgroups_out) {
    if (g16.U != QU || (g16.H % QU) != 0) return false;
    const int nkb = ceil_div(g16.H, 32), nkw = ceil_div(nkb, QNW);
    if (nkw > (g16.kind == DSMI_RNN_LSTM ? 4 : 5)) return false;
    if (g16.nwg * g16.D > n_cus) return false;
    const int ntiles = ceil_div(B, QB);
    const int pg = std::min(ntiles, n_cus / (g16.nwg * g16.D));
    if (ceil_div(ntiles, pg) > QMAXZ) return false;
    if (pgroups_out) *pgroups_out = pg;
    return true;
}

// The half-CU (four-wave) variant: its register budget holds 6 + 1 k-blocks per wave for three gates (H <= 896), 4 for an
// LSTM (H <= 512); at most two tiles per workgroup; one workgroup slot per CU and lane.
bool rnn_persist16_half_eligible(const RnnGeom& g16, int B, int n_cus, int* pgroups_out) {
    if (g16.U != QU || (g16.H % QU) != 0) return false;
    const int nkw = ceil_div(ceil_div(g16.H, 32), 4);
    if (nkw > (g16.kind == DSMI_RNN_LSTM ? 4 : 7)) return false;
    if (g16.nwg * g16.D > n_cus) return false;
    const int ntiles = ceil_div(B, QB);
    const int pg = std::min(ntiles, n_cus / (g16.nwg * g16.D));
    if (ceil_div(ntiles, pg) > 2) return false;
    if (pgroups_out) *pgroups_out = pg;
    return true;
}

// w_hh [G*H][H] (torch layout) of one direction -> [workgroup][kb][gate][plane][lane][8] fp16 terms (hi, lo * 2^11);
// lane (u = lane & 15, kg = lane >> 4) element e holds W[gate * H + 16 * wg + u][32 * kb + 8 * kg + e].
std::vector<uint16_t> pack_whh16(const RnnGeom& g16, const float* w_hh) {
    const int nkb = ceil_div(g16.H, 32), G = g16.G, H = g16.H;
    std::vector<uint16_t> out((size_t)g16.nwg * nkb * G * 2 * 64 * 8, 0);
    for (int w = 0; w < g16.nwg; ++w)
        for (int kb = 0; kb < nkb; ++kb)
            for (int gate = 0; gate < G; ++gate)
                for (int lane = 0; lane < 64; ++lane) {
                    const int unit = w * QU + (lane & 15);
                    if (unit >= H) continue;
                    for (int e = 0; e < 8; ++e) {
                        const int k = 32 * kb + 8 * (lane >> 4) + e;
                        if (k >= H) continue;
                        const float x = w_hh[(size_t)(gate * H + unit) * H + k];
                        const _Float16 h1 = (_Float16)x;
                        const _Float16 h2 = (_Float16)((x - (float)h1) * kLoScale);
                        const size_t base = ((((size_t)w * nkb + kb) * G + gate) * 2) * 512 + (size_t)lane * 8 + e;
                        out[base] = q_f16_bits(h1); out[base + 512] = q_f16_bits(h2);
                    }
                }
    return out;
}

size_t rnn_persist16_state_halfs(const RnnGeom& g16, int B) {
    return (size_t)2 * g16.D * ceil_div(B, QB) * ceil_div(g16.H, 32) * 1024;
}

bool launch_rnn_persist16(const RnnPersist16Launch& p, hipStream_t s) {
    P16Args a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh16[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack16; a.cnt = p.counters; a.err = p.err;
    a.B = p.B; a.T = p.T; a.H = p.g.H; a.Hs = p.g.Kp; a.Np = p.g.Np; a.nwg = p.g.nwg; a.nkb = ceil_div(p.g.H, 32);
    a.ntiles = ceil_div(p.B, QB); a.pgroups = p.pgroups; a.D = p.g.D; a.dbg = p.dbg;
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step;
    static const int skip_env = [] { const char* e = exp_env("DSMI_DEBUG_PIPE_SKIP"); return e ? std::atoi(e) : 0; }();
    a.skip = skip_env;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch16<DSMI_RNN_GRU>(a, s, p.ev, p.waves);
        case DSMI_RNN_LSTM: return launch16<DSMI_RNN_LSTM>(a, s, p.ev, p.waves);
        default: return launch16<DSMI_RNN_TANH>(a, s, p.ev, p.waves);
    }
}

}  // namespace dsmi
