// Persistent recurrent layer, ring variant: one workgroup owns 32 hidden units of one direction and walks ALL batch tiles
// of its launch round-robin, with the tiles' packed states staged through a two-slot ring in LDS.
//
// Same arithmetic, packed weights, x-projection order, state layout and hand-off protocol as rnn_persist16.hip /
// rnn_persist_duo.hip (split-fp16 products on v_mfma_f32_16x16x32_f16, sc1 stores / sc1 loads, sharded agent-scope counter
// per (chain, step), bounded spins).  What changes is how a CU's time is filled:
//
//   * A chain's step is a latency: cell -> store drain -> signal -> everybody's signal visible -> state loads -> MFMAs,
//     about 4 us, of which the CU works for about 1 us.  The paired-tile kernel fills it with two chains per CU.  Here a
//     workgroup carries every tile of the launch (four 16-clip tiles = two 32-clip batches in one launch; phantom tiles pad
//     shorter launches), so that each chain's hand-off lies under the other tiles' work.
//   * The two halves of the workgroup (waves 0-3 = A, waves 4-7 = B) are two ADJACENT 16-unit groups of the 16-unit geometry
//     (virtual workgroups 2w and 2w + 1) instead of two copies of one group: no weight is held twice, a chain has half as many
//     producers (H / 32), and the 50 KB of a tile's state are brought in ONCE per CU and step -- by LDS-DMA into the ring --
//     for both halves (the paired-tile kernel ingests them once per 16 units).  A 64-clip layer of cfgA takes 50 CUs.
//   * Fixed schedule, one workgroup barrier per slot; item q = (step s, tile j) = 4 s + j, unrolled over the four tiles:
//
//        slot 2q        A: MFMAs of item q  (B operands from ring slot q & 1)         B: signal item q - 2; cell of item q - 1 with the
//                          partial tiles -> LDS; wait for its own requests                DMA requests of item q + 1 in four groups between
//                                                                                         its parts; x-projection request for item q
//        slot 2q + 1    A: cell of item q; x-projection request for item q + 1        B: poll request for item q + 2 (wave 4); MFMAs of item q;
//                                                                                         wait for its own requests; poll answer (spin if not yet)
//
//     so every slot has one half on the matrix pipe and the other on the vector / memory side.  A half issues vector-memory
//     operations only in its cell slots (publish stores, x-projection requests, B's DMA requests) and waits for all of them at
//     the end of its next MFMA slot: that one wait is the store drain in front of the signal, the landing of the DMA before
//     the barrier that releases the readers, and the arrival of the next cell's x-projection.
//   * The DMA and x-projection requests are inline assembly: the compiler orders every LDS read behind an LDS-DMA it knows of
//     (vmcnt(0)), which would put the landing latency in front of the cell's reduce-buffer reads.
//   * Publish: the eight lanes of a clip hand their (hi | lo << 16) words to the clip's first lane by DPP row shifts; that lane
//     stores 16 bytes of each plane with sc1 -- one whole 128-byte line per plane and wave instruction (2-byte sc1 stores are one
//     fabric write each).  Stores carry no branch: a lane with nothing to store has an offset beyond the buffer's range.
//   Measured forms and what bounds the kernel: DESIGN.md 4 "The ring kernel", profiles/r04_ring_experiments.txt.
#include "common.h"
#include "rnn_cell.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace dsmi {

namespace {

constexpr int RNT = 512;               // 8 waves: two halves of four (K-split inside a half)
constexpr int RU = 16;                 // hidden units per half
constexpr int RB = 16;                 // clips per batch tile
constexpr int RRP = 20;                // row pitch (words) of the reduce buffers
constexpr int RMINT = 4;               // schedule length in tiles = tiles a window walks at most: fewer real tiles are padded with phantom
                                       // ones (the poll of an item comes 2 * NT - 3 slots after its M slot, the signal 3: NT >= 4)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

struct RingArgs {
    const uint16_t* whh[2];    // pack_whh16 per direction
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack; unsigned* cnt; unsigned* err;
    int B, T, H, Hs, Np, nwg16, nkb;
    int ntiles, D;             // 16-clip tiles of the whole batch (state and counter layout), directions
    int tile0, ntw, tile_end;  // window z (blockIdx.z) walks tiles tile0 + z * ntw .. + ntw - 1, below tile_end
    unsigned spin_limit;
    int drop_wg, drop_step;
    unsigned* tickets;         // [windows][2 directions] zeroed before the launch -> directions by XCD half (null: by blockIdx): rnn_persist_ring4.hip
    int skip;                  // timing experiments only (DSMI_DEBUG_RING_SKIP; results are garbage): 1 no state DMA, 2 no MFMAs, 4 no polls,
                               // 8 no x-projection requests, 16 no output / publish stores, 32 no wave priorities, 64 every slot ~500 cycles longer
                               // (128: nothing -- the build with the switches as it stands)
    unsigned long long* dbg;   // diagnostics build only: per wave, 100 MHz ticks: [0] M work, [1] M-end waits, [2] C work, [3] barrier
                               // behind M, [4] barrier behind C, [5] poll spin; [7] slots; shader cycles: [6] M work, [8] M head (to the
                               // first operands' arrival), [9] MFMA loop, [10] partial tiles -> LDS, [11] DMA requests, [12] reduce + cell,
                               // [13] publish + x-projection requests
};

#define RSTAMP(k)                                                                          \
    do {                                                                                   \
        if (STAMP) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                             \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
            if ((threadIdx.x & 63) == 0) tacc[k] += now_ - clast;                          \
            clast = now_;                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one 1-KiB piece of packed state, global -> LDS, bypassing this CU's L1 (sc1): lane l brings bytes [16 l, 16 l + 16)
__device__ __forceinline__ void ring_dma(const void* gbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" :: "s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}

// The schedule is RMINT = 4 tiles long and unrolled over the tiles: which ring slot, which neighbour items a slot signals,
// polls, requests and prefetches for are compile-time, and what depends on the tile alone (chain, counters, row blocks)
// is computed once; a slot's scalar work is a handful of additions.
template <int KIND, int NKW, bool STAMP = false, bool SKIPS = false>
__global__ __launch_bounds__(RNT, 2) void rnn_persist_ring_kernel(RingArgs p) {
    const int skipf = SKIPS ? p.skip : 0;          // timing experiments only (DSMI_DEBUG_RING_SKIP): the production build carries none of the tests
    unsigned long long clast = 0;
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NT = RMINT;
    extern __shared__ __attribute__((aligned(16))) unsigned char rlds[];
    const int sbytes = p.nkb * 2048;
    unsigned char* sbuf = rlds;                                                   // [2 ring slots][nkb][2 planes][1024]
    float* red_all = reinterpret_cast<float*>(rlds + 2 * sbytes);                 // [2 halves][4 waves][NG][16 units][RRP]
    float* st_h = red_all + 2 * 4 * NG * 16 * RRP;                               // [NT][512] own previous state per (tile, thread)
    float* st_c = st_h + NT * RNT;                                                // LSTM only: [NT][512]
    float* xgl = st_c + (KIND == DSMI_RNN_LSTM ? NT * RNT : 0);                    // [2 halves][NG][256] x-projection of the half's next cell item
    int* st_len = reinterpret_cast<int*>(xgl + 2 * NG * 256);                     // [NT][16]
    int* sync = st_len + NT * 16;                                                 // [0] dead flag (a hand-off wait timed out: stop waiting)
    // diagnostics build: the accumulated stamps live in LDS (sixteen 8-byte words per wave) -- in registers they cost the kernel 32
    // of the 256 it is built around, and a stamped build that spills measures the spills
    unsigned long long* tacc = reinterpret_cast<unsigned long long*>(sync + 32) + (STAMP ? (threadIdx.x >> 6) * 16 : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hx = v >> 2, vh = v & 3;
    const int tidh = tid & 255;
    const int ln = lane & 15, lg = lane >> 4;
    // (direction, unit group) by XCD half, as in the four-wave form (rnn_persist_ring4.hip: four L2s fetch a tile's state, not eight)
    int w32 = blockIdx.x, d = blockIdx.y;
    if (p.tickets && p.D == 2) {
        const int nw = (p.nwg16 + 1) >> 1;
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned* tk = p.tickets + 2 * blockIdx.z;
            unsigned want = (xcc >> 2) & 1u;
            unsigned t = atomicAdd(&tk[want], 1u);
            if (t >= (unsigned)nw) { want ^= 1u; t = atomicAdd(&tk[want], 1u); }
            sync[30] = (int)want; sync[31] = (int)t;
        }
        __syncthreads();
        d = __builtin_amdgcn_readfirstlane(sync[30]);
        w32 = __builtin_amdgcn_readfirstlane(sync[31]);
        __syncthreads();         // (sync[] is zeroed below)
    }
    const int tile0 = p.tile0 + (int)blockIdx.z * p.ntw;
    const int nt = min(min(p.ntw, p.tile_end - tile0), NT);
    const int w16 = 2 * w32 + hx;
    const bool half_ok = w16 < p.nwg16;
    const int nwg32 = (p.nwg16 + 1) >> 1;
    const int GU = NG * RU;
    float* red = red_all + hx * (4 * NG * 16 * RRP);
    for (int i = tid; i < NT * RNT; i += RNT) { st_h[i] = 0.f; if (KIND == DSMI_RNN_LSTM) st_c[i] = 0.f; }
    if (tid < 32) sync[tid] = 0;
    if (STAMP && lane < 16) tacc[lane] = 0;
    if (tid < NT * 16) {
        const int tl = tid >> 4, eb = (tile0 + tl) * RB + (tid & 15);
        st_len[tid] = (tl < nt && eb < p.B) ? p.lens[eb] : 0;
    }

    // ---- resident operand: this wave's k-blocks of its half's split W_hh, all gates
    const int kb0 = (vh * p.nkb) / 4, kb1 = ((vh + 1) * p.nkb) / 4;
    f16x8 wv[NKW][NG][2];
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh[d]) + ((size_t)(half_ok ? w16 : 0) * p.nkb) * (NG * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int kb = min(kb0 + i, max(kb1 - 1, kb0));
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wv[i][g][pl] = __builtin_bit_cast(f16x8, wp[(((size_t)kb * NG + g) * 2 + pl) * 64]);
        }
    }
    const unsigned hp_par = (unsigned)((size_t)p.D * p.ntiles * p.nkb * 2048);     // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);
    float* outd = p.out[d];
    asm volatile("" : "+s"(outd));          // held in registers: indexed by d, the compiler would re-load it from the kernel arguments per slot
    const unsigned lds_sbuf = (unsigned)(size_t)sbuf;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned char* sbr = sbuf + lane16 + kb0 * 2048;       // this lane's fragment of its wave's first k-block, ring slot 0

    // ---- what depends on the tile alone is affine in the tile index (uniform): base of tile 0 + J * stride, J compile-time
    const int chain0 = d * p.ntiles + tile0;
    unsigned hch0 = (unsigned)chain0 * (unsigned)(p.nkb * 2048), hchs_ = (unsigned)(p.nkb * 2048);      // state bytes of a chain
    unsigned cnt0 = (unsigned)chain0 * (unsigned)p.T * kPersist16CntWords, cnts_ = (unsigned)p.T * kPersist16CntWords;   // counter words
    unsigned orow0 = (unsigned)tile0 * RB * p.Hs, orows_ = (unsigned)(RB * p.Hs);      // elements inside a step's [B][Hs] block
    unsigned xrow0 = (unsigned)tile0 * RB * p.Np, xrows_ = (unsigned)(RB * p.Np);
unsigned hchs = hchs_, cnts = cnts_, orows = orows_, xrows = xrows_;
#define TOK(J) ((J) < nt)

    // ---- cell role inside the half: thread -> (unit cu = 8 * (tidh >> 7) + (tidh & 7), clip cj = (tidh >> 3) & 15); everything a
    // thread needs per item is a uniform base plus one of these per-thread constants
    const int cuh = tidh >> 7, ce = tidh & 7, cj = (tidh >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w16 * RU + cu;
    const bool cunit_ok = half_ok && cunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    const int nb_last = p.B - (tile0 + nt - 1) * RB;                                   // clips of the window's last real tile (may exceed 16)
    unsigned actbits = 0;                                                              // bit j: this thread's (unit, clip) exists in tile j
#pragma unroll
    for (int j = 0; j < NT; ++j) actbits |= (unsigned)(cunit_ok && j < nt && (j < nt - 1 || cj < nb_last)) << j;
    // Stores are unconditional instructions: a lane that has nothing to store carries an offset beyond the buffer's range and the
    // hardware drops it (no branch in the cell).  OOR lies beyond every buffer here (sizes checked below 2 GiB), also after + 1024.
    constexpr unsigned OOR = 0x80000000u;
    unsigned o_by = (unsigned)(cj * p.Hs + (half_ok ? cunit : 0)) * 4u;          // byte offset inside an out row block [16 clips][Hs]
    const unsigned xcol = half_ok ? (unsigned)((d * p.nwg16 + w16) * GU + cu) : 0u;
    unsigned x_by = ((unsigned)cj * p.Np + xcol) * 4u;                          // byte offset inside an x-projection row block
    unsigned x_by_last = ((unsigned)min(cj, nb_last - 1) * p.Np + xcol) * 4u;   // ... clamped to the last tile's clips
    const float* redr = red + cu * RRP + cj;
    unsigned pub_off = (ce == 0 && half_ok) ? (unsigned)(w16 >> 1) * 2048u + (unsigned)(2 * (w16 & 1) + cuh) * 256u + (unsigned)cj * 16u : OOR;   // hi plane; lo at + 1024
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)outd, 0, (int)((size_t)p.T * p.B * p.Hs * 4), 0x00020000);
    const unsigned shard = (unsigned)(w32 & (kPersist16Shards - 1)) * 64u;
    const unsigned need = (unsigned)((nwg32 + kPersist16Shards - 1 - (lane & (kPersist16Shards - 1))) / kPersist16Shards);
    const unsigned lds_xg = (unsigned)(size_t)xgl + (unsigned)((hx * NG * 256 + vh * 64) * 4);
    const float* xgr = xgl + hx * NG * 256 + tidh;
    unsigned pollv = 0;             // B's wave 0: the counter shard this lane read at the start of its MFMA slot

    // x-projection operands of a cell item of tile J, from the step whose rows start at xstep: by LDS-DMA, 4 bytes per lane, into
    // this thread's own words of xgl -- no register is held across the MFMA slot in between, and the compiler, which does not
    // see the requests, waits for nothing
    auto xg_request = [&](auto jc, const float* xstep) {
        constexpr int J = decltype(jc)::value;
        const float* row = xstep + (xrow0 + min(J, nt - 1) * xrows);       // (a phantom tile's request is clamped into the batch)
        const unsigned by = J >= nt - 1 ? x_by_last : x_by;
        if (skipf & 8) return;
        const unsigned lx = lds_xg;         // (a local copy: an asm operand inside a generic lambda does not capture by itself)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(lx + g * 1024), "v"(by + g * RU * 4), "s"(row) : "memory");
    };

    // MFMAs of an item of tile J at step s (B operands from ring slot J & 1), partial tiles -> LDS
    auto mfma_item = [&](auto jc, int s) {
        constexpr int J = decltype(jc)::value;
        if (!TOK(J)) return;
        const bool no_mfma = skipf & 2;
        f32x4 acc[NG], acl[NG];        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (s > 0 && !no_mfma) {
            const unsigned char* sb = sbr + (J & 1) * sbytes;
            f16x8 bc[2], bn[2];
            bc[0] = *reinterpret_cast<const f16x8*>(sb);
            bc[1] = *reinterpret_cast<const f16x8*>(sb + 1024);
            if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RSTAMP(8); }
#pragma unroll
            for (int i = 0; i < NKW; ++i) {
                if (i + 1 < NKW) {          // (a wave with NKW - 1 blocks reads its neighbour's first one here and does not use it)
                    bn[0] = *reinterpret_cast<const f16x8*>(sb + (i + 1) * 2048);
                    bn[1] = *reinterpret_cast<const f16x8*>(sb + (i + 1) * 2048 + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);     // the next block's operands are requested BEFORE this block's MFMAs
                if (i + 1 < NKW || kb0 + i < kb1) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][1], bc[0], acl[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], bc[0], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], bc[1], acl[g], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bc[0] = bn[0]; bc[1] = bn[1];
            }
            RSTAMP(9);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((vh * NG + g) * 16 + 4 * lg + r) * RRP + ln] = acc[g][r] + acl[g][r] * kLoInv;
        if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RSTAMP(10); }
    };

    // B's DMA requests for an item of tile J whose chain's previous step lies at parity offset `par`: each wave brings the k-blocks
    // it will multiply, into ring slot J & 1, in four groups (G = 0..3) placed between the parts of the cell -- the texture path
    // takes a 1-KiB piece per ~80 cycles, and a wave that asks faster stands at the request instead of working.  One base (M0, scalar
    // address) serves two k-blocks: the instruction offset applies to the global AND the LDS address (tools/exp/dma_off_probe.hip).
    auto dma_group = [&](auto jc, auto gc, bool on, unsigned par) {
        constexpr int J = decltype(jc)::value, G = decltype(gc)::value;
        constexpr int i = 2 * G;
        if (i >= NKW || !on || !TOK(J) || (skipf & 1)) return;
        const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(p.hpack) + (par + hch0 + J * hchs + (unsigned)(kb0 + i) * 2048u);
        const unsigned ldst = lds_sbuf + (unsigned)((J & 1) * sbytes) + (unsigned)(kb0 + i) * 2048u;
        const unsigned l16 = lane16;
        if (i + 1 < NKW || kb0 + i < kb1)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 sc1"
                         :: "s"(ldst), "v"(l16), "s"(gsrc) : "memory");
        if (i + 1 < NKW && (i + 2 < NKW || kb0 + i + 1 < kb1))
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048 sc1\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 sc1"
                         :: "s"(ldst), "v"(l16), "s"(gsrc) : "memory");
    };

    // Cell of an item of tile J: K-split reduction (fixed order), cell, own state, output row, publish -- one (unit, clip) pair per
    // thread.  t: time index; osoff: byte offset of the step's output rows; parw: parity offset the new state is written at.  DJ / don / dpar:
    // the item B requests meanwhile (dma_group).  The slot is a chain of latencies in ONE wave per SIMD (the partner multiplies), so
    // it is kept short: every LDS read goes out first, and the publish needs no LDS -- the eight lanes of a clip hand their
    // (hi | lo << 16) words to the clip's first lane by DPP row shifts, which stores both planes' 16 bytes.
    auto cell_item = [&](auto jc, int t, unsigned osoff, unsigned parw, auto djc, bool don, unsigned dpar, auto xjc, const float* xptr, bool xon) {
        constexpr int J = decltype(jc)::value;
        constexpr std::integral_constant<int, 0> G0; constexpr std::integral_constant<int, 1> G1;
        constexpr std::integral_constant<int, 2> G2; constexpr std::integral_constant<int, 3> G3;
        if (!TOK(J)) { if (xon) xg_request(xjc, xptr); dma_group(djc, G0, don, dpar); dma_group(djc, G1, don, dpar); dma_group(djc, G2, don, dpar); dma_group(djc, G3, don, dpar); return; }
        float rv[NG][4], xg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[g][k] = redr[(k * NG + g) * 16 * RRP];
            xg[g] = xgr[g * 256];
        }
        const int mylen = st_len[J * 16 + cj];
        const float hprev = st_h[J * RNT + tid];
        float cprev = KIND == DSMI_RNN_LSTM ? st_c[J * RNT + tid] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        // the x-projection of this half's NEXT cell item: from HBM, the slowest of what the end of the next slot waits for -- requested
        // as early as xgl's present values have been read (the reads above went out first; LDS serves a wave in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (xon) xg_request(xjc, xptr);
        dma_group(djc, G0, don, dpar);
        dma_group(djc, G1, don, dpar);
        RSTAMP(11);
        __builtin_amdgcn_sched_barrier(0);
        float hg[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) hg[g] = ((rv[g][0] + rv[g][1]) + rv[g][2]) + rv[g][3] + bh[g];
        const bool act = (actbits >> J) & 1u;
        float hn = rnn_cell<KIND, true>(xg, hg, hprev, cprev, t < mylen);
        hn = act ? hn : 0.f;
        st_h[J * RNT + tid] = hn;
        if (KIND == DSMI_RNN_LSTM) st_c[J * RNT + tid] = cprev;
        const _Float16 h1 = (_Float16)hn;
        const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
        const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16);
        unsigned u[8];
        u[0] = pk;
        u[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x101, 0xF, 0xF, false);      // row_shl:n: lane i receives lane i + n's word
        u[2] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x102, 0xF, 0xF, false);
        u[3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x103, 0xF, 0xF, false);
        u[4] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x104, 0xF, 0xF, false);
        u[5] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x105, 0xF, 0xF, false);
        u[6] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x106, 0xF, 0xF, false);
        u[7] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x107, 0xF, 0xF, false);
        u32x4 phi, plo;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            phi[m] = __builtin_amdgcn_perm(u[2 * m + 1], u[2 * m], 0x05040100u);     // low halves: units 2m, 2m + 1 of the hi plane
            plo[m] = __builtin_amdgcn_perm(u[2 * m + 1], u[2 * m], 0x07060302u);     // high halves: the lo plane
        }
        RSTAMP(12);
        __builtin_amdgcn_sched_barrier(0);
        dma_group(djc, G2, don, dpar);
        __builtin_amdgcn_sched_barrier(0);
        if (!(skipf & 16)) {
            __builtin_amdgcn_raw_buffer_store_b128(phi, hrs, pub_off, parw + hch0 + J * hchs, 16);
            __builtin_amdgcn_raw_buffer_store_b128(plo, hrs, pub_off + 1024u, parw + hch0 + J * hchs, 16);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), ors, act ? o_by : OOR, osoff + (orow0 + J * orows) * 4u, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        dma_group(djc, G3, don, dpar);
    };

    // end of an MFMA slot: everything this wave requested in its last cell slot -- publish stores (drained), B's DMA (landed),
    // x-projection (arrived)
    auto m_end_wait = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);      // the same wait where the compiler sees it: nothing of this wave's is in flight
    };
    // B, at the start of its cell slot: step `ss` of tile J's chain is published by this workgroup -- both halves' stores of that item
    // were drained before the barrier that opened this slot (A's a slot earlier)
    auto signal_item = [&](auto jc, int ss, bool on) {
        constexpr int J = decltype(jc)::value;
        if (vh == 0 && on && TOK(J)) {
            const bool drop = d == 0 && tile0 + J == 0 && w32 == p.drop_wg && ss == p.drop_step;
            if (lane == 0 && !drop)
                __hip_atomic_fetch_add(p.cnt + (cnt0 + J * cnts + (unsigned)ss * kPersist16CntWords + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // B's wave 0: has step `sp` of tile J's chain been published by every workgroup?  Asked at the start of the MFMA slot (the raw
    // word only, at ONE place per slot: a select there, or a second request site, makes the compiler wait for the load, and with it
    // for every request of the wave's last cell slot, in front of the MFMAs), answered behind it.
    auto poll_issue = [&](auto jc, int sp, bool on) {
        constexpr int J = decltype(jc)::value;
        if (on && TOK(J) && vh == 0 && lane < kPersist16Shards && !(skipf & 4))
            pollv = __hip_atomic_load(p.cnt + (cnt0 + J * cnts + (unsigned)sp * kPersist16CntWords + lane * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto poll_finish = [&](auto jc, int sp, bool on) {
        constexpr int J = decltype(jc)::value;
        if (on && TOK(J) && vh == 0 && !sync[0] && !(skipf & 4)) {
            unsigned long long tp_ = 0;
            if (STAMP) tp_ = __builtin_amdgcn_s_memrealtime();
            const unsigned* cp = p.cnt + (cnt0 + J * cnts + (unsigned)sp * kPersist16CntWords + (lane & (kPersist16Shards - 1)) * 64);
            unsigned spins = 0;
            unsigned got = lane < kPersist16Shards ? pollv : need;
            while (__builtin_amdgcn_ballot_w64(got < need) != 0) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { sync[0] = 1; break; }
                if (spins > p.spin_limit) { atomicExch(p.err, 1u); sync[0] = 1; break; }
                got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
            }
            if (STAMP && lane == 0) tacc[5] += __builtin_amdgcn_s_memrealtime() - tp_;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // (free here: every poll has been consumed) no load is pending at the loop's back edge
    };

    if (hx == 0) xg_request(std::integral_constant<int, 0>{}, p.xp + (size_t)(d == 0 ? 0 : p.T - 1) * p.B * p.Np);
    // through the builtin, so that the compiler knows the prologue's loads (W_hh, biases, lengths) have arrived: told by inline
    // assembly it would wait for them at their first use INSIDE the loop, where vmcnt(0) also waits for that slot's requests
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring_barrier();

    // Stamps (diagnostics build): [0] M work, [1] M-end waits, [2] C work, [3] barrier behind M, [4] barrier behind C in 100 MHz ticks
    unsigned long long tm0 = 0, tm1 = 0;
#define RT_BEGIN() do { if (STAMP) { __builtin_amdgcn_sched_barrier(0); tm0 = __builtin_amdgcn_s_memrealtime(); clast = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define RT_MARK(k) do { if (STAMP) { __builtin_amdgcn_sched_barrier(0); tm1 = __builtin_amdgcn_s_memrealtime(); if ((threadIdx.x & 63) == 0) tacc[k] += tm1 - tm0; tm0 = tm1; __builtin_amdgcn_sched_barrier(0); } } while (0)

    const size_t xstride = (size_t)p.B * p.Np;
    const unsigned ostride_by = (unsigned)((size_t)p.B * p.Hs * 4);
    for (int s = 0; s < p.T; ++s) {
        const int t = d == 0 ? s : p.T - 1 - s;
        const unsigned osoff = (unsigned)t * ostride_by;              // this step's output rows (bytes)
        const float* xstep = p.xp + (size_t)t * xstride;
        const unsigned parw = (unsigned)(s & 1) * hp_par;            // parity offset this step's cells write h_s at
        const unsigned parr = hp_par - parw;                          // ... and its MFMAs read h_(s-1) from (= where step s + 1 writes)
        const bool more = s + 1 < p.T;
        // opaque per step: what is derived from these (a dozen addresses per unrolled tile and half) is recomputed where it is
        // used -- an addition -- instead of being hoisted out of the loop into registers the kernel does not have
        asm volatile("" : "+s"(hch0), "+s"(cnt0), "+s"(orow0), "+s"(xrow0), "+s"(hchs), "+s"(cnts), "+s"(orows), "+s"(xrows));
        asm volatile("" : "+v"(o_by), "+v"(x_by), "+v"(x_by_last), "+v"(pub_off));
        const int tn = d == 0 ? s + 1 : p.T - 2 - s;
        const float* xnext = p.xp + (size_t)(more ? tn : t) * xstride;
        auto slots = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr std::integral_constant<int, (J + 1) % NT> JN;          // next item's tile
            constexpr std::integral_constant<int, (J + NT - 1) % NT> JP;     // previous item's tile
            constexpr std::integral_constant<int, (J + 2) % NT> JQ;          // tile of the item two ahead
            // ---------------- even slot 2q: A multiplies item (s, J); B finishes item q - 1 and requests item q + 1's state
            RT_BEGIN();
            // the cell slot is a short chain of dependent vector and LDS instructions, the MFMA slot a long stream that only needs
            // the matrix pipe kept fed: the cell wave goes first wherever both want the SIMD's issue port
            if (!(skipf & 32)) { if (hx == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
            if (skipf & 64) __builtin_amdgcn_s_sleep(8);       // (timing experiments: every slot ~500 cycles longer)
            if (hx == 0) {
                mfma_item(jc, s);
                RT_MARK(0);
                m_end_wait();
                RT_MARK(1);
            } else {
                // item q - 2, whose stores B drained at the end of the slot before this one: (s, J - 2), or (s - 1, J + 2)
                if (J >= 2) signal_item(JQ, s, true); else signal_item(JQ, s - 1, s >= 1);
                // item q + 1 = (s, J + 1) reads h_(s-1) [needs s >= 1], or (s + 1, 0) reads h_s
                const bool don = J + 1 < NT ? s >= 1 : more;
                const unsigned dpar = J + 1 < NT ? parr : parw;
                // (B's next cell item is q = (s, J): its x-projection is requested inside this cell)
                if (J >= 1) cell_item(JP, t, osoff, parw, JN, don, dpar, jc, xstep, true);
                else if (s >= 1) cell_item(JP, d == 0 ? t - 1 : t + 1, d == 0 ? osoff - ostride_by : osoff + ostride_by, parr, JN, don, dpar, jc, xstep, true);
                else { constexpr std::integral_constant<int, 0> G0; constexpr std::integral_constant<int, 1> G1;
                       constexpr std::integral_constant<int, 2> G2; constexpr std::integral_constant<int, 3> G3;
                       xg_request(jc, xstep);
                       dma_group(JN, G0, don, dpar); dma_group(JN, G1, don, dpar); dma_group(JN, G2, don, dpar); dma_group(JN, G3, don, dpar); }
                RSTAMP(13);
                RT_MARK(2);
            }
            ring_barrier();
            RT_MARK(3 + hx);
            // ---------------- odd slot 2q + 1: A finishes item q; B multiplies it and polls for item q + 2
            if (!(skipf & 32)) { if (hx == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
            if (skipf & 64) __builtin_amdgcn_s_sleep(8);
            if (hx == 0) {
                cell_item(jc, t, osoff, parw, jc, false, 0u, JN, J + 1 < NT ? xstep : xnext, J + 1 < NT || more);
                RSTAMP(13);
                RT_MARK(2);
            } else {
                // item q + 2 = (s, J + 2): its chain's step s - 1 [s >= 1]; or (s + 1, J - 2): its chain's step s
                const bool pon = J + 2 < NT ? s >= 1 : more;
                const int psp = J + 2 < NT ? s - 1 : s;
                poll_issue(JQ, psp, pon);
                mfma_item(jc, s);
                RT_MARK(0);
                m_end_wait();
                poll_finish(JQ, psp, pon);
                RT_MARK(1);
            }
            ring_barrier();
            RT_MARK(4 - hx);
        };
        slots(std::integral_constant<int, 0>{});
        slots(std::integral_constant<int, 1>{});
        slots(std::integral_constant<int, 2>{});
        slots(std::integral_constant<int, 3>{});
        if (STAMP && lane == 0) tacc[7] += 1;
    }
    // ---------------- last slot: B finishes item (T - 1, NT - 1)
    if (hx == 1) {
        const int t = d == 0 ? p.T - 1 : 0;
        constexpr std::integral_constant<int, NT - 1> JL;
        cell_item(JL, t, (unsigned)t * ostride_by, (unsigned)((p.T - 1) & 1) * hp_par, JL, false, 0u, JL, p.xp, false);
    }
    ring_barrier();
#undef TOK
#undef RT_BEGIN
#undef RT_MARK
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + v) * 16;
        for (int k = 0; k < 16; ++k) o[k] = tacc[k];
    }
}

size_t ring_lds_bytes(int kind, int nkb) {
    const int NG = kind == DSMI_RNN_GRU ? 3 : (kind == DSMI_RNN_LSTM ? 4 : 1);
    return (size_t)2 * nkb * 2048 + (size_t)2 * 4 * NG * 16 * RRP * 4 + (size_t)RMINT * RNT * 4 * (kind == DSMI_RNN_LSTM ? 2 : 1) +
           (size_t)2 * NG * 256 * 4 + RMINT * 16 * 4 + 32 * 4 + 8 * 16 * 8;
}

template <int KIND>
bool launch_ring(const RingArgs& a, hipStream_t s, const EvPair& ev) {
    const int nkw = ceil_div(a.nkb, 4);
    const size_t lds = ring_lds_bytes(KIND, a.nkb);
    const dim3 grid((a.nwg16 + 1) / 2, a.D, ceil_div(a.tile_end - a.tile0, a.ntw)), block(RNT);
#define LAUNCH_R(N, ST)                                                                                              \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_ring_kernel<KIND, N, ST>),                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        DSMI_LAUNCH((rnn_persist_ring_kernel<KIND, N, ST>), grid, block, lds, s, ev, a);                              \
    } while (0)
    if (a.dbg) {
        if constexpr (KIND == DSMI_RNN_GRU) { if (nkw == 7) { LAUNCH_R(7, true); return true; } }
        return false;
    }
#ifdef DSMI_EXPERIMENTS
    if (a.skip) {           // timing experiments (DSMI_DEBUG_RING_SKIP): cfgA's shape only
        if constexpr (KIND == DSMI_RNN_GRU) {
            if (nkw == 7) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist_ring_kernel<KIND, 7, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                DSMI_LAUNCH((rnn_persist_ring_kernel<KIND, 7, false, true>), grid, block, lds, s, ev, a);
                return true;
            }
        }
        return false;
    }
#endif
    // NKW exact: every wave owns NKW or NKW - 1 k-blocks, so only the last block of the unrolled loops is conditional
    switch (nkw) {
        case 1: LAUNCH_R(1, false); break;
        case 2: LAUNCH_R(2, false); break;
        case 3: LAUNCH_R(3, false); break;
        case 4: LAUNCH_R(4, false); break;
        default:
            if constexpr (KIND == DSMI_RNN_LSTM) return false;
            else {
                if (nkw == 5) LAUNCH_R(5, false);
                else if (nkw == 6) LAUNCH_R(6, false);
                else if (nkw == 7) LAUNCH_R(7, false);
                else return false;
            }
    }
#undef LAUNCH_R
    return true;
}

}  // namespace

// Tiles one launch of the ring kernel can walk for this shape on `n_cus` CUs (0: not this shape): the 16-unit geometry,
// W_hh of a half in its four waves' registers (GRU / RNN: H <= 896, LSTM: H <= 512), ring + reduce buffers within the CU's LDS,
// both directions co-resident.
int rnn_persist_ring_tiles(const RnnGeom& g16, int B, int n_cus) {
    if (g16.U != RU || (g16.H % RU) != 0) return 0;
    const int nkb = ceil_div(g16.H, 32);
    const int nkw = ceil_div(nkb, 4);
    if (nkw > (g16.kind == DSMI_RNN_LSTM ? 4 : 7)) return 0;
    if (ring_lds_bytes(g16.kind, nkb) > 160 * 1024) return 0;
    if (((g16.nwg + 1) / 2) * g16.D > n_cus) return 0;
    if ((size_t)g16.D * ceil_div(B, RB) * nkb * 2048 * 2 >= (1ull << 31)) return 0;      // packed state below 2 GiB (store offsets, see OOR)
    return std::min(ceil_div(B, RB), RMINT);
}

int rnn_persist_ring_cus(const RnnGeom& g16) { return ((g16.nwg + 1) / 2) * g16.D; }

bool launch_rnn_persist_ring(const RnnPersist16Launch& p, hipStream_t s) {
    RingArgs a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh16[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack16; a.cnt = p.counters; a.err = p.err;
    a.B = p.B; a.T = p.T; a.H = p.g.H; a.Hs = p.g.Kp; a.Np = p.g.Np; a.nwg16 = p.g.nwg; a.nkb = ceil_div(p.g.H, 32);
    a.ntiles = ceil_div(p.B, RB); a.D = p.g.D;
    a.tile0 = p.tile0; a.ntw = p.ntw > 0 ? p.ntw : a.ntiles - p.tile0;
    a.tile_end = std::min(a.ntiles, a.tile0 + a.ntw * std::max(p.nwin, 1));
    if (a.ntw < 1 || a.ntw > RMINT || a.tile_end <= a.tile0) return false;
    if ((size_t)p.T * p.B * p.g.Kp * 4 >= (1ull << 31)) return false;          // a direction's output rows below 2 GiB (store offsets, see OOR)
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step; a.dbg = p.dbg; a.tickets = p.tickets;
    static const int skip = exp_env("DSMI_DEBUG_RING_SKIP") ? std::atoi(exp_env("DSMI_DEBUG_RING_SKIP")) : 0;
    a.skip = skip;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_ring<DSMI_RNN_GRU>(a, s, p.ev);
        case DSMI_RNN_LSTM: return launch_ring<DSMI_RNN_LSTM>(a, s, p.ev);
        default: return launch_ring<DSMI_RNN_TANH>(a, s, p.ev);
    }
}

}  // namespace dsmi
