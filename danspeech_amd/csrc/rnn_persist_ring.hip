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
//   * Fixed schedule, one workgroup barrier per slot; item q = (step s, tile j) = s * NT + j:
//
//        slot 2q        A: MFMAs of item q  (B operands from ring slot q & 1)         B: cell of item q - 1, DMA requests of item q + 1
//        slot 2q + 1    A: cell of item q                                             B: MFMAs of item q, poll for item q + 2,
//                                                                                        signal item q - 1
//
//     so every slot has one half on the matrix pipe and the other on the vector / memory side.  A half issues vector-memory
//     operations only in its cell slots (publish stores, x-projection loads, B's DMA requests) and waits for all of them at
//     the end of its next MFMA slot: that one wait is the store drain in front of the signal, the landing of the DMA before
//     the barrier that releases the readers, and the arrival of the next cell's x-projection.
//   * The DMA requests are inline assembly: the compiler orders every LDS read behind an LDS-DMA it knows of (vmcnt(0)),
//     which would put the landing latency in front of the cell's reduce-buffer reads.
//   * Publish: a wave's new state values go through 256 bytes of LDS into 16-byte sc1 stores, one whole 128-byte line per
//     plane and wave instruction (2-byte sc1 stores are one fabric write each).
#include "common.h"
#include "rnn_cell.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace dsmi {

namespace {

constexpr int RNT = 512;               // 8 waves: two halves of four (K-split inside a half)
constexpr int RU = 16;                 // hidden units per half
constexpr int RB = 16;                 // clips per batch tile
constexpr int RRP = 20;                // row pitch (words) of the reduce buffers
constexpr int RMAXT = 8;               // tiles one launch can walk
constexpr int RMINT = 4;               // schedule length in tiles: fewer real tiles are padded with phantom ones (the poll of an
                                       // item comes 2 * NT - 3 slots after its M slot, the signal 3: NT >= 4)

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
            tacc[k] += now_ - clast; clast = now_;                                         \
            __builtin_amdgcn_sched_barrier(0);                                             \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one 1-KiB piece of packed state, global -> LDS, bypassing this CU's L1 (sc1): lane l brings bytes [16 l, 16 l + 16)
__device__ __forceinline__ void ring_dma(const void* gbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" :: "s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}

template <int KIND, int NKW, bool STAMP = false>
__global__ __launch_bounds__(RNT, 2) void rnn_persist_ring_kernel(RingArgs p) {
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char rlds[];
    const int sbytes = p.nkb * 2048;
    unsigned char* sbuf = rlds;                                                   // [2 ring slots][nkb][2 planes][1024]
    float* red_all = reinterpret_cast<float*>(rlds + 2 * sbytes);                 // [2 halves][4 waves][NG][16 units][RRP]
    float* st_h = red_all + 2 * 4 * NG * 16 * RRP;                               // [RMAXT][512] own previous state per (tile, thread)
    float* st_c = st_h + RMAXT * RNT;                                             // LSTM only: [RMAXT][512]
    unsigned short* stg = reinterpret_cast<unsigned short*>(st_c + (KIND == DSMI_RNN_LSTM ? RMAXT * RNT : 0));   // [8 waves][128]
    float* xgl = reinterpret_cast<float*>(stg + 8 * 128);                         // [2 halves][NG][256] x-projection of the half's next cell item
    int* st_len = reinterpret_cast<int*>(xgl + 2 * NG * 256);                     // [RMAXT][16]
    int* sync = st_len + RMAXT * 16;                                              // [0] dead flag, [8] drained-waves counter
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hx = v >> 2, vh = v & 3;
    const int tidh = tid & 255;
    const int ln = lane & 15, lg = lane >> 4;
    const int w32 = blockIdx.x, d = blockIdx.y;
    const int tile0 = p.tile0 + (int)blockIdx.z * p.ntw;
    const int nt = min(p.ntw, p.tile_end - tile0);
    const int w16 = 2 * w32 + hx;
    const bool half_ok = w16 < p.nwg16;
    const int nwg32 = (p.nwg16 + 1) >> 1;
    const int GU = NG * RU;
    const size_t xcol = (size_t)d * p.nwg16 * GU + (size_t)w16 * GU;
    float* red = red_all + hx * (4 * NG * 16 * RRP);
    for (int i = tid; i < RMAXT * RNT; i += RNT) { st_h[i] = 0.f; if (KIND == DSMI_RNN_LSTM) st_c[i] = 0.f; }
    if (tid < 32) sync[tid] = 0;
    if (tid < RMAXT * 16) {
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
    const size_t hp_par = (size_t)p.D * p.ntiles * p.nkb * 2048;     // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);
    const unsigned lds_sbuf = (unsigned)(size_t)sbuf;

    // cell role inside the half: thread -> (unit cu = 8 * (tidh >> 7) + (tidh & 7), clip cj = (tidh >> 3) & 15)
    const int cuh = tidh >> 7, ce = tidh & 7, cj = (tidh >> 3) & 15;
    const int cu = 8 * cuh + ce;
    const int cunit = w16 * RU + cu;
    const bool cunit_ok = half_ok && cunit < p.H;
    float bh[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = cunit_ok ? p.bhh[d][g * p.H + cunit] : 0.f;
    const unsigned shard = (unsigned)(w32 & (kPersist16Shards - 1)) * 64u;
    const int nte = max(nt, RMINT);
    const int NQ = p.T * nte;

    // x-projection operands of a cell item (step s, tile j), clamped so that the requests are always legal: by LDS-DMA, 4 bytes
    // per lane, into this thread's own words of xgl -- no register is held across the MFMA slot in between, and the compiler,
    // which does not see the requests, puts no wait for them in front of anything
    const unsigned lds_xg = (unsigned)(size_t)xgl + (unsigned)((hx * NG * 256 + vh * 64) * 4);
    const float* xgr = xgl + hx * NG * 256 + tidh;
    auto load_xg = [&](int s, int j) {
        const int eb = min((tile0 + min(j, nt - 1)) * RB + cj, p.B - 1);
        const int t = d == 0 ? s : p.T - 1 - s;
        const float* xr = p.xp + ((size_t)t * p.B + eb) * p.Np + (half_ok ? xcol : 0) + cu;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(lds_xg + g * 1024), "v"(xr + g * RU) : "memory");
    };
    load_xg(0, 0);
    unsigned pollv = 0;             // B's wave 0: the counter shard this lane read at the start of its MFMA slot
    // through the builtin, so that the compiler knows the prologue's loads (W_hh, biases, lengths) have arrived: told by inline
    // assembly it would wait for them at their first use INSIDE the loop, where vmcnt(0) also waits for that slot's requests
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring_barrier();

    const int nslots = 2 * NQ + 1;
    for (int gs = 0; gs < nslots; ++gs) {
        const bool mrole = (gs & 1) == hx;
        const int q = mrole ? (gs - hx) >> 1 : (gs - 1 - hx) >> 1;
        const bool q_ok = q >= 0 && q < NQ;
        const int s = q_ok ? q / nte : 0, j = q_ok ? q - s * nte : 0;
        const bool tile_ok = q_ok && j < nt;
        unsigned long long t0_ = 0, c0_ = 0, t1_ = 0, t2_ = 0, clast = 0;
        if (STAMP) { __builtin_amdgcn_sched_barrier(0); t0_ = __builtin_amdgcn_s_memrealtime(); c0_ = __builtin_amdgcn_s_memtime(); clast = c0_; __builtin_amdgcn_sched_barrier(0); }
        if (mrole) {
            // ---- B's wave 0 asks whether item q + 2's chain has finished its previous step; the answer is read behind the MFMAs
            const int qp = q + 2;
            const int sp = qp / nte, jp = qp - sp * nte;
            const bool poll = hx == 1 && vh == 0 && qp < NQ && sp >= 1 && jp < nt;
            const unsigned* cp = p.cnt + ((size_t)(d * p.ntiles + tile0 + (poll ? jp : 0)) * p.T + (poll ? sp - 1 : 0)) * kPersist16CntWords +
                                 (lane & (kPersist16Shards - 1)) * 64;
            const unsigned need = (unsigned)((nwg32 + kPersist16Shards - 1 - (lane & (kPersist16Shards - 1))) / kPersist16Shards);
            if (poll) {
                pollv = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tile_ok) {
                f32x4 acc[NG], acl[NG];        // hi.hi ; (hi.lo + lo.hi) * 2^11
#pragma unroll
                for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (s > 0) {
                    const unsigned char* sb = sbuf + (q & 1) * sbytes + lane * 16;
                    f16x8 bc[2], bn[2];
                    bc[0] = *reinterpret_cast<const f16x8*>(sb + (kb0 * 2) * 1024);
                    bc[1] = *reinterpret_cast<const f16x8*>(sb + (kb0 * 2 + 1) * 1024);
                    if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RSTAMP(8); }
#pragma unroll
                    for (int i = 0; i < NKW; ++i) {
                        if (i + 1 < NKW) {
                            const int kbn = min(kb0 + i + 1, max(kb1 - 1, kb0));
                            bn[0] = *reinterpret_cast<const f16x8*>(sb + (kbn * 2) * 1024);
                            bn[1] = *reinterpret_cast<const f16x8*>(sb + (kbn * 2 + 1) * 1024);
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
            }
            if (STAMP) { __builtin_amdgcn_sched_barrier(0); t1_ = __builtin_amdgcn_s_memrealtime(); tacc[6] += __builtin_amdgcn_s_memtime() - c0_; __builtin_amdgcn_sched_barrier(0); }
            // ---- everything this wave requested in its last cell slot: publish stores (drained), B's DMA (landed), x-projection
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);      // the same wait where the compiler sees it: nothing of this wave's is in flight
            if (hx == 1) {
                // the wave that drains last signals item q - 1 for the whole workgroup (A's stores of it drained a slot ago)
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(&sync[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                const int qs = q - 1;
                if ((old & 3) == 3 && qs >= 0 && qs < NQ) {
                    const int ss = qs / nte, js = qs - ss * nte;
                    const int chain = d * p.ntiles + tile0 + js;
                    const bool drop = chain == 0 && w32 == p.drop_wg && ss == p.drop_step;
                    if (js < nt && lane == 0 && !drop)
                        __hip_atomic_fetch_add(&p.cnt[((size_t)chain * p.T + ss) * kPersist16CntWords + shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (poll && !sync[0]) {
                    unsigned long long tp_ = 0;
                    if (STAMP) tp_ = __builtin_amdgcn_s_memrealtime();
                    unsigned spins = 0;
                    unsigned got = pollv;
                    while (__builtin_amdgcn_ballot_w64(got < need) != 0) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 1023u) == 0 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { sync[0] = 1; break; }
                        if (spins > p.spin_limit) { atomicExch(p.err, 1u); sync[0] = 1; break; }
                        got = lane < kPersist16Shards ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                    }
                    if (STAMP) tacc[5] += __builtin_amdgcn_s_memrealtime() - tp_;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // (free here: every poll has been consumed) no load is pending at the loop's back edge
            }
        } else {
            // ---- B: DMA requests for item q + 2 into the ring slot its last readers left two barriers ago (its chain's
            // previous step was found complete in the slot before this one); each wave brings the k-blocks it will multiply
            if (hx == 1) {
                const int qd = q + 2;
                const int sd = qd / nte, jd = qd - sd * nte;
                if (qd < NQ && sd >= 1 && jd < nt) {
                    const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(p.hpack) + (size_t)((sd - 1) & 1) * hp_par +
                                                (size_t)(d * p.ntiles + tile0 + jd) * p.nkb * 2048;
                    const unsigned ldst = lds_sbuf + (unsigned)((qd & 1) * sbytes);
#pragma unroll
                    for (int i = 0; i < NKW; ++i)
                        if (i + 1 < NKW || kb0 + i < kb1) {
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl) {
                                const unsigned po = (unsigned)((kb0 + i) * 2 + pl) * 1024u;
                                ring_dma(gsrc + po, (unsigned)lane * 16u, ldst + po);      // piece offset on the scalar side: one address register
                            }
                        }
                }
            }
            RSTAMP(11);
            if (tile_ok) {
                // ---- K-split reduction (fixed order) + cell + publish, one (unit, clip) pair per thread
                const int t = d == 0 ? s : p.T - 1 - s;
                const int tile = tile0 + j;
                const int eb = tile * RB + cj;
                const bool eact = cunit_ok && eb < p.B;
                const int mylen = st_len[j * 16 + cj];
                float hn = 0.f;
                if (eact) {
                    float hg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) sum += red[((k * NG + g) * 16 + cu) * RRP + cj];
                        hg[g] = sum + bh[g];
                    }
                    float cprev = KIND == DSMI_RNN_LSTM ? st_c[j * RNT + tid] : 0.f;
                    float xg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) xg[g] = xgr[g * 256];
                    hn = rnn_cell<KIND>(xg, hg, st_h[j * RNT + tid], cprev, t < mylen);
                    st_h[j * RNT + tid] = hn;
                    if (KIND == DSMI_RNN_LSTM) st_c[j * RNT + tid] = cprev;
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = hn;
                } else if (half_ok && eb < p.B && cunit < p.Hs) {
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit] = 0.f;     // padding units of the last workgroup
                }
                RSTAMP(12);
                // publish: the wave's 8 clips x 8 units x 2 planes -> 256 bytes of LDS -> one 16-byte sc1 store for each of 16 lanes
                const _Float16 h1 = (_Float16)hn;
                const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
                unsigned short* sw = stg + v * 128;
                sw[(lane >> 3) * 8 + ce] = __builtin_bit_cast(unsigned short, h1);
                sw[64 + (lane >> 3) * 8 + ce] = __builtin_bit_cast(unsigned short, h2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave, LDS in order: the 16-byte rows are complete
                if (lane < 16 && half_ok) {
                    const u32x4 row = *reinterpret_cast<const u32x4*>(sw + lane * 8);
                    const unsigned off = (unsigned)((s & 1) * hp_par) + (unsigned)((size_t)(d * p.ntiles + tile) * p.nkb * 2048) +
                                         (unsigned)(w16 >> 1) * 2048u + (unsigned)(2 * (w16 & 1) + cuh) * 256u +
                                         (unsigned)((vh & 1) * 8 + (lane & 7)) * 16u + (unsigned)(lane >> 3) * 1024u;
                    __builtin_amdgcn_raw_buffer_store_b128(row, hrs, off, 0, 16);
                }
            }
            // ---- x-projection of this half's next cell item, consumed two slots from now
            if (q + 1 >= 0 && q + 1 < NQ) {
                const int sn = (q + 1) / nte, jn = (q + 1) - sn * nte;
                load_xg(sn, jn);
            }
            RSTAMP(13);
        }
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            t2_ = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
        ring_barrier();
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t3_ = __builtin_amdgcn_s_memrealtime();
            if (mrole) { tacc[0] += t1_ - t0_; tacc[1] += t2_ - t1_; tacc[3] += t3_ - t2_; }
            else { tacc[2] += t2_ - t0_; tacc[4] += t3_ - t2_; }
            tacc[7] += 1;
        }
    }
    if (STAMP && lane == 0) {
        unsigned long long* o = p.dbg + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + v) * 16;
        for (int k = 0; k < 16; ++k) o[k] = tacc[k];
    }
}

size_t ring_lds_bytes(int kind, int nkb) {
    const int NG = kind == DSMI_RNN_GRU ? 3 : (kind == DSMI_RNN_LSTM ? 4 : 1);
    return (size_t)2 * nkb * 2048 + (size_t)2 * 4 * NG * 16 * RRP * 4 + (size_t)RMAXT * RNT * 4 * (kind == DSMI_RNN_LSTM ? 2 : 1) +
           8 * 128 * 2 + (size_t)2 * NG * 256 * 4 + RMAXT * 16 * 4 + 32 * 4;
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
    return std::min(ceil_div(B, RB), RMAXT);
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
    if (a.ntw < 1 || a.ntw > RMAXT || a.tile_end <= a.tile0) return false;
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step; a.dbg = p.dbg;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch_ring<DSMI_RNN_GRU>(a, s, p.ev);
        case DSMI_RNN_LSTM: return launch_ring<DSMI_RNN_LSTM>(a, s, p.ev);
        default: return launch_ring<DSMI_RNN_TANH>(a, s, p.ev);
    }
}

}  // namespace dsmi
