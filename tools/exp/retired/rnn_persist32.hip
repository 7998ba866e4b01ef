// Persistent recurrent layer, throughput variant: 32 hidden units per workgroup, one workgroup per CU, 16-clip batch tiles.
//
// Same contract, arithmetic, state layout and hand-off protocol as rnn_persist16.hip (all T steps of a BatchRNN in one
// launch, W_hh resident on chip, split-fp16 products on v_mfma_f32_16x16x32_f16, counter form of the agent-scope
// hand-off with sc1 stores / sc1 loads, bounded spins).  What changes is how much of a chain one CU carries:
//
//   * a workgroup owns 32 hidden units = one whole 32-wide k-block of the state = G x 2 MFMA row tiles, so a chain is
//     H / 32 workgroups wide (cfgA: 25) and the four chains of a 32-clip batch (2 directions x 2 tiles) need 100 CUs;
//   * four waves, one per SIMD, up to 512 registers each: a wave's K-slice of W_hh (up to 7 k-blocks x G x 2 tiles x 2
//     planes) lives in registers, the seventh k-block in LDS when there is one;
//   * every thread computes two cells.
//
// Why: with two batches in flight (two handles, two streams, one gate lane each: api.hip) the 16-unit kernel puts 400
// half-CU workgroups on 256 CUs and the two workgroups of a CU hide each other only partly -- the busy part of a step
// (state ingest from L2, MFMAs, K-split reduction, cell) is bound by pipes the two share: a co-resident layer takes 2.3 ms
// against 1.57 ms alone.  Two kernels on DISJOINT CUs barely disturb each other (+7 %: DESIGN.md 4), and a CU that
// carries 32 units ingests the chain's state once for twice the output.  This kernel gives each batch in flight its own
// 100 CUs.  Alone it is slower per step than the 16-unit kernel (twice the MFMAs behind one state load), so it is used
// when the handle has been told that two batches are in flight (dsmi_model_set_inflight).
#include "common.h"
#include "rnn_cell.h"
#include <cstring>

namespace dsmi {

namespace {

constexpr int RNW = 4;                 // waves per workgroup (K-split)
constexpr int RNT = RNW * 64;
constexpr int RU = 32;                 // hidden units per workgroup
constexpr int RB = 16;                 // clips per batch tile
constexpr int RRP = 20;                // row pitch (words) of the reduce buffer
constexpr int RMAXZ = 4;               // batch tiles one workgroup can walk
constexpr int RNKR = 6;                // k-blocks of W_hh a wave keeps in registers; further ones sit in LDS
constexpr size_t R_LDS = 144 * 1024;   // > half of the CU's LDS: one workgroup per CU (the register budget says the same)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

struct P32Args {
    const uint16_t* whh[2];    // pack_whh32 per direction
    const float* bhh[2]; const float* xp; float* out[2];
    const int32_t* lens; uint16_t* hpack; unsigned* cnt; unsigned* err;
    int B, T, H, Hs, Np, nwg, nkb;
    int ntiles, pgroups, D;
    unsigned spin_limit;
    int drop_wg, drop_step;
};

template <int KIND, int NKW>
__global__ __launch_bounds__(RNT, 1) void rnn_persist32_kernel(P32Args p) {
    constexpr int NG = KIND == DSMI_RNN_GRU ? 3 : (KIND == DSMI_RNN_LSTM ? 4 : 1);
    constexpr int NKR = NKW < RNKR ? NKW : RNKR;
    constexpr int NS = NG * 2;                                        // accumulator tiles per wave: gate x row tile
    extern __shared__ __attribute__((aligned(16))) float rlds[];
    float* red = rlds;                                               // [RNW][8 slots][16 units][RRP]
    int& s_dead = *reinterpret_cast<int*>(red + RNW * 8 * 16 * RRP);
    float* st_h = red + RNW * 8 * 16 * RRP + 32;                    // [RMAXZ][2][256] carried state when a workgroup walks several tiles
    float* st_c = st_h + RMAXZ * 512;
    int* st_len = reinterpret_cast<int*>(st_c + RMAXZ * 512);       // [RMAXZ][256]
    u32x4* wlds = reinterpret_cast<u32x4*>(st_len + RMAXZ * 256);   // [RNW][NKW - NKR][NG][2 tiles][2 planes][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lg = lane >> 4;
    const int w = blockIdx.x;
    const int d = blockIdx.y / p.pgroups, pg = blockIdx.y - d * p.pgroups;
    const int nz = (p.ntiles - pg + p.pgroups - 1) / p.pgroups;
    const bool multi = nz > 1;
    const int GU = NG * RU;
    const size_t xcol = (size_t)d * p.nwg * GU + (size_t)w * GU;
    if (tid == 0) s_dead = 0;

    // ---- resident operand: this wave's k-blocks of the split W_hh, all gates, both row tiles
    const int kb0 = (v * p.nkb) / RNW, kb1 = ((v + 1) * p.nkb) / RNW;
    f16x8 wv[NKR][NG][2][2];
    u32x4* wl = wlds + (size_t)v * (NKW - NKR) * NG * 4 * 64 + lane;
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(p.whh[d]) + ((size_t)w * p.nkb) * (NG * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < NKW; ++i) {
            const int kb = min(kb0 + i, max(kb1 - 1, kb0));
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const u32x4 frag = wp[((((size_t)kb * NG + g) * 2 + rt) * 2 + pl) * 64];
                        if (i < NKR) wv[i < NKR ? i : 0][g][rt][pl] = __builtin_bit_cast(f16x8, frag);
                        else wl[((((i - NKR) * NG + g) * 2 + rt) * 2 + pl) * 64] = frag;
                    }
        }
    }
    const size_t hp_par = (size_t)p.D * p.ntiles * p.nkb * 2048;     // bytes per parity
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * hp_par), 0x00020000);

    // cell role: thread -> clip cj, units cu and cu + 16 (row tiles 0 and 1); a wave's 2-byte stores of the new state are
    // 128 contiguous bytes of one k-group block
    const int cuh = tid >> 7, ce = tid & 7, cj = (tid >> 3) & 15;
    const int cu = 8 * cuh + ce;
    int cunit[2];
    bool cok[2];
    float bh[2][NG];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        cunit[r] = w * RU + 16 * r + cu;
        cok[r] = cunit[r] < p.H;
#pragma unroll
        for (int g = 0; g < NG; ++g) bh[r][g] = cok[r] ? p.bhh[d][g * p.H + cunit[r]] : 0.f;
    }
    int mylen = 0;
    float hprev[2] = {0.f, 0.f}, cprev[2] = {0.f, 0.f};
    if (multi) {
        for (int z = 0; z < nz; ++z) {
            const int eb = (pg + z * p.pgroups) * RB + cj;
            st_h[z * 512 + tid] = 0.f; st_h[z * 512 + 256 + tid] = 0.f;
            st_c[z * 512 + tid] = 0.f; st_c[z * 512 + 256 + tid] = 0.f;
            st_len[z * 256 + tid] = eb < p.B ? p.lens[eb] : 0;
        }
    } else {
        const int eb = pg * RB + cj;
        mylen = eb < p.B ? p.lens[eb] : 0;
    }
    __syncthreads();

    unsigned* pend = nullptr;
    bool pend_drop = false;
    for (int s = 0; s < p.T; ++s) {
        const int t = d == 0 ? s : p.T - 1 - s;
        for (int z = 0; z < nz; ++z) {
            const int tile = pg + z * p.pgroups;
            const int b0 = tile * RB;
            const int nb = min(RB, p.B - b0);
            const int eb = b0 + cj;
            const bool clip_ok = cj < nb;
            const int chain = d * p.ntiles + tile;
            unsigned* cnt = p.cnt + (size_t)chain * p.T * kPersist16CntWords;
            const unsigned hchain = (unsigned)((size_t)chain * p.nkb * 2048);
            if (multi) {
                mylen = st_len[z * 256 + tid];
#pragma unroll
                for (int r = 0; r < 2; ++r) { hprev[r] = st_h[z * 512 + r * 256 + tid]; cprev[r] = st_c[z * 512 + r * 256 + tid]; }
            }
            // x-projection operands of this step do not depend on other workgroups: request them first
            float xg[2][NG];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < NG; ++g) xg[r][g] = 0.f;
            if (clip_ok) {
                const float* xr = p.xp + ((size_t)t * p.B + eb) * p.Np + xcol + cu;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (cok[r]) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) xg[r][g] = xr[g * RU + 16 * r];
                    }
            }
            f32x4 acc[NS], acl[NS];        // hi.hi ; (hi.lo + lo.hi) * 2^11, slot = gate * 2 + row tile
#pragma unroll
            for (int q = 0; q < NS; ++q) { acc[q] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (s > 0) {
                // ---- wait until every workgroup of this chain has published h_{s-1} (bounded)
                if (v == 0 && !s_dead) {
                    unsigned spins = 0;
                    const unsigned* cp = &cnt[(size_t)(s - 1) * kPersist16CntWords + (lane & (kPersist16Shards - 1)) * 64];
                    const unsigned need = (unsigned)((p.nwg + kPersist16Shards - 1 - (lane & (kPersist16Shards - 1))) / kPersist16Shards);
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
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    if (kb0 + i < kb1) {
                        f16x8 wa[NS][2];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                                for (int pl = 0; pl < 2; ++pl)
                                    wa[g * 2 + rt][pl] = i < NKR ? wv[i < NKR ? i : 0][g][rt][pl]
                                                                 : __builtin_bit_cast(f16x8, wl[((((i - NKR) * NG + g) * 2 + rt) * 2 + pl) * 64]);
#pragma unroll
                        for (int q = 0; q < NS; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[q][1], hv[i][0], acl[q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < NS; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[q][0], hv[i][0], acc[q], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < NS; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[q][0], hv[i][1], acl[q], 0, 0, 0);
                    }
                }
            }
            // partial tiles -> LDS: D[row = unit 4 * lg + r][col = clip ln]
#pragma unroll
            for (int q = 0; q < NS; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((v * 8 + q) * 16 + 4 * lg + r) * RRP + ln] = acc[q][r] + acl[q][r] * kLoInv;
            if (multi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile's state stores are acknowledged
            __syncthreads();
            if (multi && tid == 0 && pend && !pend_drop) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ---- K-split reduction (fixed order) + cell + publish: two (unit, clip) pairs per thread
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float hn = 0.f;
                if (cok[r] && clip_ok) {
                    float hg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < RNW; ++k) sum += red[((k * 8 + g * 2 + r) * 16 + cu) * RRP + cj];
                        hg[g] = sum + bh[r][g];
                    }
                    hn = rnn_cell<KIND>(xg[r], hg, hprev[r], cprev[r], t < mylen);
                    hprev[r] = hn;
                    if (multi) { st_h[z * 512 + r * 256 + tid] = hn; if (KIND == DSMI_RNN_LSTM) st_c[z * 512 + r * 256 + tid] = cprev[r]; }
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit[r]] = hn;
                } else if (clip_ok && cunit[r] < p.Hs) {
                    p.out[d][((size_t)t * p.B + eb) * p.Hs + cunit[r]] = 0.f;     // padding units of the last workgroup
                }
                const _Float16 h1 = (_Float16)hn;
                const _Float16 h2 = (_Float16)((hn - (float)h1) * kLoScale);
                // k-block w of the chain: [plane][k-group = unit / 8][clip][unit % 8]
                const unsigned off = (unsigned)((s & 1) * hp_par) + hchain + (unsigned)w * 2048u +
                                     (unsigned)(2 * r + cuh) * 256u + (unsigned)cj * 16u + (unsigned)ce * 2u;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h1), hrs, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h2), hrs, off + 1024u, 0, 16);
            }
            const bool drop = chain == 0 && w == p.drop_wg && s == p.drop_step;
            if (multi) {
                pend = &cnt[(size_t)s * kPersist16CntWords + (w & (kPersist16Shards - 1)) * 64];
                pend_drop = drop;
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own stores
                __syncthreads();
                if (tid == 0 && !drop) __hip_atomic_fetch_add(&cnt[(size_t)s * kPersist16CntWords + (w & (kPersist16Shards - 1)) * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

inline uint16_t r_f16_bits(_Float16 h) {
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

template <int KIND>
bool launch32(const P32Args& a, hipStream_t s, const EvPair& ev) {
    const int nkw = ceil_div(a.nkb, RNW);
    const dim3 grid(a.nwg, a.D * a.pgroups, 1), block(RNT);
#define LAUNCH_R(N)                                                                                                  \
    do {                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_persist32_kernel<KIND, N>),                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)R_LDS);                            \
        DSMI_LAUNCH((rnn_persist32_kernel<KIND, N>), grid, block, R_LDS, s, ev, a);                                   \
    } while (0)
    if (KIND == DSMI_RNN_LSTM) {
        if (nkw <= 2) LAUNCH_R(2);
        else if (nkw <= 4) LAUNCH_R(4);
        else return false;
        return true;
    }
    if (nkw <= 2) LAUNCH_R(2);
    else if (nkw <= 4) LAUNCH_R(4);
    else if (nkw <= 6) LAUNCH_R(6);
    else if (nkw <= 7) LAUNCH_R(7);
    else return false;
#undef LAUNCH_R
    return true;
}

}  // namespace

// H a multiple of 32, at most 7 k-blocks per wave (H <= 896; 4 for an LSTM: H <= 512), both directions of a tile group
// on `n_cus` CUs (the caller passes the CUs of ONE gate lane), at most RMAXZ tiles per workgroup.
bool rnn_persist32_eligible(const RnnGeom& g32, int B, int n_cus, int* pgroups_out) {
    if (g32.U != RU || (g32.H % RU) != 0) return false;
    const int nkw = ceil_div(ceil_div(g32.H, 32), RNW);
    if (nkw > (g32.kind == DSMI_RNN_LSTM ? 4 : 7)) return false;
    if (g32.nwg * g32.D > n_cus) return false;
    const int ntiles = ceil_div(B, RB);
    const int pg = std::min(ntiles, n_cus / (g32.nwg * g32.D));
    if (ceil_div(ntiles, pg) > RMAXZ) return false;
    if (pgroups_out) *pgroups_out = pg;
    return true;
}

// w_hh [G*H][H] (torch layout) of one direction -> [workgroup][kb][gate][row tile][plane][lane][8] fp16 terms (hi, lo * 2^11);
// lane (u = lane & 15, kg = lane >> 4) element e holds W[gate * H + 32 * wg + 16 * tile + u][32 * kb + 8 * kg + e].
std::vector<uint16_t> pack_whh32(const RnnGeom& g32, const float* w_hh) {
    const int nkb = ceil_div(g32.H, 32), G = g32.G, H = g32.H;
    std::vector<uint16_t> out((size_t)g32.nwg * nkb * G * 2 * 2 * 64 * 8, 0);
    for (int w = 0; w < g32.nwg; ++w)
        for (int kb = 0; kb < nkb; ++kb)
            for (int gate = 0; gate < G; ++gate)
                for (int rt = 0; rt < 2; ++rt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int unit = w * RU + 16 * rt + (lane & 15);
                        if (unit >= H) continue;
                        for (int e = 0; e < 8; ++e) {
                            const int k = 32 * kb + 8 * (lane >> 4) + e;
                            if (k >= H) continue;
                            const float x = w_hh[(size_t)(gate * H + unit) * H + k];
                            const _Float16 h1 = (_Float16)x;
                            const _Float16 h2 = (_Float16)((x - (float)h1) * kLoScale);
                            const size_t base = ((((((size_t)w * nkb + kb) * G + gate) * 2 + rt) * 2) * 64 + lane) * 8 + e;
                            out[base] = r_f16_bits(h1); out[base + 512] = r_f16_bits(h2);
                        }
                    }
    return out;
}

bool launch_rnn_persist32(const RnnPersist16Launch& p, hipStream_t s) {
    P32Args a;
    for (int d = 0; d < 2; ++d) { a.whh[d] = p.whh16[d]; a.bhh[d] = p.bhh[d]; a.out[d] = p.out[d]; }
    a.xp = p.xp; a.lens = p.lens_dev; a.hpack = p.hpack16; a.cnt = p.counters; a.err = p.err;
    a.B = p.B; a.T = p.T; a.H = p.g.H; a.Hs = p.g.Kp; a.Np = p.g.Np; a.nwg = p.g.nwg; a.nkb = ceil_div(p.g.H, 32);
    a.ntiles = ceil_div(p.B, RB); a.pgroups = p.pgroups; a.D = p.g.D;
    a.spin_limit = p.spin_limit; a.drop_wg = p.drop_wg; a.drop_step = p.drop_step;
    switch (p.g.kind) {
        case DSMI_RNN_GRU: return launch32<DSMI_RNN_GRU>(a, s, p.ev);
        case DSMI_RNN_LSTM: return launch32<DSMI_RNN_LSTM>(a, s, p.ev);
        default: return launch32<DSMI_RNN_TANH>(a, s, p.ev);
    }
}

}  // namespace dsmi
