// Decoders on the GPU (gfx950): the handle behind dsmi_decoder*, greedy decode, and CTC prefix
// beam search with an optional word-level n-gram scorer.
//
// Replaces GreedyDecoder.decode and BeamCTCDecoder.decode of the reference
// (danspeech/deepspeech/decoder.py:183-198, 129-144).  The reference's beam search is the
// un-vendored third-party ctcdecode (C++, thread pool over the batch); here one workgroup
// decodes one utterance, the batch runs in parallel across CUs, and nothing leaves the GPU
// until the final beams.  Algorithm and the deliberate float64 carry: see oracle/beam.py,
// which restates the same published algorithm and is what the parity tests compare with.
//
// Per frame, for one utterance (256 threads):
//   1. log p(c) = log(p + FLT_MIN), vocabulary pruning (cutoff_prob / cutoff_top_n);
//   2. every (beam entry, character) pair in parallel: blank / repeat contributions go to the
//      entry, an extension looks its child up in the utterance's prefix-trie hash table,
//      checks the dictionary trie, adds alpha * ln P_lm + beta on the space character, and
//      becomes a candidate (or feeds an entry already in the beam);
//   3. exact top-`beam` of the candidates by (score desc, last character asc) with an 8-pass
//      radix select on the order-preserving bits of the float64 score;
//   4. survivors are committed (new trie nodes, hash insert), the rest are unlinked with the
//      reference's "remove" semantics so that emission timesteps behave the same.
#include "common.h"
#include "lm.h"
#include "lm.cpp.inc"
#include "lm_klm.cpp.inc"

#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace dsmi;

namespace {

constexpr int MAXBT = 1024;       // threads per utterance: a template parameter of the kernel (BT)
constexpr double BIN_SPAN = 64.0; // the selection histogram has one bin per thread, over the 64 nats below the frame's score bound
constexpr int LISTCAP = 256;      // members of the threshold bin that are ranked from a list (more: ranked against all keys)
constexpr int MAXCTX = kMaxOrder - 1;
constexpr int MAXC = 128;
constexpr int MEMO_UNSET = 0x7fc00001;   // a NaN pattern no float computation produces

// One record of the node pool (HBM).  Written at creation, tstep / lpc updated in place by get_path_trie's "better emission
// frame" rule; read only when a dormant prefix re-enters the beam (the walk in resolve()) and by the final path output.
struct __attribute__((aligned(32))) NodeRec { int parent, ch, tstep, depth; double lpc; double pad; };
static_assert(sizeof(NodeRec) == 32, "NodeRec is two 16-byte stores");

struct BeamArgs {
    const float* probs; const int32_t* sizes; int T, C, blank, space, beam, cutoff_top_n; float cutoff_prob;
    int has_lm, order; double alpha, beta;
    LmView lm; const int32_t* trie_next; const int32_t* trie_word; int unk, bos;
    int ncap;                    // per-utterance node pool capacity
    NodeRec* nodes;
    int32_t* dbg;                // optional [B][4]: dormant prefixes revived, walk hops, list rankings, full rankings
    // outputs
    int32_t *out_tok, *out_step, *out_len, *out_n; double* out_score;
};

// On-chip layout of one utterance's search, shared by the kernel and the launcher's size check.
struct Carve {
    size_t lp, sd, bprev, nbprev, score, uplpc, ownlpc, bcur, nbcur, ckey, lkey, su;          // 8-byte items
    size_t node, ch, ds, depth, up, upch, upnode, memo, ctx, newslot, use, lidx, hist, wtot, wsurv, wfresh, si, pfp;   // 4-byte
    size_t cell;                                                                                // 2-byte
    size_t lmv;
    size_t bytes;
};
__host__ __device__ inline Carve carve(int BW, int C, int BT) {
    const int NBINS = BT, NWAVE = BT / 64;
    Carve k; size_t o = 0;
    const size_t NMAX = (size_t)BW * (C + 1);
    auto take = [&](size_t n, size_t sz) { const size_t at = o; o += ((n * sz + 15) & ~(size_t)15); return at; };
    k.lp = take(2 * MAXC, 8); k.sd = take(8, 8);
    k.bprev = take(2 * BW, 8); k.nbprev = take(2 * BW, 8); k.score = take(2 * BW, 8); k.uplpc = take(2 * BW, 8); k.ownlpc = take(2 * BW, 8);
    k.bcur = take(BW, 8); k.nbcur = take(BW, 8); k.ckey = take(NMAX, 8); k.lkey = take(LISTCAP, 8); k.su = take(4, 8);
    k.node = take(2 * BW, 4); k.ch = take(2 * BW, 4); k.ds = take(2 * BW, 4); k.depth = take(2 * BW, 4); k.up = take(2 * BW, 4);
    k.upch = take(2 * BW, 4); k.upnode = take(2 * BW, 4); k.memo = take(2 * BW, 4); k.ctx = take((size_t)2 * BW * MAXCTX, 4);
    k.newslot = take(BW, 4); k.use = take(2 * MAXC, 4); k.lidx = take(LISTCAP, 4); k.hist = take(NBINS, 4);
    k.wtot = take(NWAVE, 4); k.wsurv = take(NWAVE, 4); k.wfresh = take(NWAVE, 4); k.si = take(16, 4); k.pfp = take(2 * MAXC, 4);
    k.cell = take((size_t)2 * (((size_t)BW * C + 1) & ~(size_t)1), 2);
    k.lmv = take(1, sizeof(LmView));
    k.bytes = o;
    return k;
}

__device__ __forceinline__ double lse2(double x, double y) {
    // log(exp(x - m) + exp(y - m)) + m with m = max: the larger term's exp is exactly 1
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    const double m = fmax(x, y), d = fmin(x, y) - m;
    return log(1.0 + exp(d)) + m;
}

__device__ __forceinline__ uint64_t okey(double v) {   // order-preserving bits, > 0 for every double
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unokey(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
template <int NBINS>
__device__ __forceinline__ int score_bin(double U, double s) {     // monotone: a better score never lands in a later bin
    const double q = (U - s) * (NBINS / BIN_SPAN);
    if (!(q < (double)(NBINS - 1))) return NBINS - 1;              // also NaN (U - s with both infinite)
    return q > 0.0 ? (int)q : 0;
}
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Tup { int up, upch, upnode; double uplpc; };

// The scorer's n-gram look-ups stay out of line: three call sites, rarely taken, a long chain of dependent loads each.
__device__ __noinline__ float lm_word_log10(const LmView* v, const int32_t* ctx, int n, int32_t w, int32_t unk) {
    return lm_cond_log10(*v, ctx, n, w, unk);
}

// CTC prefix beam search of one utterance by one workgroup; the algorithm, array for array, is oracle/beam_flat.py
// (tests/test_oracle_beam_flat.py holds that formulation to ctcdecode's pointer trie as restated in oracle/beam.py).
// A frame is six barrier-separated phases, all on LDS; HBM sees fire-and-forget stores (new node records, timestep updates)
// and, one frame ahead, the load of the next probability row:
//   F1  every candidate in registers: thread idx < nb = beam entry idx (its blank / repeat terms, and -- pull form -- the
//       extension it receives from its parent when that is a beam entry), the others = (entry, character) pairs; keys go
//       into a 1024-bin histogram of (bound - score) at 1/16 nat;
//   F2  prefix sums of the histogram, wave level; the last wave turns the prefetched row into log-probabilities;
//   F3  the bin in which the beam_width-th best score lies, and how many of its members survive;
//   F4  members of that bin ranked exactly (score desc, character asc, index asc) when not all of them survive; survivors
//       numbered inside each wave;
//   F5  new slot numbers; what left the beam, which dormant prefixes came back;
//   F6  commit: the new beam entry by entry into the other buffer (edge tuples resolved through whatever left), node
//       records to HBM, the child table of the new beam.
template <int BT, int KMAX>
__global__ __launch_bounds__(BT) void beam_kernel(BeamArgs a) {
    constexpr int NWAVE = BT / 64, NBINS = BT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int C = a.C, BW = a.beam;
    const Carve kv = carve(BW, C, BT);
    LmView* s_lm = reinterpret_cast<LmView*>(smem_raw + kv.lmv);
    double* lp = reinterpret_cast<double*>(smem_raw + kv.lp);            // [2][MAXC] by frame parity
    double* sd = reinterpret_cast<double*>(smem_raw + kv.sd);            // [0..1] ln p(blank), [2..3] max_c lp
    double* e_bprev = reinterpret_cast<double*>(smem_raw + kv.bprev);    // entry arrays [2][BW]
    double* e_nbprev = reinterpret_cast<double*>(smem_raw + kv.nbprev);
    double* e_score = reinterpret_cast<double*>(smem_raw + kv.score);
    double* e_uplpc = reinterpret_cast<double*>(smem_raw + kv.uplpc);
    double* e_ownlpc = reinterpret_cast<double*>(smem_raw + kv.ownlpc);
    double* e_bcur = reinterpret_cast<double*>(smem_raw + kv.bcur);      // [BW]
    double* e_nbcur = reinterpret_cast<double*>(smem_raw + kv.nbcur);
    uint64_t* c_key = reinterpret_cast<uint64_t*>(smem_raw + kv.ckey);   // [NMAX] 0 = no candidate, else okey(score)
    uint64_t* l_key = reinterpret_cast<uint64_t*>(smem_raw + kv.lkey);   // [LISTCAP]
    uint64_t* su = reinterpret_cast<uint64_t*>(smem_raw + kv.su);        // [0] smallest, [1] largest key of the beam
    int* e_node = reinterpret_cast<int*>(smem_raw + kv.node);
    int* e_ch = reinterpret_cast<int*>(smem_raw + kv.ch);
    int* e_ds = reinterpret_cast<int*>(smem_raw + kv.ds);
    int* e_depth = reinterpret_cast<int*>(smem_raw + kv.depth);
    int* e_up = reinterpret_cast<int*>(smem_raw + kv.up);
    int* e_upch = reinterpret_cast<int*>(smem_raw + kv.upch);
    int* e_upnode = reinterpret_cast<int*>(smem_raw + kv.upnode);
    int* e_memo = reinterpret_cast<int*>(smem_raw + kv.memo);
    int* e_ctx = reinterpret_cast<int*>(smem_raw + kv.ctx);              // [2][BW][MAXCTX]
    int* newslot = reinterpret_cast<int*>(smem_raw + kv.newslot);
    int* usev = reinterpret_cast<int*>(smem_raw + kv.use);               // [2][MAXC]
    int* l_idx = reinterpret_cast<int*>(smem_raw + kv.lidx);
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw + kv.hist);
    int* wtot = reinterpret_cast<int*>(smem_raw + kv.wtot);
    int* wsurv = reinterpret_cast<int*>(smem_raw + kv.wsurv);
    int* wfresh = reinterpret_cast<int*>(smem_raw + kv.wfresh);
    int* si = reinterpret_cast<int*>(smem_raw + kv.si);
    float* pfp = reinterpret_cast<float*>(smem_raw + kv.pfp);            // [MAXC] row being pruned, [MAXC] the same sorted
    short* cell = reinterpret_cast<short*>(smem_raw + kv.cell);          // [2][CELLS]: (entry, label) -> an entry of that edge
    const int CELLS = (BW * C + 1) & ~1;
#define s_nb si[0]
#define s_bstar si[1]
#define s_r si[2]
#define s_mb si[3]
#define s_nlist si[4]
#define s_nrev si[5]
#define s_kmin su[0]
#define s_kmax su[1]

    NodeRec* nodes = a.nodes + (size_t)b * a.ncap;
    const int T = a.sizes ? min(a.sizes[b], a.T) : a.T;
    const float* pb = a.probs + (size_t)b * a.T * C;
    const int NCTX = a.order - 1;
    const double betap = fmax(0.0, a.beta);
    const bool prune = a.cutoff_prob < 1.0f || a.cutoff_top_n < C;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int dbg_rev = 0, dbg_hops = 0, dbg_list = 0, dbg_full = 0;

    // The last wave turns row tt of the probabilities (held in p0, p1: labels lane and lane + 64; pbl = p(blank)) into the
    // log-probabilities, the vocabulary mask and the two scalars of frame tt (decoder_utils get_pruned_log_probs).
    auto make_row = [&](int tt, float p0, float p1, float pbl) {
        const int par = tt & 1;
        double mx = -INFINITY;
        if (lane < C) { const double l = log((double)p0 + 1.17549435e-38); lp[par * MAXC + lane] = l; usev[par * MAXC + lane] = 1; mx = l; }
        if (lane + 64 < C) { const double l = log((double)p1 + 1.17549435e-38); lp[par * MAXC + lane + 64] = l; usev[par * MAXC + lane + 64] = 1; mx = fmax(mx, l); }
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) { sd[par] = pbl > 0.f ? log((double)pbl) : -INFINITY; sd[2 + par] = mx; }
        if (prune) {
            // rank of every label by (probability desc, index asc); the cumulative sum runs in that order, in double, like the
            // reference's loop, so that `cum >= cutoff_prob` falls on the same label
            if (lane < C) pfp[lane] = p0;
            if (lane + 64 < C) pfp[lane + 64] = p1;
            __builtin_amdgcn_wave_barrier();
            int rk[2] = {0, 0};
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c >= C) continue;
                const float pc = pfp[c];
                int r = 0;
                for (int o = 0; o < C; ++o) { const float po = pfp[o]; r += (po > pc) || (po == pc && o < c); }
                rk[h] = r;
                pfp[MAXC + r] = pc;
            }
            __builtin_amdgcn_wave_barrier();
            int len = C;
            if (a.cutoff_prob < 1.0f) {
                int mine = C;
                for (int h = 0; h < 2; ++h) {
                    const int c = lane + 64 * h;
                    if (c >= C) continue;
                    double cum = 0.0;
                    for (int k = 0; k <= rk[h]; ++k) cum += (double)pfp[MAXC + k];
                    if (cum >= (double)a.cutoff_prob || rk[h] + 1 >= a.cutoff_top_n) mine = min(mine, rk[h] + 1);
                }
                for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
                len = mine;
            } else len = a.cutoff_top_n;
            for (int h = 0; h < 2; ++h) { const int c = lane + 64 * h; if (c < C) usev[par * MAXC + c] = rk[h] < len; }
        }
    };

    // ---- start: the root is the whole beam
    for (int q = tid; q < NBINS; q += BT) hist[q] = 0;
    for (int q = tid; q < CELLS; q += BT) reinterpret_cast<int*>(cell)[q] = -1;       // both buffers (2 * CELLS shorts)
    if (tid == 0) {
        NodeRec r; r.parent = -1; r.ch = -1; r.tstep = 0; r.depth = 0; r.lpc = -INFINITY; r.pad = 0.0;
        nodes[0] = r;
        e_node[0] = 0; e_ch[0] = -1; e_ds[0] = 0; e_depth[0] = 0; e_up[0] = -1; e_upch[0] = -1; e_upnode[0] = -1; e_memo[0] = MEMO_UNSET;
        e_bprev[0] = 0.0; e_nbprev[0] = -INFINITY; e_score[0] = 0.0; e_uplpc[0] = -INFINITY; e_ownlpc[0] = -INFINITY;
        for (int k = 0; k < MAXCTX; ++k) e_ctx[k] = a.bos;
        s_nb = 1; s_kmin = okey(0.0); s_kmax = okey(0.0); s_nlist = 0; s_nrev = 0;
        *s_lm = a.lm;
    }
    if (wid == NWAVE - 1 && T > 0)
        make_row(0, lane < C ? pb[lane] : 0.f, lane + 64 < C ? pb[lane + 64] : 0.f, pb[a.blank]);
    __syncthreads();
    int cur = 0, nn = 1;

    for (int t = 0; t < T; ++t) {
        const int nxt = cur ^ 1, par = t & 1;
        const int nb = s_nb;
        const double* lpv = lp + par * MAXC; const int* use = usev + par * MAXC;
        const double* bprev = e_bprev + cur * BW; const double* nbprev = e_nbprev + cur * BW; const double* score = e_score + cur * BW;
        double* uplpc = e_uplpc + cur * BW; const double* ownlpc = e_ownlpc + cur * BW;
        const int* node = e_node + cur * BW; const int* ech = e_ch + cur * BW; const int* eds = e_ds + cur * BW;
        const int* depth = e_depth + cur * BW; const int* up = e_up + cur * BW; const int* upch = e_upch + cur * BW;
        const int* upnode = e_upnode + cur * BW; int* memo = e_memo + cur * BW; const int* ctx = e_ctx + (size_t)cur * BW * MAXCTX;
        short* cellc = cell + (size_t)cur * CELLS; short* celln = cell + (size_t)nxt * CELLS;
        // next frame's row, on its way while this frame runs
        float pn0 = 0.f, pn1 = 0.f, pnb = 0.f;
        if (wid == NWAVE - 1 && t + 1 < T) {
            const float* pr = pb + (size_t)(t + 1) * C;
            if (lane < C) pn0 = pr[lane];
            if (lane + 64 < C) pn1 = pr[lane + 64];
            pnb = pr[a.blank];
        }
        const bool full = a.has_lm && nb == BW;
        const double mincut = a.has_lm ? unokey(s_kmin) + sd[par] - betap : -INFINITY;
        const double U = unokey(s_kmax) + sd[2 + par] + betap + 1.2;      // no score of this frame exceeds it (alpha >= 0)
        const int NP = nb * C, N = nb + NP;
        const float invC = 1.0f / (float)C;

        // the extension of prefix P by character c (not a repeat of its blank-terminated self): ctc_beam_search_decoder.cpp's
        // log_p, with the scorer's word score on the space character
        auto pair_logp = [&](int P, int c, int arcw) -> double {
            const double l = lpv[c];
            double logp = -INFINITY;
            if (c == ech[P]) { if (bprev[P] > -INFINITY) logp = l + bprev[P]; }
            else logp = l + score[P];
            if (a.has_lm && c == a.space) {
                double lm = kOovScore;
                if (arcw >= 0) {
                    float l10;
                    const int mm = memo[P];
                    if (mm != MEMO_UNSET) l10 = __int_as_float(mm);
                    else {
                        l10 = lm_word_log10(s_lm, ctx + (size_t)P * MAXCTX, NCTX, arcw, a.unk);
                        memo[P] = __float_as_int(l10);
                    }
                    lm = (double)l10 / (double)kLog10E;
                }
                logp += lm * a.alpha;
                logp += a.beta;
            }
            return logp;
        };

        // ---- F1: candidates
        uint64_t key[KMAX]; int mark[KMAX], arcv[KMAX], binv[KMAX];     // mark: -1 none, 0 entry, 1 fresh child, 2 + rep: dormant top
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * BT;
            key[k] = 0; mark[k] = -1; arcv[k] = -1; binv[k] = NBINS;
            if (idx < nb) {
                const int j = idx;
                const double sc = score[j];
                double bl = -INFINITY, rp = -INFINITY, ex = -INFINITY;
                if (use[a.blank] && !(full && lpv[a.blank] + sc < mincut)) bl = lpv[a.blank] + sc;
                const int cj = ech[j];
                if (cj >= 0 && use[cj] && !(full && lpv[cj] + sc < mincut)) rp = lpv[cj] + nbprev[j];
                const int P = up[j];
                if (P >= 0) {
                    const int c = upch[j];
                    if (use[c] && !(full && lpv[c] + score[P] < mincut)) {
                        if (uplpc[j] < lpv[c]) {            // get_path_trie: a better emission frame for the edge's top node
                            uplpc[j] = lpv[c];
                            NodeRec* tn = nodes + upnode[j];
                            tn->lpc = lpv[c]; tn->tstep = t;
                        }
                        if (upnode[j] == node[j]) {
                            int arcw = -1;
                            if (a.has_lm && c == a.space) arcw = a.trie_word[eds[P]];
                            ex = pair_logp(P, c, arcw);
                        }
                    }
                }
                const double nbc = lse2(rp, ex), s = lse2(bl, nbc);
                e_bcur[j] = bl; e_nbcur[j] = nbc;
                key[k] = okey(s); mark[k] = 0;
            } else if (idx < N) {
                const int p = idx - nb;
                const int i = min((int)(((float)p + 0.5f) * invC), nb - 1), c = p - i * C;
                if (c != a.blank && use[c] && !(full && lpv[c] + score[i] < mincut)) {
                    const int r = cellc[i * C + c];
                    bool ok = true; int mk = 1, arc = -1;
                    if (r >= 0) { if (upnode[r] == node[r]) ok = false; else mk = 2 + r; }
                    if (ok && a.has_lm && (r < 0 || c == a.space)) {
                        arc = c == a.space ? a.trie_word[eds[i]] : a.trie_next[(size_t)eds[i] * C + c];
                        if (r < 0 && arc < 0) ok = false;           // the dictionary has no such arc (space: no word ends here)
                    }
                    if (ok) { key[k] = okey(pair_logp(i, c, arc)); mark[k] = mk; arcv[k] = arc; }
                }
            }
            const bool valid = key[k] != 0;
            if (idx < N) c_key[idx] = key[k];
            int bin = NBINS;
            if (valid) bin = score_bin<NBINS>(U, unokey(key[k]));
            binv[k] = bin;
            const bool lastb = valid && bin == NBINS - 1;          // the hopeless ones pile up in the last bin: one add per wave
            const unsigned long long bl = __ballot(lastb);
            if (valid && !lastb) atomicAdd(&hist[bin], 1u);
            if (bl && lane == __builtin_ctzll(bl)) atomicAdd(&hist[NBINS - 1], (unsigned)__builtin_popcountll(bl));
        }
        __syncthreads();

        // ---- F2: prefix sums inside each wave; the next row; housekeeping
        const int hv = (int)hist[tid];
        hist[tid] = 0;
        int incl = hv;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wtot[wid] = incl;
        for (int q = tid; q < CELLS / 2; q += BT) reinterpret_cast<int*>(celln)[q] = -1;
        if (tid == 0) { s_kmin = ~0ull; s_kmax = 0; s_nlist = 0; s_nrev = 0; }
        if (wid == NWAVE - 1 && t + 1 < T) make_row(t + 1, pn0, pn1, pnb);
        __syncthreads();

        // ---- F3: the threshold bin
        int total = 0, wbase = 0;
        for (int w = 0; w < NWAVE; ++w) { const int v = wtot[w]; total += v; if (w < wid) wbase += v; }
        const int exb = wbase + incl - hv;
        if (total <= BW) { if (tid == 0) { s_bstar = NBINS; s_r = 0; s_mb = 0; } }
        else if (exb < BW && exb + hv >= BW) { s_bstar = tid; s_r = BW - exb; s_mb = hv; }
        __syncthreads();

        // ---- F4: exact ranking inside the threshold bin; survivors numbered per wave
        const int bstar = s_bstar, rr = s_r, mb = s_mb;
        bool surv[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) surv[k] = binv[k] < bstar || (binv[k] == bstar && rr == mb);
        if (bstar < NBINS && rr != mb) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (binv[k] == bstar) {
                    const int at = atomicAdd(&s_nlist, 1);
                    if (at < LISTCAP) { l_key[at] = key[k]; l_idx[at] = tid + k * BT; }
                }
            __syncthreads();
            const int nl = s_nlist;
            auto label_of = [&](int idx) { if (idx < nb) return ech[idx]; const int p = idx - nb; const int i = min((int)(((float)p + 0.5f) * invC), nb - 1); return p - i * C; };
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (binv[k] != bstar) continue;
                const int idx = tid + k * BT, myc = label_of(idx);
                const uint64_t mk = key[k];
                int rank = 0;
                if (nl <= LISTCAP) {
                    for (int q = 0; q < nl; ++q) {
                        const uint64_t ok = l_key[q]; const int oi = l_idx[q];
                        if (oi == idx) continue;
                        if (ok > mk) { ++rank; continue; }
                        if (ok == mk) { const int oc = label_of(oi); rank += (oc < myc) || (oc == myc && oi < idx); }
                    }
                    ++dbg_list;
                } else {
                    for (int q = 0; q < N; ++q) {
                        const uint64_t ok = c_key[q];
                        if (ok == 0 || q == idx || score_bin<NBINS>(U, unokey(ok)) != bstar) continue;
                        if (ok > mk) { ++rank; continue; }
                        if (ok == mk) { const int oc = label_of(q); rank += (oc < myc) || (oc == myc && q < idx); }
                    }
                    ++dbg_full;
                }
                surv[k] = rank < rr;
            }
        }
        int within[KMAX], fwithin[KMAX];
        int wcount = 0, fcount = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned long long bs = __ballot(surv[k]), bf = __ballot(surv[k] && mark[k] == 1);
            within[k] = wcount + __builtin_popcountll(bs & lt_mask); wcount += __builtin_popcountll(bs);
            fwithin[k] = fcount + __builtin_popcountll(bf & lt_mask); fcount += __builtin_popcountll(bf);
        }
        if (lane == 0) { wsurv[wid] = wcount; wfresh[wid] = fcount; }
        __syncthreads();

        // ---- F5: slot numbers of the next beam
        int sbase = 0, fbase = 0, nnext = 0, nfresh = 0;
        for (int w = 0; w < NWAVE; ++w) { const int v = wsurv[w], f = wfresh[w]; nnext += v; nfresh += f; if (w < wid) { sbase += v; fbase += f; } }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * BT;
            if (idx < nb) newslot[idx] = surv[k] ? sbase + within[k] : -1;
            if (surv[k]) {
                atomicMin((unsigned long long*)&s_kmin, (unsigned long long)key[k]);
                atomicMax((unsigned long long*)&s_kmax, (unsigned long long)key[k]);
                if (mark[k] >= 2) {
                    const int p = idx - nb; const int i = min((int)(((float)p + 0.5f) * invC), nb - 1), c = p - i * C;
                    cellc[i * C + c] = (short)(-2 - (sbase + within[k]));       // this edge's dormant top is a beam entry again
                    s_nrev = 1;
                }
            }
        }
        __syncthreads();

        // ---- F6: commit
        if (s_nrev) {     // the walk below reads node records other waves have stored: have every store land first
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // an entry's edge tuple in the new slot numbers: through ancestors that left the beam, under a top that came back
        auto resolve = [&](Tup tp, int node_j, int walk_from, int walk_depth, int ch_j, double own_j) -> Tup {
            while (tp.up >= 0) {
                const int m = cellc[tp.up * C + tp.upch];
                if (m <= -2 && tp.upnode != node_j) {
                    int n = walk_from;
                    const int hops = walk_depth - depth[tp.up] - 2;
                    for (int h = 0; h < hops; ++h) n = nodes[n].parent;
                    dbg_hops += hops > 0 ? hops : 0;
                    Tup r; r.up = -2 - m; r.upnode = n;
                    if (n == node_j) { r.upch = ch_j; r.uplpc = own_j; }
                    else { const NodeRec x = nodes[n]; r.upch = x.ch; r.uplpc = x.lpc; }
                    return r;
                }
                const int ns = newslot[tp.up];
                if (ns >= 0) { tp.up = ns; return tp; }
                const int P = tp.up;                         // P left the beam: it is part of this edge now
                tp.up = up[P]; tp.upch = upch[P]; tp.upnode = upnode[P]; tp.uplpc = uplpc[P];
            }
            return Tup{-1, -1, -1, -INFINITY};
        };
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (!surv[k]) continue;
            const int idx = tid + k * BT, sl = sbase + within[k];
            const double lg = unokey(key[k]);
            const size_t o = (size_t)nxt * BW + sl;
            int* nctxp = e_ctx + o * MAXCTX;
            if (mark[k] == 0) {                                  // an entry that stays
                const int j = idx;
                const bool direct = upnode[j] == node[j];
                const double own = direct ? uplpc[j] : ownlpc[j];
                const Tup tp = resolve(Tup{up[j], upch[j], upnode[j], uplpc[j]}, node[j], node[j], depth[j], ech[j], own);
                e_node[o] = node[j]; e_ch[o] = ech[j]; e_ds[o] = eds[j]; e_depth[o] = depth[j]; e_memo[o] = memo[j];
                e_bprev[o] = e_bcur[j]; e_nbprev[o] = e_nbcur[j]; e_score[o] = lg; e_ownlpc[o] = own;
                e_up[o] = tp.up; e_upch[o] = tp.upch; e_upnode[o] = tp.upnode; e_uplpc[o] = tp.uplpc;
                for (int q = 0; q < MAXCTX; ++q) nctxp[q] = ctx[(size_t)j * MAXCTX + q];
                if (tp.up >= 0) celln[tp.up * C + tp.upch] = (short)sl;
                continue;
            }
            const int p = idx - nb;
            const int i = min((int)(((float)p + 0.5f) * invC), nb - 1), c = p - i * C;
            const bool sp = a.has_lm && c == a.space;
            int id; double lpc;
            if (mark[k] == 1) {                                  // a new node
                id = nn + fbase + fwithin[k];
                lpc = lpv[c];
                NodeRec r; r.parent = node[i]; r.ch = c; r.tstep = t; r.depth = depth[i] + 1; r.lpc = lpc; r.pad = 0.0;
                nodes[id] = r;
            } else {                                             // a dormant prefix is back: same node, its log_prob_c kept
                const int rep = mark[k] - 2;
                id = upnode[rep]; lpc = uplpc[rep];
                ++dbg_rev;
            }
            int nds = 0;
            if (a.has_lm && !sp) nds = mark[k] == 1 ? arcv[k] : a.trie_next[(size_t)eds[i] * C + c];
            const int* pctx = ctx + (size_t)i * MAXCTX;
            for (int q = 0; q < MAXCTX; ++q) {
                int v = pctx[q];
                if (sp) v = q + 1 < NCTX ? pctx[q + 1] : (q + 1 == NCTX ? arcv[k] : a.bos);
                nctxp[q] = v;
            }
            Tup tp{newslot[i], c, id, lpc};
            if (tp.up < 0) tp = resolve(Tup{up[i], upch[i], upnode[i], uplpc[i]}, id, node[i], depth[i], c, lpc);
            e_node[o] = id; e_ch[o] = c; e_ds[o] = nds; e_depth[o] = depth[i] + 1; e_memo[o] = MEMO_UNSET;
            e_bprev[o] = -INFINITY; e_nbprev[o] = lg; e_score[o] = lg; e_ownlpc[o] = lpc;
            e_up[o] = tp.up; e_upch[o] = tp.upch; e_upnode[o] = tp.upnode; e_uplpc[o] = tp.uplpc;
            if (tp.up >= 0) celln[tp.up * C + tp.upch] = (short)sl;
        }
        nn += nfresh;
        if (tid == 0) s_nb = nnext;
        __syncthreads();
        cur = nxt;
    }

    // ---- final: trailing partial word, order, write out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int nb = s_nb;
        double* score = e_score + cur * BW; const int* node = e_node + cur * BW; const int* ech = e_ch + cur * BW;
        const int* eds = e_ds + cur * BW; const int* depth = e_depth + cur * BW; const int* ctx = e_ctx + (size_t)cur * BW * MAXCTX;
        if (a.has_lm) {
            for (int i = tid; i < nb; i += BT) {
                if (node[i] != 0 && ech[i] != a.space) {
                    const int w = a.trie_word[eds[i]];
                    double lm = kOovScore;
                    if (w >= 0) lm = (double)lm_word_log10(s_lm, ctx + (size_t)i * MAXCTX, NCTX, w, a.unk) / (double)kLog10E;
                    double s = lm * a.alpha;
                    s += a.beta;
                    score[i] += s;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < nb; i += BT) {
            int rank = 0;
            for (int j = 0; j < nb; ++j) {
                if (j == i) continue;
                rank += score[j] > score[i] || (score[j] == score[i] && (ech[j] < ech[i] || (ech[j] == ech[i] && j < i)));
            }
            const int len = depth[i];
            const size_t o = ((size_t)b * BW + rank) * a.T;
            int n = node[i];
            for (int k = len - 1; k >= 0; --k) { const NodeRec r = nodes[n]; a.out_tok[o + k] = r.ch; a.out_step[o + k] = r.tstep; n = r.parent; }
            a.out_len[(size_t)b * BW + rank] = len;
            a.out_score[(size_t)b * BW + rank] = score[i];
        }
        if (tid == 0) a.out_n[b] = nb;
    }
    if (a.dbg) {
        dbg_rev = wave_sum(dbg_rev); dbg_hops = wave_sum(dbg_hops); dbg_list = wave_sum(dbg_list); dbg_full = wave_sum(dbg_full);
        if (lane == 0) { atomicAdd(a.dbg + 4 * b, dbg_rev); atomicAdd(a.dbg + 4 * b + 1, dbg_hops); atomicAdd(a.dbg + 4 * b + 2, dbg_list); atomicAdd(a.dbg + 4 * b + 3, dbg_full); }
    }
#undef s_nb
#undef s_bstar
#undef s_r
#undef s_mb
#undef s_nlist
#undef s_nrev
#undef s_kmin
#undef s_kmax
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct dsmi_decoder {
    int device = 0;
    std::vector<std::string> labels;
    int blank = 0, space = -2;
    std::string err;
    // greedy scratch
    size_t greedy_cap = 0;
    int32_t *g_raw = nullptr, *g_ids = nullptr, *g_offs = nullptr, *g_nout = nullptr, *g_sizes = nullptr;
    // LM
    bool has_lm = false;
    HostLM lm;
    double alpha = 0, beta = 0;
    LmEntry* d_tab = nullptr; int32_t *d_next = nullptr, *d_word = nullptr;
    unsigned char* d_klm = nullptr;      // device copy of a KenLM probing binary's search memory
    // beam workspace
    size_t ws_bytes = 0;
    unsigned char* ws = nullptr;
    // the beam search between dsmi_beam_enqueue and dsmi_beam_collect: geometry, pinned copies of its outputs, completion event
    bool beam_pending = false;
    int pb_B = 0, pb_To = 0, pb_beam = 0;
    size_t pb_out = 0, pb_out_bytes = 0; hipStream_t pb_stream = nullptr;
    int32_t stats[4] = {0, 0, 0, 0};      // of the last collected search: see dsmi_decoder_beam_stats
    int32_t* pin_sz = nullptr; size_t pin_sz_cap = 0;
    unsigned char* pin = nullptr; size_t pin_bytes = 0;
    hipEvent_t beam_done = nullptr;
};

static thread_local std::string g_dec_error;

#define DEC_HIP(d, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { (d)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return DSMI_ERR_HIP; } \
    } while (0)

extern "C" int dsmi_decoder_create(int device, const char* const* labels, int n_labels, int blank, dsmi_decoder** out) {
    if (!labels || !out || n_labels < 1 || n_labels > 128 || blank < 0 || blank >= n_labels) { g_dec_error = "bad decoder arguments"; return DSMI_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_dec_error = "no such HIP device"; return DSMI_ERR_HIP; }
    dsmi_decoder* d = new dsmi_decoder();
    d->device = device;
    d->blank = blank;
    for (int i = 0; i < n_labels; ++i) {
        d->labels.push_back(labels[i] ? labels[i] : "");
        if (d->labels.back() == " ") d->space = i;     // Decoder.__init__, decoder.py:39-42
    }
    *out = d;
    return DSMI_OK;
}

static void free_lm(dsmi_decoder* d) {
    if (d->d_tab) (void)hipFree(d->d_tab);
    if (d->d_next) (void)hipFree(d->d_next);
    if (d->d_word) (void)hipFree(d->d_word);
    if (d->d_klm) (void)hipFree(d->d_klm);
    d->d_tab = nullptr; d->d_next = nullptr; d->d_word = nullptr; d->d_klm = nullptr;
    d->has_lm = false;
    d->lm = HostLM();
}

extern "C" void dsmi_decoder_destroy(dsmi_decoder* d) {
    if (d && d->pin) { (void)hipSetDevice(d->device); (void)hipDeviceSynchronize(); (void)hipHostFree(d->pin); d->pin = nullptr; }
    if (d && d->beam_done) { (void)hipEventDestroy(d->beam_done); d->beam_done = nullptr; }
    if (d && d->pin_sz) { (void)hipHostFree(d->pin_sz); d->pin_sz = nullptr; }
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)d->g_raw, (void*)d->g_ids, (void*)d->g_offs, (void*)d->g_nout, (void*)d->g_sizes, (void*)d->ws}) if (p) (void)hipFree(p);
    free_lm(d);
    delete d;
}

extern "C" const char* dsmi_decoder_last_error(const dsmi_decoder* d) { return d ? d->err.c_str() : g_dec_error.c_str(); }

extern "C" int dsmi_decoder_info(const dsmi_decoder* d, int* n_labels, int* blank_index, int* device) {
    if (!d) return DSMI_ERR_INVALID;
    if (n_labels) *n_labels = (int)d->labels.size();
    if (blank_index) *blank_index = d->blank;
    if (device) *device = d->device;
    return DSMI_OK;
}

extern "C" const char* dsmi_decoder_label(const dsmi_decoder* d, int index) {
    return d && index >= 0 && index < (int)d->labels.size() ? d->labels[(size_t)index].c_str() : nullptr;
}

extern "C" int dsmi_decoder_set_lm(dsmi_decoder* d, const char* path, double alpha, double beta) {
    if (!d) return DSMI_ERR_INVALID;
    DEC_HIP(d, hipSetDevice(d->device));
    DEC_HIP(d, hipDeviceSynchronize());
    free_lm(d);
    d->alpha = alpha; d->beta = beta;
    if (!path || !*path) return DSMI_OK;
    const std::string msg = d->lm.load(path, d->labels);         // ARPA text or KenLM binary (probing / trie)
    if (!msg.empty()) { d->lm = HostLM(); d->err = msg; return DSMI_ERR_IO; }
    if (d->lm.order > kMaxOrder) { d->err = "n-gram order above 6 is not supported"; d->lm = HostLM(); return DSMI_ERR_IO; }
    if (d->lm.kind == 1) {
        DEC_HIP(d, hipMalloc((void**)&d->d_klm, d->lm.klm_blob.size()));
        DEC_HIP(d, hipMemcpy(d->d_klm, d->lm.klm_blob.data(), d->lm.klm_blob.size(), hipMemcpyHostToDevice));
    }
    DEC_HIP(d, hipMalloc((void**)&d->d_tab, sizeof(LmEntry) * d->lm.table.size()));
    DEC_HIP(d, hipMalloc((void**)&d->d_next, sizeof(int32_t) * d->lm.trie_next.size()));
    DEC_HIP(d, hipMalloc((void**)&d->d_word, sizeof(int32_t) * d->lm.trie_word.size()));
    DEC_HIP(d, hipMemcpy(d->d_tab, d->lm.table.data(), sizeof(LmEntry) * d->lm.table.size(), hipMemcpyHostToDevice));
    DEC_HIP(d, hipMemcpy(d->d_next, d->lm.trie_next.data(), sizeof(int32_t) * d->lm.trie_next.size(), hipMemcpyHostToDevice));
    DEC_HIP(d, hipMemcpy(d->d_word, d->lm.trie_word.data(), sizeof(int32_t) * d->lm.trie_word.size(), hipMemcpyHostToDevice));
    d->has_lm = true;
    return DSMI_OK;
}

extern "C" int dsmi_greedy(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To,
                           int32_t* ids, int32_t* offsets, int32_t* n_out, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    if (!probs || !ids || !offsets || !n_out || B < 1 || To < 1) { d->err = "bad greedy arguments"; return DSMI_ERR_INVALID; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)B * To > d->greedy_cap) {
        DEC_HIP(d, hipDeviceSynchronize());
        for (void* p : {(void*)d->g_raw, (void*)d->g_ids, (void*)d->g_offs, (void*)d->g_nout, (void*)d->g_sizes}) if (p) (void)hipFree(p);
        d->greedy_cap = (size_t)B * To;
        DEC_HIP(d, hipMalloc((void**)&d->g_raw, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_ids, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_offs, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_nout, sizeof(int32_t) * d->greedy_cap));
        DEC_HIP(d, hipMalloc((void**)&d->g_sizes, sizeof(int32_t) * d->greedy_cap));
    }
    if (sizes) DEC_HIP(d, hipMemcpyAsync(d->g_sizes, sizes, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    launch_greedy(probs, sizes ? d->g_sizes : nullptr, B, To, (int)d->labels.size(), d->blank, d->g_raw, d->g_ids, d->g_offs, d->g_nout, s);
    DEC_HIP(d, hipMemcpyAsync(ids, d->g_ids, sizeof(int32_t) * (size_t)B * To, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(offsets, d->g_offs, sizeof(int32_t) * (size_t)B * To, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipMemcpyAsync(n_out, d->g_nout, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    DEC_HIP(d, hipStreamSynchronize(s));
    DEC_HIP(d, hipGetLastError());
    return DSMI_OK;
}

// Launches the beam search of a batch, asynchronous on `stream`; dsmi_beam_collect copies the results.
extern "C" int dsmi_beam_enqueue(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, int beam, int cutoff_top_n,
                                 double cutoff_prob, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    const int C = (int)d->labels.size();
    if (!probs || B < 1 || To < 1 || beam < 1) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    if (d->beam_pending) { d->err = "the previous beam search has not been collected"; return DSMI_ERR_INVALID; }
    const size_t NMAX = (size_t)beam * (C + 1);
    static const int BT = getenv("DSMI_BEAM_THREADS") ? atoi(getenv("DSMI_BEAM_THREADS")) : 1024;      // (experiment switch)
    const size_t lds = carve(beam, C, BT).bytes;
    if (lds > 160 * 1024 - 256 || NMAX > (size_t)(BT == 1024 ? 9 : 17) * BT || beam > BT || C > MAXC) { d->err = "beam_width * (n_labels + 1) exceeds the on-chip candidate buffer"; return DSMI_ERR_CAPACITY; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    // ---- workspace carve
    const int ncap = 2 + To * beam;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_nodes = off; off += al((size_t)B * ncap * sizeof(NodeRec));
    size_t o_sizes = off; off += al((size_t)B * 4);
    // outputs, contiguous, in the order of the pinned image: tokens, steps, lens, counts, scores
    const size_t o_out = off;
    size_t o_tok = off; off += al((size_t)B * beam * To * 4);
    size_t o_step = off; off += al((size_t)B * beam * To * 4);
    size_t o_len = off; off += al((size_t)B * beam * 4);
    size_t o_n = off; off += al((size_t)B * 4);
    size_t o_score = off; off += al((size_t)B * beam * 8);
    size_t o_dbg = off; off += al((size_t)B * 16);
    const size_t out_bytes = off - o_out;
    if (off > d->ws_bytes) {
        DEC_HIP(d, hipDeviceSynchronize());
        if (d->ws) (void)hipFree(d->ws);
        d->ws = nullptr; d->ws_bytes = 0;
        DEC_HIP(d, hipMalloc((void**)&d->ws, off));
        d->ws_bytes = off;
    }
    if (out_bytes > d->pin_bytes) {
        DEC_HIP(d, hipDeviceSynchronize());
        if (d->pin) (void)hipHostFree(d->pin);
        d->pin = nullptr; d->pin_bytes = 0;
        DEC_HIP(d, hipHostMalloc((void**)&d->pin, out_bytes + out_bytes / 4, hipHostMallocDefault));
        d->pin_bytes = out_bytes + out_bytes / 4;
    }
    if (!d->beam_done) DEC_HIP(d, hipEventCreateWithFlags(&d->beam_done, hipEventDisableTiming));
    unsigned char* w = d->ws;
    BeamArgs a{};
    a.probs = probs; a.T = To; a.C = C; a.blank = d->blank; a.space = d->space; a.beam = beam;
    a.cutoff_top_n = cutoff_top_n; a.cutoff_prob = (float)cutoff_prob;
    a.has_lm = d->has_lm ? 1 : 0; a.order = d->has_lm ? d->lm.order : 1; a.alpha = d->alpha; a.beta = d->beta;
    a.lm = d->lm.view(); a.lm.tab = d->d_tab; a.lm.klm.base = d->d_klm; a.trie_next = d->d_next; a.trie_word = d->d_word; a.unk = d->lm.unk; a.bos = d->lm.bos;
    a.ncap = ncap;
    a.nodes = (NodeRec*)(w + o_nodes); a.dbg = (int32_t*)(w + o_dbg);
    a.out_tok = (int32_t*)(w + o_tok); a.out_step = (int32_t*)(w + o_step); a.out_len = (int32_t*)(w + o_len); a.out_n = (int32_t*)(w + o_n);
    a.out_score = (double*)(w + o_score);
    a.sizes = nullptr;
    if (sizes) {
        // through pinned memory: an asynchronous copy from pageable memory makes the HOST wait for the stream to get there
        // (i.e. for the forward this search was queued behind), and the caller's array need not outlive this call
        if ((size_t)B > d->pin_sz_cap) {
            if (d->pin_sz) { DEC_HIP(d, hipDeviceSynchronize()); (void)hipHostFree(d->pin_sz); d->pin_sz = nullptr; d->pin_sz_cap = 0; }
            DEC_HIP(d, hipHostMalloc((void**)&d->pin_sz, sizeof(int32_t) * std::max(B, 256), hipHostMallocDefault));
            d->pin_sz_cap = (size_t)std::max(B, 256);
        }
        std::memcpy(d->pin_sz, sizes, sizeof(int32_t) * B);
        DEC_HIP(d, hipMemcpyAsync(w + o_sizes, d->pin_sz, sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
        a.sizes = (const int32_t*)(w + o_sizes);
    }
    DEC_HIP(d, hipMemsetAsync(w + o_len, 0, (size_t)B * beam * 4, s));
    DEC_HIP(d, hipMemsetAsync(w + o_dbg, 0, (size_t)B * 16, s));
    // candidates per thread: a compile-time bound, so that they live in registers
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(B), dim3(BT), lds, s, a);
        return hipSuccess;
    };
    if (BT == 1024) {
        if (NMAX <= (size_t)3 * BT) DEC_HIP(d, launch(beam_kernel<1024, 3>));
        else if (NMAX <= (size_t)5 * BT) DEC_HIP(d, launch(beam_kernel<1024, 5>));
        else DEC_HIP(d, launch(beam_kernel<1024, 9>));
    } else {
        if (NMAX <= (size_t)5 * BT) DEC_HIP(d, launch(beam_kernel<512, 5>));
        else if (NMAX <= (size_t)9 * BT) DEC_HIP(d, launch(beam_kernel<512, 9>));
        else DEC_HIP(d, launch(beam_kernel<512, 17>));
    }
    DEC_HIP(d, hipGetLastError());
    // Only the kernel is queued here.  A device-to-host copy queued behind it would sit in a DMA queue until the search is
    // over, and with it whatever upload another stream has been given the same engine for -- the next batch's samples: its
    // forward then starts only when this search has ended (measured: two batches in flight ran one after the other).
    DEC_HIP(d, hipEventRecord(d->beam_done, s));
    d->beam_pending = true; d->pb_B = B; d->pb_To = To; d->pb_beam = beam; d->pb_out = o_out; d->pb_out_bytes = out_bytes; d->pb_stream = s;
    return DSMI_OK;
}

// Waits for the enqueued beam search and hands its results over (layouts of dsmi_beam).
extern "C" int dsmi_beam_collect(dsmi_decoder* d, int32_t* tokens, int32_t* tsteps, int32_t* lens, float* scores) {
    if (!d) return DSMI_ERR_INVALID;
    if (!d->beam_pending) { d->err = "no beam search enqueued"; return DSMI_ERR_INVALID; }
    if (!tokens || !tsteps || !lens || !scores) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    d->beam_pending = false;
    DEC_HIP(d, hipSetDevice(d->device));
    DEC_HIP(d, hipEventSynchronize(d->beam_done));
    const int B = d->pb_B, To = d->pb_To, beam = d->pb_beam;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t tok_bytes = al((size_t)B * beam * To * 4);
    // lengths, counts and scores first (they are behind the two token arrays in the image) ...
    const unsigned char* dev = d->ws + d->pb_out;
    // (on the search's own stream, which is idle once the search is over: a blocking copy would go to the null stream and
    // wait there for whatever forward the caller has queued meanwhile)
    DEC_HIP(d, hipMemcpyAsync(d->pin + 2 * tok_bytes, dev + 2 * tok_bytes, d->pb_out_bytes - 2 * tok_bytes, hipMemcpyDeviceToHost, d->pb_stream));
    DEC_HIP(d, hipStreamSynchronize(d->pb_stream));
    const unsigned char* q0 = d->pin;
    const int32_t* p_tok = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* p_step = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* h_len = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * beam * 4);
    const int32_t* h_n = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * 4);
    const double* h_score = reinterpret_cast<const double*>(q0); q0 += al((size_t)B * beam * 8);
    const int32_t* h_dbg = reinterpret_cast<const int32_t*>(q0);
    for (int k = 0; k < 4; ++k) { d->stats[k] = 0; for (int b = 0; b < B; ++b) d->stats[k] += h_dbg[4 * b + k]; }
    // ... then, of the token arrays, only the columns that hold tokens: transcripts are a fraction of T_out long
    int maxlen = 0;
    for (size_t i = 0; i < (size_t)B * beam; ++i) maxlen = std::max(maxlen, (int)h_len[i]);
    if (maxlen > 0) {
        DEC_HIP(d, hipMemcpy2DAsync(d->pin, (size_t)To * 4, dev, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost, d->pb_stream));
        DEC_HIP(d, hipMemcpy2DAsync(d->pin + tok_bytes, (size_t)To * 4, dev + tok_bytes, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost, d->pb_stream));
        DEC_HIP(d, hipStreamSynchronize(d->pb_stream));
    }
    for (size_t r = 0; r < (size_t)B * beam && maxlen > 0; ++r) {
        std::memcpy(tokens + r * To, p_tok + r * To, (size_t)maxlen * 4);
        std::memcpy(tsteps + r * To, p_step + r * To, (size_t)maxlen * 4);
    }
    // ---- ctcdecode "approx_ctc": strip the word bonus and the LM weight; score = -approx
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < beam; ++p) {
            const size_t q = (size_t)b * beam + p;
            if (p >= h_n[b]) { lens[q] = 0; scores[q] = 0.f; continue; }
            lens[q] = h_len[q];
            double approx = h_score[q];
            if (d->has_lm) {
                std::vector<int32_t> words;
                std::string cur;
                auto flush = [&]() {
                    if (cur.empty()) return;
                    auto it = d->lm.word2id.find(cur);
                    words.push_back(it == d->lm.word2id.end() ? -1 : it->second);
                    cur.clear();
                };
                const int32_t* tk = p_tok + q * To;
                for (int k = 0; k < h_len[q]; ++k) { if (tk[k] == d->space) flush(); else cur += d->labels[tk[k]]; }
                flush();
                approx -= (double)h_len[q] * d->beta;
                approx -= d->lm.sent_ln(words) * d->alpha;
            }
            scores[q] = (float)-approx;
        }
    return DSMI_OK;
}

extern "C" int dsmi_decoder_beam_stats(const dsmi_decoder* d, int32_t* counts4) {
    if (!d || !counts4) return DSMI_ERR_INVALID;
    for (int k = 0; k < 4; ++k) counts4[k] = d->stats[k];
    return DSMI_OK;
}

extern "C" int dsmi_beam(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, int beam, int cutoff_top_n,
                         double cutoff_prob, int32_t* tokens, int32_t* tsteps, int32_t* lens, float* scores, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    if (!tokens || !tsteps || !lens || !scores) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    const int rc = dsmi_beam_enqueue(d, probs, sizes, B, To, beam, cutoff_top_n, cutoff_prob, stream);
    if (rc) return rc;
    return dsmi_beam_collect(d, tokens, tsteps, lens, scores);
}


// ---- host-only view of a language model file (no GPU needed): lets callers and the CPU tests inspect what
// dsmi_decoder_set_lm would load.  See include/dsmi.h.
struct dsmi_lm { dsmi::HostLM lm; std::string err; };
static thread_local std::string g_lm_error;

extern "C" int dsmi_lm_open(const char* path, dsmi_lm** out) {
    if (!path || !out) { g_lm_error = "null argument"; return DSMI_ERR_INVALID; }
    dsmi_lm* h = new dsmi_lm();
    const std::string msg = h->lm.load(path, std::vector<std::string>());
    if (!msg.empty()) { g_lm_error = msg; delete h; return DSMI_ERR_IO; }
    *out = h;
    return DSMI_OK;
}
extern "C" void dsmi_lm_close(dsmi_lm* h) { delete h; }
extern "C" const char* dsmi_lm_last_error(const dsmi_lm* h) { return h ? h->err.c_str() : g_lm_error.c_str(); }
extern "C" int dsmi_lm_info(const dsmi_lm* h, int* order, int64_t* vocab_size, int* kind) {
    if (!h) return DSMI_ERR_INVALID;
    if (order) *order = h->lm.order;
    if (vocab_size) *vocab_size = (int64_t)h->lm.vocab.size();
    if (kind) *kind = h->lm.kind;
    return DSMI_OK;
}
extern "C" int dsmi_lm_word_index(const dsmi_lm* h, const char* word_utf8) {
    if (!h || !word_utf8) return DSMI_ERR_INVALID;
    auto it = h->lm.word2id.find(word_utf8);
    return it == h->lm.word2id.end() ? -1 : it->second;
}
extern "C" int dsmi_lm_lookup(const dsmi_lm* h, const int32_t* ids, int n, float* log10_prob, float* log10_backoff) {
    if (!h || !ids || n < 1 || n > h->lm.order) return DSMI_ERR_INVALID;
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= (int32_t)h->lm.vocab.size()) return DSMI_ERR_INVALID;
    float lp = 0.f, bo = 0.f;
    const bool found = dsmi::lm_lookup(h->lm.view(), ids, n, &lp, &bo);
    if (log10_prob) *log10_prob = lp;
    if (log10_backoff) *log10_backoff = bo;
    return found ? 1 : 0;
}
extern "C" double dsmi_lm_cond_log10(const dsmi_lm* h, const int32_t* ids, int n) {
    if (!h || !ids || n < 1 || n > h->lm.order) return 0.0 / 0.0;
    return (double)dsmi::lm_cond_log10(h->lm.view(), ids, n - 1, ids[n - 1], h->lm.unk);
}
