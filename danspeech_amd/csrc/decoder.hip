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

using namespace dsmi;

namespace {

constexpr int BT = 1024;          // threads per utterance: the frame loop is latency-bound, so 16 waves hide it
constexpr int NWAVE = BT / 64;
constexpr int MAXCTX = kMaxOrder - 1;
constexpr int F_EXISTS = 1, F_DELETED = 2;
constexpr int MEMO_UNSET = 0x7fc00001;   // a NaN pattern no float computation produces

// One prefix-trie node (64 bytes: one L2 line).  Children are found through a direct table
// childtab[node][label] (-1: none), so a pair needs two dependent loads: the child id, then its record.
struct __attribute__((aligned(64))) Node {
    int parent, ch, tstep, dstate;
    int nchild, flags, slot, memo;       // memo: bits of log10 P_lm(word ending here | ctx), or MEMO_UNSET
    double lpc;
    int ctx[MAXCTX];
    int pad;
};
static_assert(sizeof(Node) == 64, "Node must be one 64-byte line");

struct BeamArgs {
    const float* probs; const int32_t* sizes; int T, C, blank, space, beam, cutoff_top_n; float cutoff_prob;
    int has_lm, order; double alpha, beta;
    LmView lm; const int32_t* trie_next; const int32_t* trie_word; int unk, bos;
    int ncap;                    // per-utterance node pool capacity
    Node* nodes; int32_t* childtab; int32_t* nnodes;
    // outputs
    int32_t *out_tok, *out_step, *out_len, *out_n; double* out_score;
};

__device__ __forceinline__ double lse2(double x, double y) {
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    const double m = fmax(x, y);
    return log(exp(x - m) + exp(y - m)) + m;
}

__device__ __forceinline__ uint64_t okey(double v) {   // order-preserving bits, > 0 for every double
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unokey(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(BT) void beam_kernel(BeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int C = a.C, BW = a.beam, NMAX = BW * (C + 1);
    // ---- LDS carve
    double* lp = reinterpret_cast<double*>(smem_raw);                 // [128]
    double* sd = lp + 128;                                            // [2] scalars (min_cutoff)
    double* e_bprev = sd + 2;                                         // entry arrays [2][BW] (double-buffered)
    double* e_nbprev = e_bprev + 2 * BW;
    double* e_score = e_nbprev + 2 * BW;
    double* e_bcur = e_score + 2 * BW;                                // [BW]
    double* e_rep = e_bcur + BW;
    double* e_ext = e_rep + BW;
    uint64_t* c_key = reinterpret_cast<uint64_t*>(e_ext + BW);        // [NMAX] 0 = no candidate, else okey(logp)
    uint64_t* su = c_key + NMAX;                                      // [2] scalars (radix prefix)
    uint64_t* wkmin = su + 2;                                         // [NWAVE] per-wave smallest / largest candidate key
    uint64_t* wkmax = wkmin + NWAVE;
    int* e_node = reinterpret_cast<int*>(wkmax + NWAVE);              // [2][BW]
    int* e_ch = e_node + 2 * BW;                                      // [2][BW]
    int* e_ds = e_ch + 2 * BW;                                        // [2][BW] dictionary state of the entry's node
    int* c_child = e_ds + 2 * BW;                                     // [NMAX]
    int* c_surv = c_child + NMAX;                                     // [NMAX]
    int* newn = c_surv + NMAX;                                        // [BW] nodes created this frame
    int* use = newn + BW;                                             // [128]
    unsigned* hist = reinterpret_cast<unsigned*>(use + 128);          // [2][256]
    int* wtot = reinterpret_cast<int*>(hist + 512);                   // [NWAVE + 1]
    int* si = wtot + NWAVE + 1;                                       // [8] scalars
#define s_nb si[0]
#define s_full si[1]
#define s_m si[2]
#define s_nuse si[3]
#define s_kk si[4]
#define s_done si[5]
#define s_nnew si[6]
#define s_pass0 si[7]
#define s_mincut sd[0]
#define s_prefix su[0]

    // ---- per-utterance global state
    Node* nodes = a.nodes + (size_t)b * a.ncap;
    int32_t* childtab = a.childtab + (size_t)b * a.ncap * C;
    int32_t* nnodes = a.nnodes + b;
    const int T = a.sizes ? min(a.sizes[b], a.T) : a.T;
    const float* pb = a.probs + (size_t)b * a.T * C;
    const int NCTX = a.order - 1;

    if (tid < C) childtab[tid] = -1;
    if (tid == 0) {
        Node r{};
        r.parent = -1; r.ch = -1; r.tstep = 0; r.dstate = 0; r.nchild = 0; r.flags = F_EXISTS; r.slot = 0; r.memo = MEMO_UNSET;
        r.lpc = -INFINITY;
        for (int k = 0; k < MAXCTX; ++k) r.ctx[k] = a.bos;
        nodes[0] = r;
        *nnodes = 1;
        e_node[0] = 0; e_ch[0] = -1; e_ds[0] = 0; e_bprev[0] = 0.0; e_nbprev[0] = -INFINITY; e_score[0] = 0.0;
        s_nb = 1;
    }
    float pnext = (tid < C && T > 0) ? pb[tid] : 0.f;     // next frame's row, fetched one frame ahead
    __syncthreads();
    int cur = 0;

    for (int t = 0; t < T; ++t) {
        const int nb = s_nb;
        double* bprev = e_bprev + cur * BW; double* nbprev = e_nbprev + cur * BW; double* score = e_score + cur * BW;
        int* node = e_node + cur * BW; int* ech = e_ch + cur * BW; int* eds = e_ds + cur * BW;
        const float* pr = pb + (size_t)t * C;
        // ---- 1. log-probabilities and vocabulary pruning (decoder_utils get_pruned_log_probs)
        if (tid < C) { lp[tid] = log((double)pnext + 1.17549435e-38); use[tid] = 1; }
        if (tid < C && t + 1 < T) pnext = pr[C + tid];
        for (int i = tid; i < nb; i += BT) { e_bcur[i] = -INFINITY; e_rep[i] = -INFINITY; e_ext[i] = -INFINITY; }
        if (wid == 1) {
            // ---- min_cutoff / full_beam (scorer only)
            double mn = INFINITY;
            if (a.has_lm) {
                for (int i = lane; i < nb; i += 64) mn = fmin(mn, score[i]);
                for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o, 64));
            }
            if (lane == 0) {
                s_full = 0; s_mincut = -INFINITY;
                if (a.has_lm) {
                    const float pbl = pr[a.blank];
                    const double blp = pbl > 0.f ? log((double)pbl) : -INFINITY;
                    s_mincut = mn + blp - fmax(0.0, a.beta);
                    s_full = nb == BW;
                }
            }
        }
        if (tid == 0) {
            s_nuse = C;
            if (a.cutoff_prob < 1.0f || a.cutoff_top_n < C) {
                // selection by repeated maximum: C is small (<= 128)
                bool used[128];
                for (int c = 0; c < C; ++c) used[c] = false;
                double cum = 0.0; int len = 0;
                const int maxlen = a.cutoff_prob < 1.0f ? C : a.cutoff_top_n;
                while (len < maxlen) {
                    int best = -1; float bv = -1.f;
                    for (int c = 0; c < C; ++c) if (!used[c] && pr[c] > bv) { bv = pr[c]; best = c; }
                    if (best < 0) break;
                    used[best] = true; ++len; cum += (double)bv;
                    if (a.cutoff_prob < 1.0f && (cum >= (double)a.cutoff_prob || len >= a.cutoff_top_n)) break;
                }
                s_nuse = -len;     // negative: the mask below still has to be applied
                for (int c = 0; c < C; ++c) c_surv[c] = used[c];
            }
        }
        __syncthreads();
        if (s_nuse < 0) {
            if (tid < C) use[tid] = c_surv[tid];
            __syncthreads();
        }
        const bool full = s_full != 0;
        const double mincut = s_mincut;

        // ---- 2. all (entry, character) pairs
        const int NP = nb * C;
        for (int idx = tid; idx < NP; idx += BT) {
            const int i = idx / C, c = idx - i * C;
            uint64_t key = 0; int cc = -2;
            do {
                if (!use[c]) break;
                const double l = lp[c], sc = score[i];
                if (full && l + sc < mincut) break;
                if (c == a.blank) { e_bcur[i] = l + sc; break; }
                const int lastc = ech[i];
                if (c == lastc) e_rep[i] = l + nbprev[i];
                const int pn = node[i];
                const int ds = eds[i];
                // independent loads first: child id, dictionary arc, memoised LM score of the parent
                const int child = childtab[(size_t)pn * C + c];
                const bool sp = a.has_lm && c == a.space;
                int arc = 0, memo = 0;
                if (a.has_lm) arc = c == a.space ? a.trie_word[ds] : a.trie_next[(size_t)ds * C + c];
                if (sp) memo = nodes[pn].memo;
                int cflags = F_DELETED, cslot = -1;
                if (child >= 0) {
                    Node* cn = nodes + child;
                    cflags = cn->flags; cslot = cn->slot;
                    if (!(cflags & F_DELETED) && cn->lpc < l) { cn->lpc = l; cn->tstep = t; }
                }
                const bool alive = !(cflags & F_DELETED);
                if (!alive && a.has_lm && arc < 0) break;   // dictionary: a new child needs an arc (space: a word must end here)
                double logp = -INFINITY;
                if (c == lastc) { if (bprev[i] > -INFINITY) logp = l + bprev[i]; }
                else logp = l + sc;
                if (sp) {
                    double lm = kOovScore;
                    if (arc >= 0) {
                        float l10;
                        if (memo != MEMO_UNSET) l10 = __int_as_float(memo);
                        else {
                            l10 = lm_cond_log10(a.lm, nodes[pn].ctx, NCTX, arc, a.unk);
                            nodes[pn].memo = __float_as_int(l10);
                        }
                        lm = (double)l10 / (double)kLog10E;
                    }
                    logp += lm * a.alpha;
                    logp += a.beta;
                }
                if (alive && (cflags & F_EXISTS)) { e_ext[cslot] = logp; break; }
                cc = alive ? child : (child >= 0 ? -3 - child : -1);   // <= -3: reuse deleted id
                key = okey(logp);
            } while (false);
            c_key[idx] = key; c_child[idx] = cc;
        }
        __syncthreads();
        // entries themselves
        for (int i = tid; i < nb; i += BT) {
            const double nbc = lse2(e_rep[i], e_ext[i]);
            e_rep[i] = nbc;                         // now nb_cur
            const double s = lse2(e_bcur[i], nbc);
            c_child[NP + i] = node[i]; c_key[NP + i] = okey(s);
        }
        __syncthreads();
        const int N = NP + nb;

        // ---- 3. exact top-BW selection
        {
            int cnt = 0;
            uint64_t kmin = ~0ull, kmax = 0;
            for (int idx = tid; idx < N; idx += BT) {
                const uint64_t k = c_key[idx];
                if (k != 0) { ++cnt; kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
            }
            cnt = wave_sum(cnt);
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t a0 = __shfl_xor(kmin, o, 64), a1 = __shfl_xor(kmax, o, 64);
                kmin = a0 < kmin ? a0 : kmin; kmax = a1 > kmax ? a1 : kmax;
            }
            if (lane == 0) { wtot[wid] = cnt; wkmin[wid] = kmin; wkmax[wid] = kmax; }
            for (int q = tid; q < 512; q += BT) hist[q] = 0;      // both histograms: the first pass may be an odd one
            __syncthreads();
            if (tid == 0) {
                int m = 0;
                uint64_t lo = ~0ull, hi = 0;
                for (int w = 0; w < NWAVE; ++w) { m += wtot[w]; lo = wkmin[w] < lo ? wkmin[w] : lo; hi = wkmax[w] > hi ? wkmax[w] : hi; }
                // the radix select starts at the first byte in which the candidates differ at all (scores of one frame share
                // sign, exponent and often the leading mantissa bits: the passes over those bytes would select nothing)
                const int skip = (m > 0 && lo != hi) ? __builtin_clzll(lo ^ hi) >> 3 : 0;
                s_m = m; s_prefix = skip ? hi >> (64 - 8 * skip) : 0; s_kk = BW; s_done = 0; s_pass0 = skip;
            }
            __syncthreads();
        }
        const int M = s_m;
        if (M <= BW) {
            for (int idx = tid; idx < N; idx += BT) c_surv[idx] = c_key[idx] != 0;
        } else {
            // radix select, 8 bits per pass from the top; stops as soon as the threshold bin is taken whole
            int pass = s_pass0;
            for (; pass < 8; ++pass) {
                const int shift = 56 - 8 * pass;
                unsigned* h = hist + (pass & 1) * 256;
                unsigned* hn = hist + ((pass + 1) & 1) * 256;
                const uint64_t pre = s_prefix;
                for (int idx = tid; idx < N; idx += BT) {
                    const uint64_t k = c_key[idx];
                    if (k != 0 && (pass == 0 || (k >> (shift + 8)) == pre)) atomicAdd(&h[(unsigned)(k >> shift) & 255u], 1u);
                }
                for (int q = tid; q < 256; q += BT) hn[q] = 0;       // (the other histogram: nobody counts into it in this pass)
                __syncthreads();
                if (wid == 0) {
                    // lane l owns bins 4l .. 4l+3; suffix sums from the top bin down
                    const unsigned h0 = h[4 * lane], h1 = h[4 * lane + 1], h2 = h[4 * lane + 2], h3 = h[4 * lane + 3];
                    const unsigned mine = h0 + h1 + h2 + h3;
                    unsigned suf = mine;                         // inclusive suffix sum over lanes >= lane
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned v = __shfl_down(suf, o, 64);
                        if (lane + o < 64) suf += v;
                    }
                    const unsigned kk = (unsigned)s_kk;
                    const unsigned long long bal = __ballot(suf >= kk);
                    const int owner = 63 - __builtin_clzll(bal);   // highest lane whose suffix reaches kk
                    if (lane == owner) {
                        unsigned cum = suf - mine;               // keys in bins above this lane's
                        int d; unsigned hd;
                        if (cum + h3 >= kk) { d = 3; hd = h3; }
                        else { cum += h3; if (cum + h2 >= kk) { d = 2; hd = h2; }
                        else { cum += h2; if (cum + h1 >= kk) { d = 1; hd = h1; }
                        else { cum += h1; d = 0; hd = h0; } } }
                        s_kk = (int)(kk - cum);
                        s_prefix = (pre << 8) | (unsigned)(4 * lane + d);
                        s_done = hd == kk - cum;                 // the whole bin survives: no deeper pass needed
                    }
                }
                __syncthreads();
                if (s_done) break;
            }
            if (pass < 8) {
                const int shift = 56 - 8 * pass;
                const uint64_t thr = s_prefix;
                for (int idx = tid; idx < N; idx += BT) {
                    const uint64_t k = c_key[idx];
                    c_surv[idx] = k != 0 && (k >> shift) >= thr;
                }
            } else {
                const uint64_t thr = s_prefix;
                const int need = (int)s_kk;          // how many of the keys equal to thr survive
                // ties at the threshold: (character asc, candidate index asc)
                for (int idx = tid; idx < N; idx += BT) {
                    const uint64_t k = c_key[idx];
                    int sv = k > thr;
                    if (k == thr) {
                        const int myc = idx < NP ? idx % C : ech[idx - NP];
                        int rank = 0;
                        for (int j = 0; j < N; ++j) {
                            if (c_key[j] != thr || j == idx) continue;
                            const int oc = j < NP ? j % C : ech[j - NP];
                            rank += (oc < myc) || (oc == myc && j < idx);
                        }
                        sv = rank < need;
                    }
                    c_surv[idx] = sv;
                }
            }
        }
        __syncthreads();
        // ---- deterministic slot numbers: exclusive scan of the survivor flags in index order
        const int per = (N + BT - 1) / BT;
        const int i0 = min(tid * per, N), i1 = min(i0 + per, N);
        int local = 0;
        for (int idx = i0; idx < i1; ++idx) local += c_surv[idx];
        int incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wtot[wid] = incl;
        if (tid == 0) s_nnew = 0;
        __syncthreads();
        int wbase = 0, nnext = 0;
        for (int w = 0; w < NWAVE; ++w) { const int v = wtot[w]; if (w < wid) wbase += v; nnext += v; }
        const int nxt = cur ^ 1;
        // ---- 4a. commit survivors (new nodes first so that child counts are up before any removal)
        {
            int pos = wbase + incl - local;
            for (int idx = i0; idx < i1; ++idx) {
                if (!c_surv[idx]) continue;
                const int sl = pos++;
                const double lg = unokey(c_key[idx]);
                if (idx >= NP) {                                   // an entry that stays
                    const int i = idx - NP;
                    e_node[nxt * BW + sl] = node[i]; e_ch[nxt * BW + sl] = ech[i]; e_ds[nxt * BW + sl] = eds[i];
                    e_bprev[nxt * BW + sl] = e_bcur[i]; e_nbprev[nxt * BW + sl] = e_rep[i]; e_score[nxt * BW + sl] = lg;
                    nodes[node[i]].slot = sl;
                    continue;
                }
                const int i = idx / C, c = idx - i * C;
                const int pn = node[i];
                int id = c_child[idx];
                int nds;
                if (id >= 0) {                                     // dormant node (kept alive by descendants) comes back
                    atomicOr(&nodes[id].flags, F_EXISTS);
                    nds = nodes[id].dstate;
                } else {
                    const bool reuse = id <= -3;
                    id = reuse ? -3 - id : atomicAdd(nnodes, 1);
                    const int ds = eds[i];
                    const bool sp = a.has_lm && c == a.space;
                    nds = a.has_lm ? (sp ? 0 : a.trie_next[(size_t)ds * C + c]) : 0;
                    const int w = sp ? a.trie_word[ds] : -1;
                    Node r;
                    r.parent = pn; r.ch = c; r.tstep = t; r.dstate = nds; r.nchild = 0; r.flags = F_EXISTS; r.slot = sl; r.memo = MEMO_UNSET;
                    r.lpc = lp[c]; r.pad = 0;
                    const Node* pnode = nodes + pn;
                    for (int k = 0; k < MAXCTX; ++k) {
                        int v = pnode->ctx[k];
                        if (sp) v = k + 1 < NCTX ? pnode->ctx[k + 1] : (k + 1 == NCTX ? w : a.bos);
                        r.ctx[k] = v;
                    }
                    nodes[id] = r;
                    atomicAdd(&nodes[pn].nchild, 1);
                    if (!reuse) {
                        childtab[(size_t)pn * C + c] = id;
                        newn[atomicAdd(&s_nnew, 1)] = id;
                    }
                }
                nodes[id].slot = sl;
                e_node[nxt * BW + sl] = id; e_ch[nxt * BW + sl] = c; e_ds[nxt * BW + sl] = nds;
                e_bprev[nxt * BW + sl] = -INFINITY; e_nbprev[nxt * BW + sl] = lg; e_score[nxt * BW + sl] = lg;
            }
        }
        // ---- 4b. PathTrie::remove for the entries that fell out
        for (int i = tid; i < nb; i += BT) {
            if (c_surv[NP + i]) continue;
            Node* n = nodes + node[i];
            atomicAnd(&n->flags, ~F_EXISTS);
            n->slot = -1;
        }
        __syncthreads();
        {   // fresh nodes start with an empty child row
            const int nnew = s_nnew;
            for (int q = tid; q < nnew * C; q += BT) { const int j = q / C; childtab[(size_t)newn[j] * C + (q - j * C)] = -1; }
        }
        for (int i = tid; i < nb; i += BT) {
            if (c_surv[NP + i]) continue;
            int n = node[i];
            while (n > 0) {
                Node* r = nodes + n;
                if (atomicAdd(&r->nchild, 0) != 0) break;
                const int f = atomicAdd(&r->flags, 0);
                if (f & F_EXISTS) break;
                if (atomicOr(&r->flags, F_DELETED) & F_DELETED) break;    // someone else unlinked it
                const int p = r->parent;
                if (atomicSub(&nodes[p].nchild, 1) != 1) break;
                n = p;
            }
        }
        if (tid == 0) s_nb = nnext;
        __syncthreads();
        cur = nxt;
    }

    // ---- final: trailing partial word, order, write out
    {
        const int nb = s_nb;
        double* score = e_score + cur * BW; int* node = e_node + cur * BW; int* ech = e_ch + cur * BW;
        if (a.has_lm) {
            for (int i = tid; i < nb; i += BT) {
                const int n = node[i];
                if (n != 0 && ech[i] != a.space) {
                    const int w = a.trie_word[nodes[n].dstate];
                    double lm = kOovScore;
                    if (w >= 0) lm = (double)lm_cond_log10(a.lm, nodes[n].ctx, NCTX, w, a.unk) / (double)kLog10E;
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
            int len = 0;
            for (int n = node[i]; n > 0; n = nodes[n].parent) ++len;
            const size_t o = ((size_t)b * BW + rank) * a.T;
            int k = len;
            for (int n = node[i]; n > 0; n = nodes[n].parent) { --k; a.out_tok[o + k] = nodes[n].ch; a.out_step[o + k] = nodes[n].tstep; }
            a.out_len[(size_t)b * BW + rank] = len;
            a.out_score[(size_t)b * BW + rank] = score[i];
        }
        if (tid == 0) a.out_n[b] = nb;
    }
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

// Launches the beam search of a batch and the copies of its results into pinned host memory, all asynchronous on `stream`.
extern "C" int dsmi_beam_enqueue(dsmi_decoder* d, const float* probs, const int32_t* sizes, int B, int To, int beam, int cutoff_top_n,
                                 double cutoff_prob, void* stream) {
    if (!d) return DSMI_ERR_INVALID;
    const int C = (int)d->labels.size();
    if (!probs || B < 1 || To < 1 || beam < 1) { d->err = "bad beam arguments"; return DSMI_ERR_INVALID; }
    if (d->beam_pending) { d->err = "the previous beam search has not been collected"; return DSMI_ERR_INVALID; }
    const size_t NMAX = (size_t)beam * (C + 1);
    const size_t lds = sizeof(double) * (128 + 2 + 6 * (size_t)beam + 3 * (size_t)beam) + sizeof(uint64_t) * (NMAX + 2 + 2 * NWAVE) +
                       sizeof(int) * (6 * (size_t)beam + 2 * NMAX + (size_t)beam + 128) + sizeof(unsigned) * 512 +
                       sizeof(int) * (NWAVE + 1 + 8) + 64;
    if (lds > 160 * 1024 - 256) { d->err = "beam_width * (n_labels + 1) exceeds the on-chip candidate buffer"; return DSMI_ERR_CAPACITY; }
    DEC_HIP(d, hipSetDevice(d->device));
    hipStream_t s = (hipStream_t)stream;
    // ---- workspace carve
    const int ncap = 2 + To * beam;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_nodes = off; off += al((size_t)B * ncap * sizeof(Node));
    size_t o_child = off; off += al((size_t)B * ncap * C * 4);
    size_t o_nn = off; off += al((size_t)B * 4);
    size_t o_sizes = off; off += al((size_t)B * 4);
    // outputs, contiguous, in the order of the pinned image: tokens, steps, lens, counts, scores
    const size_t o_out = off;
    size_t o_tok = off; off += al((size_t)B * beam * To * 4);
    size_t o_step = off; off += al((size_t)B * beam * To * 4);
    size_t o_len = off; off += al((size_t)B * beam * 4);
    size_t o_n = off; off += al((size_t)B * 4);
    size_t o_score = off; off += al((size_t)B * beam * 8);
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
    a.nodes = (Node*)(w + o_nodes); a.childtab = (int32_t*)(w + o_child); a.nnodes = (int32_t*)(w + o_nn);
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
    DEC_HIP(d, hipFuncSetAttribute(reinterpret_cast<const void*>(beam_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(beam_kernel, dim3(B), dim3(BT), lds, s, a);
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
    DEC_HIP(d, hipMemcpy(d->pin + 2 * tok_bytes, dev + 2 * tok_bytes, d->pb_out_bytes - 2 * tok_bytes, hipMemcpyDeviceToHost));
    const unsigned char* q0 = d->pin;
    const int32_t* p_tok = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* p_step = reinterpret_cast<const int32_t*>(q0); q0 += tok_bytes;
    const int32_t* h_len = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * beam * 4);
    const int32_t* h_n = reinterpret_cast<const int32_t*>(q0); q0 += al((size_t)B * 4);
    const double* h_score = reinterpret_cast<const double*>(q0);
    // ... then, of the token arrays, only the columns that hold tokens: transcripts are a fraction of T_out long
    int maxlen = 0;
    for (size_t i = 0; i < (size_t)B * beam; ++i) maxlen = std::max(maxlen, (int)h_len[i]);
    if (maxlen > 0) {
        DEC_HIP(d, hipMemcpy2D(d->pin, (size_t)To * 4, dev, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost));
        DEC_HIP(d, hipMemcpy2D(d->pin + tok_bytes, (size_t)To * 4, dev + tok_bytes, (size_t)To * 4, (size_t)maxlen * 4, (size_t)B * beam, hipMemcpyDeviceToHost));
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
