// Host-side n-gram language model + vocabulary dictionary for the CTC beam search.
//
// Stands in for ctcdecode's Scorer (KenLM model + OpenFST dictionary), which the reference
// reaches through BeamCTCDecoder (danspeech/deepspeech/decoder.py:95-100) and which is not
// part of the reference tree.  Built from an ARPA text file.  The same tables are uploaded
// to the GPU for the in-kernel scorer (beam.hip) and kept on the host for the final
// sentence-level rescoring (ctcdecode "approx_ctc").
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace dsmi {

constexpr int kMaxOrder = 6;
constexpr double kOovScore = -1000.0;             // ctcdecode scorer.h OOV_SCORE
constexpr float kLog10E = 0.4342944819f;          // ctcdecode decoder_utils.h NUM_FLT_LOGE (a C float)

struct LmEntry { uint64_t key; float lp; float bo; };   // key 0 = empty slot

// FNV-1a over the word ids of an n-gram, shared by host and device.
__host__ __device__ inline uint64_t ngram_hash(const int32_t* ids, int n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int i = 0; i < n; ++i) {
        h ^= (uint64_t)(uint32_t)(ids[i] + 1);
        h *= 1099511628211ull;
    }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h ? h : 1;
}

__host__ __device__ inline bool lm_lookup(const LmEntry* tab, uint64_t mask, const int32_t* ids, int n, float* lp, float* bo) {
    const uint64_t k = ngram_hash(ids, n);
    for (uint64_t s = k & mask;; s = (s + 1) & mask) {
        const uint64_t e = tab[s].key;
        if (e == k) { *lp = tab[s].lp; *bo = tab[s].bo; return true; }
        if (e == 0) return false;
    }
}

// log10 p(w | ctx[0..n-1]) by back-off (float accumulation like KenLM's), all ids in vocabulary.
__host__ __device__ inline float lm_cond_log10(const LmEntry* tab, uint64_t mask, const int32_t* ctx, int n, int32_t w, int32_t unk) {
    int32_t g[kMaxOrder];
    float acc = 0.f;
    for (int start = 0; start <= n; ++start) {
        const int len = n - start;
        for (int i = 0; i < len; ++i) g[i] = ctx[start + i];
        g[len] = w;
        float lp, bo;
        if (lm_lookup(tab, mask, g, len + 1, &lp, &bo)) return acc + lp;
        if (len > 0 && lm_lookup(tab, mask, g, len, &lp, &bo)) acc += bo;
    }
    float lp = 0.f, bo;
    int32_t u = unk;
    if (lm_lookup(tab, mask, &u, 1, &lp, &bo)) return acc + lp;
    return acc;
}

struct HostLM {
    int order = 0;
    std::vector<std::string> vocab;
    std::unordered_map<std::string, int32_t> word2id;
    int32_t unk = -1, bos = -1, eos = -1;
    std::vector<LmEntry> table;      // open addressing, size mask+1
    uint64_t mask = 0;
    // dictionary trie over label ids: next[node * C + label] (-1 = no arc), word[node] (-1 = no word ends here)
    int n_labels = 0;
    std::vector<int32_t> trie_next, trie_word;

    // returns "" on success, else an error message
    std::string load_arpa(const std::string& path, const std::vector<std::string>& labels);
    double cond_ln(const std::vector<int32_t>& words) const;       // Scorer::get_log_cond_prob
    double sent_ln(const std::vector<int32_t>& words) const;       // Scorer::get_sent_log_prob
};

}  // namespace dsmi
