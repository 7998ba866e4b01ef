// Host-side n-gram language model + vocabulary dictionary for the CTC beam search.
//
// Stands in for ctcdecode's Scorer (KenLM model + OpenFST dictionary), which the reference
// reaches through BeamCTCDecoder (danspeech/deepspeech/decoder.py:95-100) and which is not
// part of the reference tree.  Built from an ARPA text file or from a KenLM binary (.klm: what the reference's
// language_models.* factories hand to the decoder, danspeech/language_models/dsl_3gram.py:16-20).  The same tables are uploaded
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

// ---- a KenLM "probing" binary's search memory, used as it lies in the file (lm_klm.cpp.inc reads the header).
// Unigrams: {f32 prob, f32 backoff} by word id; order n >= 2: open-addressing table of {u64 key, f32 prob[, f32 backoff]}
// (16 bytes, 12 for the highest order), slot = key % buckets, linear probing, key 0 = empty.  Offsets are multiples
// of 4 only (the vocabulary table in front has 12-byte entries), so everything is read as 32-bit words.
struct KlmView {
    const unsigned char* base = nullptr;      // start of the search memory; nullptr = not a KenLM probing model
    int order = 0;
    uint64_t uni_off = 0;
    uint64_t off[kMaxOrder] = {0, 0, 0, 0, 0, 0};        // table of order n at [n - 2]
    uint64_t buckets[kMaxOrder] = {0, 0, 0, 0, 0, 0};
};

__host__ __device__ inline uint32_t klm_u32(const unsigned char* p) { return *reinterpret_cast<const uint32_t*>(p); }
__host__ __device__ inline uint64_t klm_u64(const unsigned char* p) { return (uint64_t)klm_u32(p) | ((uint64_t)klm_u32(p + 4) << 32); }
// a stored prob's sign bit is a flag of KenLM's (GenericProbingProxy::Prob ORs it back in): log10 p = -|stored|
__host__ __device__ inline float klm_prob(const unsigned char* p) {
    union { uint32_t u; float f; } v;
    v.u = klm_u32(p) | 0x80000000u;
    return v.f;
}
__host__ __device__ inline float klm_f32(const unsigned char* p) {
    union { uint32_t u; float f; } v;
    v.u = klm_u32(p);
    return v.f;
}

// KenLM's n-gram key: start from the LAST word, fold the earlier ones in from right to left (lm/search_hashed.hh,
// detail::CombineWordHash).
__host__ __device__ inline uint64_t klm_key(const int32_t* ids, int n) {
    uint64_t h = (uint64_t)(uint32_t)ids[n - 1];
    for (int i = n - 2; i >= 0; --i)
        h = (h * 8978948897894561157ull) ^ ((uint64_t)(1u + (uint32_t)ids[i]) * 17894857484156487943ull);
    return h;
}

__host__ __device__ inline bool klm_lookup(const KlmView& k, const int32_t* ids, int n, float* lp, float* bo) {
    if (n == 1) {
        const unsigned char* p = k.base + k.uni_off + 8ull * (uint32_t)ids[0];
        *lp = klm_prob(p); *bo = klm_f32(p + 4);
        return true;
    }
    if (n > k.order) return false;
    const uint64_t key = klm_key(ids, n), nb = k.buckets[n - 2], es = n == k.order ? 12 : 16;
    const unsigned char* tab = k.base + k.off[n - 2];
    for (uint64_t s = key % nb;; s = s + 1 == nb ? 0 : s + 1) {
        const unsigned char* p = tab + s * es;
        const uint64_t e = klm_u64(p);
        if (e == key) { *lp = klm_prob(p + 8); *bo = es == 16 ? klm_f32(p + 12) : 0.f; return true; }
        if (e == 0) return false;
    }
}

// What a scorer looks n-grams up in: the own table (ARPA text, KenLM trie binaries) or a KenLM probing image.
struct LmView {
    const LmEntry* tab = nullptr; uint64_t mask = 0;
    KlmView klm;
};

__host__ __device__ inline bool lm_lookup(const LmView& v, const int32_t* ids, int n, float* lp, float* bo) {
    if (v.klm.base) return klm_lookup(v.klm, ids, n, lp, bo);
    const uint64_t k = ngram_hash(ids, n);
    for (uint64_t s = k & v.mask;; s = (s + 1) & v.mask) {
        const uint64_t e = v.tab[s].key;
        if (e == k) { *lp = v.tab[s].lp; *bo = v.tab[s].bo; return true; }
        if (e == 0) return false;
    }
}

// log10 p(w | ctx[0..n-1]) by back-off (float accumulation like KenLM's), all ids in vocabulary.
__host__ __device__ inline float lm_cond_log10(const LmView& v, const int32_t* ctx, int n, int32_t w, int32_t unk) {
    int32_t g[kMaxOrder];
    float acc = 0.f;
    for (int start = 0; start <= n; ++start) {
        const int len = n - start;
        for (int i = 0; i < len; ++i) g[i] = ctx[start + i];
        g[len] = w;
        float lp, bo;
        if (lm_lookup(v, g, len + 1, &lp, &bo)) return acc + lp;
        if (len > 0 && lm_lookup(v, g, len, &lp, &bo)) acc += bo;
    }
    float lp = 0.f, bo;
    int32_t u = unk;
    if (lm_lookup(v, &u, 1, &lp, &bo)) return acc + lp;
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

    // KenLM binaries: kind 1 = probing (klm_blob holds the file's search memory, looked up in place), 2 = trie (converted
    // into `table` at load time); 0 = ARPA text
    int kind = 0;
    std::vector<unsigned char> klm_blob;
    KlmView klm;                     // offsets of klm_blob; base points into it (host view)
    LmView view() const {
        LmView v;
        v.tab = table.data(); v.mask = mask; v.klm = klm;
        v.klm.base = kind == 1 ? klm_blob.data() : nullptr;       // (not a stored pointer: the object may have been moved)
        return v;
    }

    // returns "" on success, else an error message
    std::string load(const std::string& path, const std::vector<std::string>& labels);    // ARPA text or KenLM binary, by its first bytes
    std::string load_arpa(const std::string& path, const std::vector<std::string>& labels);
    std::string load_klm(const std::string& path, const std::vector<std::string>& labels);
    void table_init(size_t n_grams);
    void table_put(const int32_t* ids, int n, float lp, float bo);
    void build_dictionary(const std::vector<std::string>& labels);
    double cond_ln(const std::vector<int32_t>& words) const;       // Scorer::get_log_cond_prob
    double sent_ln(const std::vector<int32_t>& words) const;       // Scorer::get_sent_log_prob
};

}  // namespace dsmi
