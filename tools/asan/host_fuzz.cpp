// CPU harness for the host-only code of libdsmi.so, built with -fsanitize=address,undefined (`make -C danspeech_amd/csrc asan`).
// The .klm and ARPA readers parse untrusted files; the GPU pool has no device sanitizer, so this is where they are held to
// "a damaged file is refused or read, never a stray access".  tests/test_asan_host.py feeds it a few hundred mutated files.
//   host_fuzz lm FILE...      load each file (ARPA text or KenLM binary); when it loads, look n-grams up and score sequences
//   host_fuzz plan SEED N     random shard plans and phrase gates checked against their definitions
#define __host__
#define __device__
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "lm.h"
#include "lm.cpp.inc"
#include "lm_klm.cpp.inc"
#include "host_logic.h"

using namespace dsmi;

static int run_lm(int argc, char** argv) {
    std::vector<std::string> labels;
    const char* lab = "_abcdefghijklmnopqrstuvwxyz\xc3\xa6\xc3\xb8\xc3\xa5\xc3\xa9\xc3\xbc ";
    for (const char* p = lab; *p;) { const unsigned char c = (unsigned char)*p; const int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : 3; labels.emplace_back(p, n); p += n; }
    int loaded = 0, refused = 0;
    for (int i = 2; i < argc; ++i) {
        HostLM lm;
        const std::string msg = lm.load(argv[i], labels);
        if (!msg.empty()) { ++refused; continue; }
        ++loaded;
        if (lm.order < 1 || lm.order > kMaxOrder || lm.vocab.empty()) { std::fprintf(stderr, "%s: loaded with an implausible shape\n", argv[i]); return 3; }
        std::mt19937 rng(12345);
        const LmView v = lm.view();
        double sink = 0;
        for (int rep = 0; rep < 300; ++rep) {
            int32_t ids[kMaxOrder];
            const int n = 1 + (int)(rng() % (unsigned)lm.order);
            for (int k = 0; k < n; ++k) ids[k] = (int32_t)(rng() % lm.vocab.size());
            float lp = 0, bo = 0;
            sink += lm_lookup(v, ids, n, &lp, &bo) ? lp + bo : 0.0;
            sink += lm_cond_log10(v, ids, n - 1, ids[n - 1], lm.unk);
        }
        std::vector<int32_t> words;
        for (int k = 0; k < 7; ++k) words.push_back((int32_t)(rng() % lm.vocab.size()));
        sink += lm.sent_ln(words);
        // the dictionary trie the beam search walks
        for (size_t s = 0; s < lm.trie_word.size(); ++s) if (lm.trie_word[s] >= (int32_t)lm.vocab.size()) { std::fprintf(stderr, "dictionary word id out of range\n"); return 3; }
        for (int32_t nx : lm.trie_next) if (nx >= (int32_t)lm.trie_word.size()) { std::fprintf(stderr, "dictionary arc out of range\n"); return 3; }
        if (std::isnan(sink)) { std::fprintf(stderr, "%s: NaN score\n", argv[i]); return 3; }
    }
    std::printf("loaded %d refused %d\n", loaded, refused);
    return 0;
}

static int run_plan(int argc, char** argv) {
    std::mt19937 rng(argc > 2 ? (unsigned)std::atoi(argv[2]) : 1u);
    const int reps = argc > 3 ? std::atoi(argv[3]) : 200;
    for (int rep = 0; rep < reps; ++rep) {
        const int n = (int)(rng() % 70), world = 1 + (int)(rng() % 9);
        std::vector<int64_t> len((size_t)n);
        for (auto& l : len) l = (int64_t)(rng() % 5) * 1000 + (rng() % 3);
        std::vector<int32_t> rank_of((size_t)n, -1), slot_of((size_t)n, -1);
        plan_shards(len.data(), n, world, rank_of.data(), slot_of.data());
        // every rank's clips in slot order are non-increasing in length, ranks differ by at most one clip, every clip placed once
        std::vector<std::vector<int64_t>> per((size_t)world);
        for (int r = 0; r < world; ++r) per[(size_t)r].assign((size_t)((n - r + world - 1) / world > 0 ? (n - r + world - 1) / world : 0), -1);
        for (int i = 0; i < n; ++i) {
            if (rank_of[(size_t)i] < 0 || rank_of[(size_t)i] >= world) return 4;
            auto& row = per[(size_t)rank_of[(size_t)i]];
            if (slot_of[(size_t)i] < 0 || (size_t)slot_of[(size_t)i] >= row.size() || row[(size_t)slot_of[(size_t)i]] != -1) return 4;
            row[(size_t)slot_of[(size_t)i]] = len[(size_t)i];
        }
        for (auto& row : per) for (size_t k = 1; k < row.size(); ++k) if (row[k] > row[k - 1]) return 4;
        // the phrase gate: random energies, every phrase inside the signal, in order, never more stored than asked for
        const int64_t nhops = (int64_t)(rng() % 400);
        const int step = 128 << (rng() % 7);
        std::vector<double> e((size_t)nhops);
        for (auto& x : e) x = (rng() % 3) ? 0.0 : 1000.0;
        const int cap = (int)(rng() % 6);
        std::vector<int64_t> s0((size_t)cap + 1, -7), s1((size_t)cap + 1, -7);
        const int found = segment_phrases(e.data(), nhops, step, 600.0, (int)(rng() % 9), (int)(rng() % 4), s0.data(), s1.data(), cap);
        if (found < 0 || s0[(size_t)cap] != -7 || s1[(size_t)cap] != -7) return 5;
        for (int k = 0; k < std::min(found, cap); ++k)
            if (s0[(size_t)k] < 0 || s1[(size_t)k] <= s0[(size_t)k] || s1[(size_t)k] > nhops * step || (k && s0[(size_t)k] < s1[(size_t)k - 1] - 2 * (int64_t)step)) return 5;
    }
    std::printf("plans and gates ok\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "lm")) return run_lm(argc, argv);
    if (argc >= 2 && !std::strcmp(argv[1], "plan")) return run_plan(argc, argv);
    std::fprintf(stderr, "usage: host_fuzz lm FILE... | host_fuzz plan [SEED [N]]\n");
    return 2;
}
