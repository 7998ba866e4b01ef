// Host-only logic of libdsmi.so that touches no HIP call: kept in a header of its own so that `make -C danspeech_amd/csrc asan`
// can build it (with the language-model readers lm.cpp.inc / lm_klm.cpp.inc) for the CPU under AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/asan/host_fuzz.cpp, tests/test_asan_host.py): GPU sanitizers are not available on the pool.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace dsmi {

// order[k] = index of the k-th longest clip (stable): rank k % world takes it as its (k / world)-th clip
// (danspeech_amd/parallel.py plan_shards; pack_padded_sequence's order, reference model.py:117)
inline std::vector<int> length_order(const int64_t* n_samples, int n) {
    std::vector<int> order((size_t)std::max(n, 0));
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_samples[a] > n_samples[b]; });
    return order;
}

inline void plan_shards(const int64_t* n_samples, int n, int world, int32_t* rank_of, int32_t* slot_of) {
    const std::vector<int> order = length_order(n_samples, n);
    for (int k = 0; k < n; ++k) { rank_of[order[(size_t)k]] = k % world; slot_of[order[(size_t)k]] = k / world; }
}

// The energy gate of the reference's example_scripts/video_transcribe_simulation.py:84-143, one pass over the hop energies:
// phrases [seg_start, seg_end) in samples.  Returns how many phrases there are; at most max_segments are stored.
inline int segment_phrases(const double* e, int64_t nhops, int step, double energy_threshold, int pause_hops, int phrase_hops,
                           int64_t* seg_start, int64_t* seg_end, int max_segments) {
    bool is_speaking = false;
    int64_t frames_counter = 0, pause_count = 0, start_index = 0, iterator = 0;
    int found = 0;
    for (int64_t i = 0; i < nhops; ++i) {
        const double energy = e[i];
        if (energy > energy_threshold && !is_speaking) {
            is_speaking = true;
            start_index = iterator - 2 * (int64_t)step;
            if (start_index < 0) start_index = iterator;
        }
        iterator += step;
        if (is_speaking) {
            ++frames_counter;
            if (energy > energy_threshold) pause_count = 0;
            else ++pause_count;
        }
        if (pause_count > pause_hops && is_speaking) {
            if (frames_counter - pause_count > phrase_hops) {
                if (found < max_segments) { seg_start[found] = start_index; seg_end[found] = iterator; }
                ++found;
            }
            is_speaking = false;
            frames_counter = 0;
            pause_count = 0;
        }
    }
    return found;
}

}  // namespace dsmi
