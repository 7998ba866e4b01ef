// A small SIMT emulation for debugging barrier-structured HIP kernels on the CPU: one host thread per GPU thread, real
// barriers for __syncthreads(), wave-wide collectives through a per-wave rendezvous.  Correct for kernels that only assume
// what HIP promises plus wave-lockstep around their collectives (every lane of a wave reaches the same collective).  Slow
// (thousands of host threads), meant for tiny inputs.  Not part of the product: tools/emu/beam_emu.cpp is its one user.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__
#define __launch_bounds__(...)

struct dim3 { unsigned x = 1, y = 1, z = 1; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline thread_local dim3 threadIdx, blockIdx, blockDim;

namespace simt {
struct Block {
    int nthreads;
    std::unique_ptr<std::barrier<>> all;
    std::vector<std::unique_ptr<std::barrier<>>> wave;
    std::vector<uint64_t> scratch;       // [nthreads]
    explicit Block(int n) : nthreads(n), scratch(n) {
        all.reset(new std::barrier<>(n));
        for (int w = 0; w < (n + 63) / 64; ++w) wave.emplace_back(new std::barrier<>(std::min(64, n - 64 * w)));
    }
};
inline Block* g_block = nullptr;
inline void wave_sync() { g_block->wave[threadIdx.x >> 6]->arrive_and_wait(); }
template <typename T> inline uint64_t bits(T v) { uint64_t u = 0; std::memcpy(&u, &v, sizeof(T)); return u; }
template <typename T> inline T unbits(uint64_t u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }
template <typename T> inline T exchange(T v, int src_lane) {       // src_lane < 0: keep own
    const int base = threadIdx.x & ~63u;
    g_block->scratch[threadIdx.x] = bits(v);
    wave_sync();
    const T r = src_lane < 0 || src_lane > 63 ? v : unbits<T>(g_block->scratch[base + src_lane]);
    wave_sync();
    return r;
}
template <typename K> void launch(int grid, int block, K kernel) {
    for (int b = 0; b < grid; ++b) {
        Block blk(block);
        g_block = &blk;
        std::vector<std::thread> th;
        for (int t = 0; t < block; ++t)
            th.emplace_back([=]() { threadIdx = dim3(t); blockIdx = dim3(b); blockDim = dim3(block); kernel(); });
        for (auto& x : th) x.join();
    }
}
}  // namespace simt

inline void __syncthreads() { simt::g_block->all->arrive_and_wait(); }
inline void __builtin_amdgcn_wave_barrier() { simt::wave_sync(); }
template <typename T> inline T __shfl_xor(T v, int mask, int = 64) { return simt::exchange(v, (int)((threadIdx.x & 63) ^ mask)); }
template <typename T> inline T __shfl(T v, int src, int = 64) { return simt::exchange(v, src); }
inline void __builtin_amdgcn_s_setprio(int) {}
template <typename T> inline T __shfl_up(T v, int d, int = 64) { const int l = threadIdx.x & 63; return simt::exchange(v, l >= d ? l - d : -1); }
template <typename T> inline T __shfl_down(T v, int d, int = 64) { const int l = threadIdx.x & 63; return simt::exchange(v, l + d < 64 ? l + d : -1); }
inline unsigned long long __ballot(int pred) {
    const int base = threadIdx.x & ~63u;
    simt::g_block->scratch[threadIdx.x] = pred ? 1 : 0;
    simt::wave_sync();
    unsigned long long m = 0;
    const int n = std::min(64, simt::g_block->nthreads - base);
    for (int l = 0; l < n; ++l) m |= (unsigned long long)(simt::g_block->scratch[base + l] & 1) << l;
    simt::wave_sync();
    return m;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
inline long long __double_as_longlong(double v) { return simt::unbits<long long>(simt::bits(v)); }
inline double __longlong_as_double(long long v) { return simt::unbits<double>(simt::bits(v)); }
inline int __float_as_int(float v) { return simt::unbits<int>(simt::bits(v)); }
inline float __int_as_float(int v) { return simt::unbits<float>(simt::bits(v)); }
inline unsigned long long wall_clock64() { return 0; }
using std::min;
using std::max;
