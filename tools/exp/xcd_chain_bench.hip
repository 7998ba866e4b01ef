// Experiment (round 3, VERDICT "one chain per XCD"): what does a recurrent step cost when all workgroups of a chain sit on ONE
// XCD and hand the state over through that XCD's L2 (plain stores, sc1 loads), against today's placement-blind hand-off
// (sc1 write-through stores, sc1 loads, chains spread over all XCDs)?  Eight chains (2 batches x 2 directions x 2 tiles of cfgA)
// of 50 half-CU workgroups; per step a workgroup waits for the chain's counter, loads the chain's packed state (50 KB, the
// production layout and instruction: 1-KiB buffer loads), issues the production MFMA count, publishes its 1 KB slice with
// 2-byte stores, drains, signals.  Every loaded word is checked (a stale or torn hand-off shows as a mismatch).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/xcd_chain_bench.hip -o tools/exp/xcd_chain_bench && tools/exp/xcd_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int NCH = 8, NWG = 50, NKB = 25, SHARDS = 4, CNTW = SHARDS * 64;

struct Args {
    unsigned short* hpack;      // [2 parity][NCH][NKB][2 planes][512] fp16
    unsigned* cnt;              // [NCH][T][CNTW]
    unsigned* tick;             // [8] per-XCC tickets, [8..15] per-XCC arrivals seen, [16] global ticket, [17] errors, [18] mismatches
    float* sink;
    int T, mode;                // mode 0: chains spread (today); 1: chain = XCC, plain stores; 2: as 1, counters added at workgroup scope (L2)
    unsigned spin_limit;
};

__global__ __launch_bounds__(256, 2) void chain_kernel(Args p) {
    extern __shared__ float lds[];
    __shared__ int role[2];
    const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        int chain, idx;
        if (p.mode == 0) { const unsigned t = atomicAdd(&p.tick[16], 1u); chain = t / NWG; idx = t % NWG; }
        else { idx = (int)atomicAdd(&p.tick[xcc], 1u); chain = (int)xcc; if (idx >= NWG) chain = -1; }
        atomicAdd(&p.tick[8 + xcc], 1u);
        role[0] = chain; role[1] = idx;
    }
    __syncthreads();
    const int chain = role[0], w = role[1];
    if (chain < 0 || chain >= NCH) { if (tid == 0) atomicAdd(&p.tick[17], 1u); return; }
    const size_t par_bytes = (size_t)NCH * NKB * 2048;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.hpack, 0, (int)(2 * par_bytes), 0x00020000);
    unsigned* cnt = p.cnt + (size_t)chain * p.T * CNTW;
    const unsigned hchain = (unsigned)((size_t)chain * NKB * 2048);
    const int kb0 = (v * NKB) / 4, kb1 = ((v + 1) * NKB) / 4;
    const unsigned shard = (unsigned)(w & (SHARDS - 1)) * 64u;
    f16x8 wa;
    for (int e = 0; e < 8; ++e) wa[e] = (_Float16)(0.001f * (lane + e));
    unsigned bad = 0;
    float keep = 0.f;
    __shared__ int dead;
    if (tid == 0) dead = 0;
    __syncthreads();
    for (int s = 0; s < p.T; ++s) {
        f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (s > 0) {
            if (v == 0 && !dead) {      // wait for step s - 1 of the whole chain
                unsigned spins = 0;
                const unsigned* cp = &cnt[(size_t)(s - 1) * CNTW + (lane & (SHARDS - 1)) * 64];
                const unsigned need = (unsigned)((NWG + SHARDS - 1 - (lane & (SHARDS - 1))) / SHARDS);
                while (true) {
                    const unsigned got = lane < SHARDS ? __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need;
                    if (__builtin_amdgcn_ballot_w64(got < need) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > p.spin_limit) { dead = 1; atomicAdd(&p.tick[17], 1u); break; }
                }
            }
            __syncthreads();
            const unsigned hbase = (unsigned)(((s - 1) & 1) * par_bytes) + hchain + (unsigned)lane * 16u;
            f16x8 hv[7][2];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int kb = min(kb0 + i, kb1 - 1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    hv[i][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs, hbase + (unsigned)(kb * 2 + pl) * 1024u, 0, 16));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (kb0 + i < kb1) {
                    // every fp16 of the state must carry the previous step's tag
                    const _Float16 want = (_Float16)(float)((s - 1) & 1023);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int e = 0; e < 8; ++e) bad += hv[i][pl][e] != want;
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, hv[i][0], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, hv[i][1], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, hv[i][0], acc[g], 0, 0, 0);
                    }
                }
            }
        }
        for (int g = 0; g < 3; ++g) for (int r = 0; r < 4; ++r) lds[((v * 4 + g) * 16 + 4 * (lane >> 4) + r) * 20 + (lane & 15)] = acc[g][r];
        __syncthreads();
        float sum = 0.f;
        for (int q = 0; q < 4; ++q) sum += lds[((q * 4) * 16 + (tid & 15)) * 20 + ((tid >> 4) & 15)];
        keep += sum;
        // publish: this workgroup's 16 units x 16 clips, both planes: 2-byte stores, 128 contiguous bytes per wave instruction
        const _Float16 tag = (_Float16)(float)(s & 1023);
        const int cuh = tid >> 7, ce = tid & 7, cj = (tid >> 3) & 15;
        const unsigned off = (unsigned)((s & 1) * par_bytes) + hchain + (unsigned)(w >> 1) * 2048u + (unsigned)(2 * (w & 1) + cuh) * 256u + (unsigned)cj * 16u + (unsigned)ce * 2u;
        const unsigned short bits = __builtin_bit_cast(unsigned short, tag);
        if (p.mode == 0) {
            __builtin_amdgcn_raw_buffer_store_b16(bits, hrs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b16(bits, hrs, off + 1024u, 0, 16);
        } else {
            __builtin_amdgcn_raw_buffer_store_b16(bits, hrs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16(bits, hrs, off + 1024u, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (p.mode == 2) __hip_atomic_fetch_add(&cnt[(size_t)s * CNTW + shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&cnt[(size_t)s * CNTW + shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(&p.tick[18], bad);
    if (keep == 12345.f) p.sink[0] = keep;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 2000;
    Args a{};
    const size_t hbytes = 2ull * NCH * NKB * 2048, cbytes = sizeof(unsigned) * (size_t)NCH * T * CNTW;
    hipMalloc(&a.hpack, hbytes); hipMalloc(&a.cnt, cbytes); hipMalloc(&a.tick, 32 * 4); hipMalloc(&a.sink, 4);
    a.T = T; a.spin_limit = 2000000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"chains spread over the XCDs, sc1 stores (today's protocol)", "one chain per XCD, plain stores, agent-scope counters", "one chain per XCD, plain stores, counters added in the XCD's L2"};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            a.mode = mode;
            hipMemset(a.hpack, 0, hbytes); hipMemset(a.cnt, 0, cbytes); hipMemset(a.tick, 0, 32 * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain_kernel, dim3(NCH * NWG), dim3(256), 78 * 1024, 0, a);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned t[32]; hipMemcpy(t, a.tick, sizeof(t), hipMemcpyDeviceToHost);
            printf("rep %d mode %d: %.3f us/step  (%s)  arrivals per XCC %u %u %u %u %u %u %u %u  errors %u  stale words %u\n", rep, mode, ms * 1e3 / T, names[mode],
                   t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], t[17], t[18]);
        }
    return 0;
}
