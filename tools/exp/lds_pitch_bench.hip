// Experiment: ds_read_b128 bandwidth of the MFMA fragment read pattern (lane = (row li, half hk)) vs row pitch / swizzle.
#include <hip/hip_runtime.h>
#include <cstdio>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int PITCH, int SWZ>   // PITCH in bytes per row; SWZ: 0 none, 1 xor chunk with (row>>2)&3, 2 xor with (row>>1)&3
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, hk = lane >> 5;
    for (int i = tid; i < 20480; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        int opaque = 0;
        asm volatile("" : "+v"(opaque));          // keeps the loads inside the loop
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (wid & 1) * 64 + (r & 1) * 32 + li;       // 128 rows
            int chunk = ((it + (r >> 1)) & 1) * 2 + hk;               // 4 chunks of 16 B per 64-B row
            if (SWZ == 1) chunk ^= (row >> 2) & 3;
            if (SWZ == 2) chunk ^= (row >> 1) & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (r >> 2) * (128 * PITCH) + row * PITCH + chunk * 16 + opaque);
            acc += v;
        }
    }
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int PITCH, int SWZ> void run(unsigned* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000; const size_t lds = 81920;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<PITCH, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<PITCH, SWZ>), dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<PITCH, SWZ>), dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 512.0 * 4 * iters * 8 * 1024.0;
    printf("pitch %3d swz %d: %.3f ms  %.1f B/clk/CU (2.4 GHz)\n", PITCH, SWZ, ms, bytes / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 256 * 4);
    run<64, 0>(d); run<80, 0>(d); run<72, 0>(d); run<96, 0>(d); run<144, 0>(d); run<64, 1>(d); run<64, 2>(d);
    return 0;
}
