// Experiment: inner loop of the split-fp16 GEMM with direct global->LDS loads (global_load_lds_dwordx4),
// 3 LDS stages, 8 waves (256 x 128 tile), XOR-swizzled unpadded rows.  Compare with mfma_lds_bench's "+global".
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int STAGE = 48 * 1024;     // A: 2 planes x 256 rows x 64 B ; W: 2 planes x 128 rows x 64 B
constexpr int NST = 3;

__global__ __launch_bounds__(512, 1) void k(float* out, int ktiles, const unsigned char* g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, li = lane & 31, hk = lane >> 5;
    f32x16 acc[2][2], acl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0; acl[i][j][r] = 0; }
    // issue the DMA of k-tile kt into stage kt % NST: 48 wave-instructions of 1 KiB, 6 per wave
    auto dma = [&](int kt) {
        unsigned char* st = smem + (kt % NST) * STAGE;
        const unsigned char* src = g + ((size_t)(blockIdx.x * 31 + kt) % 2048) * STAGE;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int j = wid * 6 + i;                       // 1-KiB chunk: 16 rows of 64 B
            const int r = (lane >> 2), pc = lane & 3;
            const int c = pc ^ ((r >> 2) & 3);               // logical 16-byte chunk this lane fetches
            __builtin_amdgcn_global_load_lds(GLB_PTR(src + j * 1024 + r * 64 + c * 16), LDS_PTR(st + j * 1024), 16, 0, 0);
        }
    };
    dma(0); dma(1);
    for (int kt = 0; kt < ktiles; ++kt) {
        if (kt + 2 < ktiles) { dma(kt + 2); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (kt + 1 < ktiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // tile kt landed for every wave
        const unsigned char* st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[2][2], wf[2][2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int ra = wr * 64 + mi * 32 + li, rw = wc * 64 + mi * 32 + li;
                    af[mi][pl] = *reinterpret_cast<const f16x8*>(st + pl * 16384 + ra * 64 + (((ks * 2 + hk) ^ ((ra >> 2) & 3)) * 16));
                    wf[mi][pl] = *reinterpret_cast<const f16x8*>(st + 32768 + pl * 8192 + rw * 64 + (((ks * 2 + hk) ^ ((rw >> 2) & 3)) * 16));
                }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][1], wf[ni][0], acl[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][0], wf[ni][0], acc[mi][ni], 0, 0, 0);
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][0], wf[ni][1], acl[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();                                     // everyone is done with stage kt before it is refilled
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r] + acl[i][j][r];
    out[blockIdx.x * 512 + tid] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    unsigned char* g; hipMalloc(&g, (size_t)2048 * STAGE + 65536); hipMemset(g, 0, (size_t)2048 * STAGE);
    const int ktiles = 10000; const size_t lds = NST * STAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, d, ktiles, g);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, d, ktiles, g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = 256.0 * 8 * ktiles * 24 * 32768.0;
    printf("dma 3-stage 8 waves: %.3f ms  MFMA %.1f TFLOP/s (fp16 executed)  err=%s\n", ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
