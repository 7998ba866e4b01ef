// Experiment: split-fp16 inner loop with 4x2 register tiles and ONE persistent accumulator per tile: the cross
// products go into a temporary accumulator that is folded into the main one on the VALU (acc += tmp * 2^-11) every
// k-step.  12 fragment reads + 24 MFMAs + 128 v_fma per k-step.  Compare with mfma_lds_pipe's 2x2 / two accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int XS = 40, XPLANE = 256 * XS;

template <bool PIPE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);          // A: 2 planes x 256 rows ; W: 2 planes x 128 rows (in the same array)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, li = lane & 31, hk = lane >> 5;
    for (int i = tid; i < 3 * XPLANE; i += 256) As[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    f16x8 fa[2][4][2], fw[2][2][2];
    auto rd = [&](int buf, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fa[buf][t][pl] = *reinterpret_cast<const f16x8*>(As + pl * XPLANE + (wr * 128 + t * 32 + li) * XS + ks * 16 + hk * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fw[buf][t][pl] = *reinterpret_cast<const f16x8*>(As + 2 * XPLANE + pl * (XPLANE / 2) + (wc * 64 + t * 32 + li) * XS + ks * 16 + hk * 8);
    };
    rd(0, 0);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int cb = PIPE ? b : 0;
            if (PIPE) rd(b ^ 1, b ^ 1); else rd(0, b);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][mi][1], fw[cb][ni][0], zero, 0, 0, 0);
                    tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][mi][0], fw[cb][ni][1], tmp, 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][mi][0], fw[cb][ni][0], acc[mi][ni], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] += tmp[r] * (1.f / 2048.f);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <bool PIPE> void run(float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 10000; const size_t lds = 3 * XPLANE * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<PIPE>, dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<PIPE>, dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("4x2 tiles, folded cross terms, %s: %.3f ms  MFMA %.1f TFLOP/s (fp16 executed)  %s\n", PIPE ? "pipelined reads" : "plain reads", ms,
           512.0 * 4 * iters * 24 * 32768.0 / ms / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() { float* d; hipMalloc(&d, 512 * 256 * 4); run<false>(d); run<true>(d); return 0; }
