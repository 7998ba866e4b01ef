// Cycles per v_mfma_f64_16x16x4_f64 on one SIMD (gfx950): one wave per SIMD (a 256-thread workgroup on one CU), 1, 2 or 4 independent
// accumulator chains, operands in registers; shader clock (s_memtime) around 4096 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void probe(double* out, unsigned long long* cyc, double a0, double b0) {
    f64x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 4096 / CHAINS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * sizeof(double)); hipMalloc(&cyc, 8);
    unsigned long long h;
#define RUN(C) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<C>, dim3(1), dim3(256), 0, 0, out, cyc, 1.0, 2.0); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%d chain(s): %.1f shader cycles per MFMA (one wave per SIMD)\n", C, (double)h / 4096.0);
    RUN(1) RUN(2) RUN(4) RUN(8)
    return 0;
}
