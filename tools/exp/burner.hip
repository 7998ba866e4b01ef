// A neighbour for timing experiments: `wgs` workgroups that multiply in registers (mode 0: MFMA burner, no memory traffic) or
// stream a buffer from HBM (mode 1) for `iters` rounds, on the caller's stream.  tools/exp/ring_with_neighbour.py
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/exp/burner.hip -o tools/exp/libburner.so
#include <hip/hip_runtime.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__global__ __launch_bounds__(256) void burn_kernel(int mode, int iters, const u32x4* buf, size_t n16, float* sink) {
    float keep = 0.f;
    if (mode == 0) {
        f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < iters; ++r)
#pragma unroll
            for (int k = 0; k < 64; ++k) c[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k & 7], 0, 0, 0);
        for (int i = 0; i < 8; ++i) keep += c[i][0];
    } else {
        u32x4 acc = {0, 0, 0, 0};
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (int r = 0; r < iters; ++r)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) acc ^= __builtin_nontemporal_load(buf + i);
        keep = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
    }
    if (keep == 123.456f) sink[threadIdx.x] = keep;
}
extern "C" void burn(int mode, int wgs, int iters, void* buf, size_t bytes, float* sink, hipStream_t s) {
    hipLaunchKernelGGL(burn_kernel, dim3(wgs), dim3(256), 0, s, mode, iters, (const u32x4*)buf, bytes / 16, sink);
}
