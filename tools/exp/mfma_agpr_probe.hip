// Does a v_mfma_f32_16x16x32_f16 whose A operand lies in AccVGPRs issue at the rate of one whose A operand lies in VGPRs?
// One workgroup of four waves (one per SIMD), NB k-blocks x 9 MFMAs per round (the ring kernel's item), B operands in registers
// (mode 0/1) or read from LDS two blocks ahead (mode 2/3); A in VGPRs (even modes) or AccVGPRs (odd modes).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_agpr_probe.hip -o mfma_agpr_probe && ./mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int NB = 8, NG = 3;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4* w, float* out, unsigned long long* cyc, int rounds) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NB * 2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * NB * 2048 / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    f16x8 wv[NB][NG][2];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                wv[i][g][pl] = __builtin_bit_cast(f16x8, w[((i * NG + g) * 2 + pl) * 64 + lane]);
                if (MODE & 1) asm volatile("" : "+a"(wv[i][g][pl])); else asm volatile("" : "+v"(wv[i][g][pl]));
            }
    __syncthreads();
    f16x8 b0 = __builtin_bit_cast(f16x8, w[lane]), b1 = __builtin_bit_cast(f16x8, w[64 + lane]);
    float r = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < rounds; ++s) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) { if (MODE & 1) asm volatile("" : "+a"(wv[i][g][pl])); else asm volatile("" : "+v"(wv[i][g][pl])); }
        f32x4 acc[NG], acl[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g] = f32x4{0, 0, 0, 0}; acl[g] = f32x4{0, 0, 0, 0}; }
        const unsigned char* sb = lds + lane * 16 + (s & 1) * NB * 2048;
        f16x8 bq[3][2];
        if (MODE & 2) {
            bq[0][0] = *reinterpret_cast<const f16x8*>(sb); bq[0][1] = *reinterpret_cast<const f16x8*>(sb + 1024);
            bq[1][0] = *reinterpret_cast<const f16x8*>(sb + 2048); bq[1][1] = *reinterpret_cast<const f16x8*>(sb + 3072);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if ((MODE & 2) && i + 2 < NB) {
                bq[(i + 2) % 3][0] = *reinterpret_cast<const f16x8*>(sb + (i + 2) * 2048);
                bq[(i + 2) % 3][1] = *reinterpret_cast<const f16x8*>(sb + (i + 2) * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 c0 = (MODE & 2) ? bq[i % 3][0] : b0, c1 = (MODE & 2) ? bq[i % 3][1] : b1;
#pragma unroll
            for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][1], c0, acl[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], c0, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], c1, acl[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) r += acc[g][0] + acl[g][1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    u32x4* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, NB * NG * 2 * 64 * 16); hipMemset(w, 0x3c, NB * NG * 2 * 64 * 16);
    hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    const int rounds = 20000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(256), 0, 0, w, out, cyc, rounds); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(256), 0, 0, w, out, cyc, rounds); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(256), 0, 0, w, out, cyc, rounds); break;
                default: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(256), 0, 0, w, out, cyc, rounds); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("A in %s, B from %s: %.2f shader cycles per MFMA (%d k-blocks x 9 per round)\n", (mode & 1) ? "AccVGPRs" : "VGPRs   ", (mode & 2) ? "LDS (two blocks ahead)" : "registers", (double)c / rounds / (NB * 9), NB);
    }
    return 0;
}
