// Experiment: what bounds the split-fp16 GEMM inner loop -- the MFMA pipe or the LDS reads?
// Same shape as gemm_f16x3_kernel's k-step: 4 waves, 2x2 tiles per wave, 2 accumulators per tile,
// 12 MFMAs and 8 ds_read_b128 per k-step; variants drop the LDS reads or the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int XS = 40, XPLANE = 128 * XS;

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
// MODE 0: LDS + MFMA, 1: MFMA only, 2: LDS only; 3: 0 + two barriers per k-tile (2 k-steps); 4: 3 + 8 ds_write_b128 per
// k-tile; 5: 4 + 8 global b128 loads per k-tile (the full staging pattern of gemm_f16x3_kernel)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const u32x4* g = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    _Float16* Ws = As + 2 * XPLANE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, li = lane & 31, hk = lane >> 5;
    for (int i = tid; i < 4 * XPLANE; i += 256) As[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    f32x16 acc[2][2], acl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0; acl[i][j][r] = 0; }
    f16x8 af[2][2], wf[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 8; ++e) { af[a][b][e] = (_Float16)(0.01f * lane); wf[a][b][e] = (_Float16)(0.02f * e); }
    u32x4 stg[8];
    for (int i = 0; i < 8; ++i) stg[i] = u32x4{(unsigned)tid, 1u, 2u, 3u};
    for (int it = 0; it < iters; ++it) {
        const int ks = it & 1;
        if (MODE >= 3 && ks == 0) {
            if (MODE >= 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = tid + 256 * (i & 3);
                    *reinterpret_cast<u32x4*>((i < 4 ? As : Ws) + (c >> 9) * XPLANE + ((c >> 2) & 127) * XS + (c & 3) * 8) = stg[i];
                }
            }
            __syncthreads();
            if (MODE >= 5) {
                const u32x4* src = g + ((size_t)(blockIdx.x * 31 + (it >> 1)) % 4096) * 2048;
#pragma unroll
                for (int i = 0; i < 8; ++i) stg[i] = src[tid + 256 * i];
            }
        }
        if (MODE != 1) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    af[mi][pl] = *reinterpret_cast<const f16x8*>(As + pl * XPLANE + (wr * 64 + mi * 32 + li) * XS + ks * 16 + hk * 8);
                    wf[mi][pl] = *reinterpret_cast<const f16x8*>(Ws + pl * XPLANE + (wc * 64 + mi * 32 + li) * XS + ks * 16 + hk * 8);
                }
        }
        if (MODE != 2) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][1], wf[ni][0], acl[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][0], wf[ni][0], acc[mi][ni], 0, 0, 0);
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][0], wf[ni][1], acl[mi][ni], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) for (int pl = 0; pl < 2; ++pl) for (int e = 0; e < 8; ++e) { acc[mi][pl][e] += (float)af[mi][pl][e]; acl[mi][pl][e] += (float)wf[mi][pl][e]; }
        }
        if (MODE >= 3 && ks == 1) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r] + acl[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> void run(const char* name, float* d, int iters, const u32x4* g = nullptr) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = 4 * XPLANE * 2;
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), lds, 0, d, iters, g);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), lds, 0, d, iters, g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mfma = 512.0 * 4 * iters * 12, flops = mfma * 32768.0, ldsb = 512.0 * 4 * iters * 8 * 1024.0;
    printf("%-10s %.3f ms  MFMA %.1f TFLOP/s (fp16 executed)   LDS %.1f TB/s (%.1f B/clk/CU at 2.4 GHz)\n", name, ms,
           MODE != 2 ? flops / ms / 1e9 : 0.0, MODE != 1 ? ldsb / ms / 1e9 : 0.0, MODE != 1 ? ldsb / (ms * 1e-3) / 256 / 2.4e9 : 0.0);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    u32x4* g; hipMalloc(&g, (size_t)4096 * 2048 * 16 + 65536); hipMemset(g, 1, (size_t)4096 * 2048 * 16);
    run<0>("lds+mfma", d, 20000); run<1>("mfma", d, 20000);
    run<3>("+barriers", d, 20000); run<4>("+ds_write", d, 20000); run<5>("+global", d, 20000, g);
    return 0;
}
