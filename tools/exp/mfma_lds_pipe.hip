// Experiment: does software-pipelining the LDS fragment reads (next k-step's reads issued before the current k-step's
// MFMAs) lift the LDS+MFMA inner loop of the split-fp16 GEMM above the 62 % of mfma_lds_bench's unpipelined loop?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int XS = 40, XPLANE = 128 * XS;

template <int NREAD>   // fragment reads per k-step: 8 = both operands from LDS, 4 = one operand only
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, li = lane & 31, hk = lane >> 5;
    for (int i = tid; i < 4 * XPLANE; i += 256) As[i] = (_Float16)(0.001f * (i % 97));
    __syncthreads();
    f32x16 acc[2][2], acl[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0; acl[i][j][r] = 0; }
    f16x8 fr[2][4][2];
    auto rd = [&](int buf, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (t * 2 + pl < NREAD) {
                    const int row = (t < 2 ? wr : wc) * 64 + (t & 1) * 32 + li;
                    fr[buf][t][pl] = *reinterpret_cast<const f16x8*>(As + (t < 2 ? 0 : 2 * XPLANE) + pl * XPLANE + row * XS + ks * 16 + hk * 8);
                }
            }
    };
    for (int t = 0; t < 4; ++t) for (int pl = 0; pl < 2; ++pl) for (int e = 0; e < 8; ++e) { fr[0][t][pl][e] = (_Float16)0.01f; fr[1][t][pl][e] = (_Float16)0.02f; }
    rd(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            rd(b ^ 1, b ^ 1);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[b][mi][1], fr[b][2 + ni][0], acl[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[b][mi][0], fr[b][2 + ni][0], acc[mi][ni], 0, 0, 0);
                    acl[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[b][mi][0], fr[b][2 + ni][1], acl[mi][ni], 0, 0, 0);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r] + acl[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <int NREAD> void run(float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000; const size_t lds = 4 * XPLANE * 2;
    hipLaunchKernelGGL(k<NREAD>, dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NREAD>, dim3(512), dim3(256), lds, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("pipelined, %d fragment reads per k-step: %.3f ms  MFMA %.1f TFLOP/s (fp16 executed)\n", NREAD, ms, 512.0 * 4 * iters * 12 * 32768.0 / ms / 1e9);
}
int main() { float* d; hipMalloc(&d, 512 * 256 * 4); run<8>(d); run<4>(d); run<0>(d); return 0; }
