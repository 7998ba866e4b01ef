// Experiment: accuracy of fp32-grade products on the low-precision MFMA: two-term fp16 split (3 products), three-term
// bf16 split (6 and 3 products) and the plain fp32 MFMA, all against an fp64 reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// two fp16 terms: x = hi + lo * 2^-11 (lo kept scaled so that it stays a normal fp16 number)
__device__ inline void split2h(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x; l = (_Float16)((x - (float)h) * 2048.f);
}

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}

// C[32][32] = A[32][K] * B[32][K]^T ; one wave
__global__ void kh(const float* A, const float* B, float* CH3, float* CH4, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 hh, hl, ll;
    for (int i = 0; i < 16; ++i) { hh[i] = 0; hl[i] = 0; ll[i] = 0; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            _Float16 x, y;
            split2h(A[r * K + k0 + 8 * h + j], x, y); ah[j] = x; al[j] = y;
            split2h(B[r * K + k0 + 8 * h + j], x, y); bh[j] = x; bl[j] = y;
        }
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hh, 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, hl, 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, hl, 0, 0, 0);
        ll = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, ll, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        CH3[row * 32 + r] = hh[i] + hl[i] * (1.f / 2048.f);
        CH4[row * 32 + r] = hh[i] + (hl[i] + ll[i] * (1.f / 2048.f)) * (1.f / 2048.f);
    }
}

__global__ void k(const float* A, const float* B, float* C6, float* C3, float* C32, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 a6, a3, a32;
    for (int i = 0; i < 16; ++i) { a6[i] = 0; a3[i] = 0; a32[i] = 0; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(A[r * K + k0 + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
            split3(B[r * K + k0 + 8 * h + j], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
        }
        // small terms first
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a6, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a3, 0, 0, 0);
        for (int j = 0; j < 8; ++j)
            a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + 2 * j + h], B[r * K + k0 + 2 * j + h], a32, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C6[row * 32 + r] = a6[i]; C3[row * 32 + r] = a3[i]; C32[row * 32 + r] = a32[i];
    }
}

int main(int argc, char** argv) {
    const int K = 800;
    std::mt19937 g(1);
    const float hs = argc > 1 ? atof(argv[1]) : 1.f;     // scale of the state operand (activations up to 20 in the conv / x-projection)
    std::uniform_real_distribution<float> uw(-0.035f, 0.035f), uh(-hs, hs);
    std::vector<float> A(32 * K), B(32 * K);
    for (auto& v : A) v = uw(g);
    for (auto& v : B) v = uh(g);
    float *dA, *dB, *d6, *d3, *d32;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d6, 4096); hipMalloc(&d3, 4096); hipMalloc(&d32, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d6, d3, d32, K);
    std::vector<float> c6(1024), c3(1024), c32(1024);
    hipMemcpy(c6.data(), d6, 4096, hipMemcpyDeviceToHost); hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost); hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
    float *dh3, *dh4; hipMalloc(&dh3, 4096); hipMalloc(&dh4, 4096);
    hipLaunchKernelGGL(kh, dim3(1), dim3(64), 0, 0, dA, dB, dh3, dh4, K);
    std::vector<float> ch3(1024), ch4(1024);
    hipMemcpy(ch3.data(), dh3, 4096, hipMemcpyDeviceToHost); hipMemcpy(ch4.data(), dh4, 4096, hipMemcpyDeviceToHost);
    double eh3 = 0, eh4 = 0;
    double e6 = 0, e3 = 0, e32 = 0, mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)A[i * K + kk] * (double)B[j * K + kk];
        e6 = fmax(e6, fabs(c6[i * 32 + j] - ref)); e3 = fmax(e3, fabs(c3[i * 32 + j] - ref)); e32 = fmax(e32, fabs(c32[i * 32 + j] - ref));
        mag = fmax(mag, fabs(ref));
        eh3 = fmax(eh3, fabs(ch3[i * 32 + j] - ref)); eh4 = fmax(eh4, fabs(ch4[i * 32 + j] - ref));
    }
    printf("f16x3=%.3e  f16x4=%.3e\n", eh3, eh4);
    printf("max|ref|=%.4f  err bf16x6=%.3e  bf16x3=%.3e  f32mfma=%.3e\n", mag, e6, e3, e32);
    return 0;
}
