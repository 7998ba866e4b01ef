// What does a wave's vector work cost beside another wave's MFMA stream on the SAME SIMD?  (the ring kernel's question:
// its cell role alone takes ~1100 cycles, its MFMA role alone ~1100, side by side on one SIMD ~2150)
//
// One workgroup of 512 threads on one CU = two waves per SIMD.  Waves 0-3 (one per SIMD) run role A, waves 4-7 role B:
//   A: `na` rounds of nine independent v_mfma_f32_16x16x32_f16 (the ring's MFMA item: 3 gates x 3 products)
//   B: `nb` rounds of a vector sequence chosen by `mode`:
//        0 one dependent v_fma_f32 chain of 64            1 four independent chains of 16 (same 64 instructions)
//        2 dependent chain of 16 v_exp_f32 + 16 v_rcp_f32  3 64 v_mov-like independent adds (no dependency at all)
//        4 32 ds_read_b32 + adds (LDS latency chain)       5 the GRU cell's shape: 12 LDS reads, 2 exp, 3 rcp, ~60 dependent VALU
// Each wave stamps s_memrealtime around its work (100 MHz: x 24 = shader cycles at 2.4 GHz); printed: per role the mean over
// its four waves, for A alone, B alone, and both together.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/coissue_probe.hip -o tools/exp/coissue_probe && tools/exp/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512, 1) void probe(int na, int nb, int mode, unsigned long long* t, float* sink, int amode, const f16x8* wsrc) {
    __shared__ __attribute__((aligned(16))) float lds[4096 + 8 * 14 * 256];     // B's words + a 14-KiB state slice per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = 0.001f * i;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    float keep = 0.f;
    if (w < 4) {
        f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x4 c[9];
        for (int i = 0; i < 9; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // amode 1: the ring kernel's item -- W_hh of 7 k-blocks x 3 gates x 2 planes in registers (168), the state operands of a
        // k-block read from LDS one block ahead of the MFMAs that use them, 6 accumulators, twelve partial-tile words written back
        f16x8 wv[7][3][2];
        if (amode == 1)
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wv[i][g][pl] = wsrc[((i * 3 + g) * 2 + pl) * 64 + lane];
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(lds + 4096) + w * 14336 + lane * 16;
        float* red = lds + 4096 + 4 * 3584 + w * 1024;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        t0 = __builtin_amdgcn_s_memrealtime();
        for (int r = 0; r < na; ++r) {
            if (amode == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            } else {
                f32x4 acc[3], acl[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                f16x8 bc[2], bn[2];
                bc[0] = *reinterpret_cast<const f16x8*>(sb);
                bc[1] = *reinterpret_cast<const f16x8*>(sb + 1024);
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if (i + 1 < 7) {
                        bn[0] = *reinterpret_cast<const f16x8*>(sb + (i + 1) * 2048);
                        bn[1] = *reinterpret_cast<const f16x8*>(sb + (i + 1) * 2048 + 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 3; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][1], bc[0], acl[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], bc[0], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 3; ++g) acl[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[i][g][0], bc[1], acl[g], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bc[0] = bn[0]; bc[1] = bn[1];
                }
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[(g * 16 + 4 * (lane >> 4) + q) * 20 + (lane & 15)] = acc[g][q] + acl[g][q] * 0.00048828125f;
            }
        }
        for (int i = 0; i < 9; ++i) keep += c[i][0];
        asm volatile("" ::"v"(keep));
        t1 = __builtin_amdgcn_s_memrealtime();
    } else {
        float x = 1.0f + 0.001f * lane, y = 0.5f, z = 0.25f, u = 0.125f;
        const float m = 0.999f, q = 0.001f;
        __builtin_amdgcn_s_barrier();
        t0 = __builtin_amdgcn_s_memrealtime();
        for (int r = 0; r < nb; ++r) {
            if (mode == 0) {
#pragma unroll
                for (int i = 0; i < 64; ++i) x = __builtin_fmaf(x, m, q);
            } else if (mode == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { x = __builtin_fmaf(x, m, q); y = __builtin_fmaf(y, m, q); z = __builtin_fmaf(z, m, q); u = __builtin_fmaf(u, m, q); }
            } else if (mode == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { x = __builtin_amdgcn_exp2f(x * 0.01f); x = __builtin_amdgcn_rcpf(x + 1.0f); }
            } else if (mode == 3) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = x + i;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = v[i] * m;
#pragma unroll
                for (int i = 0; i < 16; ++i) y += v[i];
                x = y * 1e-6f + 1.f;
            } else if (mode == 4) {
#pragma unroll
                for (int i = 0; i < 32; ++i) { x += lds[(lane * 4 + i * 67 + (int)(x) * 0) & 4095]; asm volatile("" : "+v"(x)); }
                x = x * 1e-6f + 1.f;
            } else {
                // GRU-cell-shaped: twelve partial sums from LDS, three gates, sigmoid x2 (exp + rcp), tanh (exp + rcp), blend
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { s0 += lds[(lane + 64 * i) & 4095]; s1 += lds[(lane + 64 * (i + 4)) & 4095]; s2 += lds[(lane + 64 * (i + 8)) & 4095]; }
                const float rg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-(s0 + x) * 1.4427f));
                const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-(s1 + y) * 1.4427f));
                const float pre = z + rg * s2;
                const float e2 = __builtin_amdgcn_exp2f(pre * 2.8854f);
                const float ng = 1.f - 2.f * __builtin_amdgcn_rcpf(e2 + 1.f);
                float h = ng + zg * (u - ng);
#pragma unroll
                for (int i = 0; i < 24; ++i) h = __builtin_fmaf(h, m, q);      // (split, pack, publish arithmetic)
                u = h; x = h * 0.5f; y = h * 0.25f; z = h * 0.125f;
            }
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(u));
        }
        keep = x + y + z + u;
        asm volatile("" ::"v"(keep));
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    if (lane == 0) { t[w * 2] = t0; t[w * 2 + 1] = t1; }
    if (keep == 123.456f) sink[tid] = keep;
}

int main() {
    unsigned long long* t; float* sink;
    (void)hipMalloc(&t, 16 * 8); (void)hipMalloc(&sink, 512 * 4);
    unsigned long long h[16];
    f16x8* wsrc; (void)hipMalloc(&wsrc, 42 * 64 * 16); (void)hipMemset(wsrc, 0x3c, 42 * 64 * 16);
    int amode = 0;
    auto run = [&](int na, int nb, int mode, double* a_cyc, double* b_cyc) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, na, nb, mode, t, sink, amode, wsrc);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < 4; ++w) { a += (double)(h[w * 2 + 1] - h[w * 2]); b += (double)(h[(w + 4) * 2 + 1] - h[(w + 4) * 2]); }
        *a_cyc = a / 4 * 24.0; *b_cyc = b / 4 * 24.0;     // 100 MHz ticks -> cycles at 2.4 GHz
    };
    const char* names[] = {"dependent fma chain x64", "four independent fma chains x16", "exp+rcp chain x16", "independent VALU x80", "LDS read chain x32", "GRU-cell shape"};
    for (amode = 0; amode < 2; ++amode) {
        const int NA = amode == 0 ? 2000 : 286;                  // 18000 MFMAs per A wave either way (286 items of 63)
        const int per = amode == 0 ? 9 : 63;
        double a0, b0, a1, b1, a2, b2;
        run(NA, 0, 0, &a0, &b0);
        printf("A alone (%s): %d rounds of %d MFMAs: %.0f cycles = %.2f cycles per MFMA, %.0f per round\n", amode == 0 ? "register operands" : "ring item: operands from LDS",
               NA, per, a0, a0 / (NA * (double)per), a0 / NA);
        for (int mode = 0; mode < 6; ++mode) {
            const int NB = 2000;
            run(0, NB, mode, &a1, &b1);
            const int nbb = (int)(NB * (a0 / b1));       // B sized so that both roles take about as long alone
            run(0, nbb, mode, &a1, &b1);
            run(NA, nbb, mode, &a2, &b2);
            printf("  mode %d (%s): B alone %.0f cycles (%.1f per round); together: A %.0f (x%.2f), B %.0f (x%.2f); sum alone %.0f, max together %.0f\n",
                   mode, names[mode], b1, b1 / nbb, a2, a2 / a0, b2, b2 / b1, a0 + b1, a2 > b2 ? a2 : b2);
        }
    }
    return 0;
}
