// Does the instruction offset of global_load_lds apply to the LDS address as well as to the global address?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* g, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    const unsigned voff = threadIdx.x * 16;
    const unsigned la = (unsigned)(size_t)(lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 sc1" :: "s"(la), "v"(voff), "s"(g) : "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *g, *o; hipMalloc(&g, 16384); hipMalloc(&o, 16384);
    unsigned h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(g, h, 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, g, o);
    hipMemcpy(h, o, 16384, hipMemcpyDeviceToHost);
    int first = -1; for (int i = 0; i < 4096; ++i) if (h[i] != 0xdeadbeef) { first = i; break; }
    printf("first written lds word %d holds %u (global word); offset:1024 bytes = word 256\n", first, first >= 0 ? h[first] : 0);
    return 0;
}
