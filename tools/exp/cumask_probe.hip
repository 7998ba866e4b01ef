// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  Blocks record the XCC they ran on (HW_REG_XCC_ID);
// printed: blocks per XCC for a few masks (256 CUs = 8 words of 32 bits).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/cumask_probe.hip -o tools/exp/cumask_probe && timeout 60 tools/exp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void where(unsigned* cnt, int spin) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        atomicAdd(&cnt[xcc & 7], 1u);
        atomicAdd(&cnt[8 + ((xcc & 7) * 64 + ((hwid >> 8) & 15) + 16 * ((hwid >> 13) & 3)) % 512], 1u);   // (xcc, se?, cu) histogram, coarse
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    unsigned* cnt; (void)hipMalloc(&cnt, 520 * 4);
    unsigned h[520];
    auto run = [&](const char* name, const std::vector<uint32_t>& mask) {
        hipStream_t s;
        hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
        (void)hipMemsetAsync(cnt, 0, 520 * 4, s);
        hipLaunchKernelGGL(where, dim3(2048), dim3(64), 0, s, cnt, 200);
        e = hipStreamSynchronize(s);
        (void)hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s %s  blocks per XCC:", name, e == hipSuccess ? "ok " : hipGetErrorString(e));
        for (int i = 0; i < 8; ++i) printf(" %4u", h[i]);
        printf("\n");
        (void)hipStreamDestroy(s);
    };
    run("no mask", {});
    run("all 256 bits", std::vector<uint32_t>(8, 0xFFFFFFFFu));
    run("bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 32..63", {0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0});
    run("bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu});
    run("every 8th bit (0, 8, 16, ...)", std::vector<uint32_t>(8, 0x01010101u));
    run("every 8th bit, offset 3", std::vector<uint32_t>(8, 0x08080808u));
    run("bits 0..127", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0});
    return 0;
}
