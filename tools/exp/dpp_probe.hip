// Which way does row_shl move data?  out[lane] = value lane received with row_shl:1 / row_shl:3 (source = lane id).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int a = __builtin_amdgcn_update_dpp(-1, lane, 0x101, 0xF, 0xF, false);   // row_shl:1
    const int b = __builtin_amdgcn_update_dpp(-1, lane, 0x103, 0xF, 0xF, false);   // row_shl:3
    const int c = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xF, 0xF, false);   // row_shr:1
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
}
int main() {
    int* o; (void)hipMalloc(&o, 192 * 4); int h[192];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; ++r) { printf("%s:", r == 0 ? "row_shl:1" : r == 1 ? "row_shl:3" : "row_shr:1"); for (int i = 0; i < 20; ++i) printf(" %d", h[64 * r + i]); printf("\n"); }
    return 0;
}
