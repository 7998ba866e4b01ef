// Which (row, column) of D = A x B does register r of lane l hold for v_mfma_f64_16x16x4_f64?
// A[i][k] = 100 i + k, B[k][j] = (k == 0) * (j + 1): D[i][j] = 100 i * (j + 1) -> row and column read off the value.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
__global__ void probe(double* out, int amode) {
    const int l = threadIdx.x;
    // hypothesis for the inputs: A lane l = A[i = l % 16][k = l / 16], B lane l = B[k = l / 16][j = l % 16]
    const int i = l % 16, k = l / 16, j = l % 16;
    const double a = amode == 0 ? (k == 0 ? 1.0 + i : 0.0) : (k == 1 ? 1.0 + i : 0.0);
    const double b = (amode == 0 ? (k == 0) : (k == 1)) ? 100.0 * (j + 1) : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 8);
    double h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l : {0, 1, 15, 16, 17, 32, 63}) {
            printf("mode %d lane %2d:", mode, l);
            for (int r = 0; r < 4; ++r) { const int v = (int)h[l * 4 + r]; printf("  r%d -> row %d col %d", r, (v % 100) - 1, v / 100 - 1); }
            printf("\n");
        }
    }
    return 0;
}
