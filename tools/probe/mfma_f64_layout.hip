// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, determined empirically (development probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double f64x4_t;
__global__ void k(double* out) {
    const int l = threadIdx.x;
    // hypothesis: A lane l holds A[i = l % 16][k = l / 16]; B lane l holds B[k = l / 16][j = l % 16]
    const int i = l % 16, kk = l / 16, j = l % 16;
    const double a = 1.0 + i + 100.0 * kk;          // A[i][k] = 1 + i + 100 k
    const double b = (kk == 0 ? 1.0 : 0.0) * (1.0 + j) + (kk == 2 ? 1000.0 * (1 + j) : 0.0);   // B[0][j] = 1 + j, B[2][j] = 1000 (1 + j)
    f64x4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][0] B[0][j] + A[i][2] B[2][j] = (1 + i)(1 + j) + (201 + i) * 1000 (1 + j)
    int ok_std = 1, ok_alt = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        int j = l % 16;
        int i1 = 4 * (l / 16) + v, i2 = (l / 16) + 4 * v;
        double e1 = (1.0 + i1) * (1 + j) + (201.0 + i1) * 1000.0 * (1 + j), e2 = (1.0 + i2) * (1 + j) + (201.0 + i2) * 1000.0 * (1 + j);
        if (h[l * 4 + v] != e1) ok_std = 0;
        if (h[l * 4 + v] != e2) ok_alt = 0;
    }
    printf("D[4*(l/16)+v][l%%16]: %d   D[(l/16)+4*v][l%%16]: %d\n", ok_std, ok_alt);
    for (int l = 0; l < 64; l += 15) printf("lane %d: %.0f %.0f %.0f %.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
