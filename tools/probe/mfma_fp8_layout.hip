// Operand / result layout of v_mfma_f32_16x16x32_fp8_fp8 on gfx950 (OCP e4m3), determined empirically (development probe).
// Hypothesis: A lane l holds row i = l % 16, k = 8 (l / 16) + byte; B lane l holds column j = l % 16, k = 8 (l / 16) + byte;
// D lane l holds D[4 (l / 16) + v][l % 16].  A[i][k] = (1 + i) (k == ka), B[k][j] = (1 + j) (k == kb): D = (1+i)(1+j) iff ka == kb.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned char enc(float x) { return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.f, 0, false) & 0xff); }
__global__ void k(float* out, int ka, int kb) {
    const int l = threadIdx.x, r = l % 16, g = l / 16;
    unsigned long a = 0, b = 0;
    for (int by = 0; by < 8; ++by) {
        if (8 * g + by == ka) a |= (unsigned long)enc(1.f + r) << (8 * by);
        if (8 * g + by == kb) b |= (unsigned long)enc(1.f + r) << (8 * by);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    const int cases[4][2] = {{0, 0}, {13, 13}, {31, 31}, {13, 12}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int ok_std = 1, ok_alt = 1;
        for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
            const int j = l % 16, i1 = 4 * (l / 16) + v, i2 = (l / 16) + 4 * v;
            const float z = cs[0] == cs[1] ? 1.f : 0.f;
            if (h[l * 4 + v] != z * (1.f + i1) * (1 + j)) ok_std = 0;
            if (h[l * 4 + v] != z * (1.f + i2) * (1 + j)) ok_alt = 0;
        }
        printf("ka=%d kb=%d: D[4*(l/16)+v][l%%16] %d   D[(l/16)+4*v][l%%16] %d   lane 17: %.0f %.0f %.0f %.0f\n", cs[0], cs[1], ok_std, ok_alt, h[68], h[69], h[70], h[71]);
    }
    // e4m3 encodings the kernels rely on: 448 saturates, 2^-9 is the smallest subnormal, 0x00 = 0
    return 0;
}
