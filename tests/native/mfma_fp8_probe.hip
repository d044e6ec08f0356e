// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) on gfx950: which (row / column, k) does byte t of lane l
// hold, where does D land, and what does a per-lane e8m0 scale multiply?  Prints the hypothesis that reproduces a host GEMM.
// build: hipcc --offload-arch=gfx950 -O2 tests/native/mfma_fp8_probe.hip -o /tmp/mfma_fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const unsigned char* a, const unsigned char* b, float* out, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = ((const int*)a)[l * 8 + i]; vb[i] = ((const int*)b)[l * 8 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

static float e4m3(unsigned char v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
static int kmap(int h, int g, int t) {
    switch (h) {
        case 0: return g * 32 + t;
        case 1: return (t / 16) * 64 + g * 16 + (t % 16);
        default: return (t / 8) * 32 + g * 8 + (t % 8);
    }
}

int main() {
    static const unsigned char vals[] = {0x00, 0x38, 0xb8, 0x40, 0xc0, 0x30, 0xb0, 0x44};   // 0, 1, -1, 2, -2, .5, -.5, 3
    unsigned char ha[64 * 32], hb[64 * 32];
    srand(7);
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = vals[rand() % 8]; hb[i] = vals[rand() % 8]; }
    int hsa[64], hsb[64];
    unsigned char *da, *db; float* dout; int *dsa, *dsb;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, 256 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    float out[256];
    for (int pass = 0; pass < 3; ++pass) {
        // pass 0: unit scales (e8m0 127); pass 1: lanes with g == 1 carry A scale 2^3; pass 2: lanes with g == 2 carry B scale 2^-2
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
        if (pass == 1) for (int l = 16; l < 32; ++l) hsa[l] = 130;
        if (pass == 2) for (int l = 32; l < 48; ++l) hsb[l] = 125;
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout, dsa, dsb);
        if (hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
        int found = 0;
        // unscaled products are invariant under any k permutation shared by A and B, so the k map only shows through the
        // scales: byte t of lane (x, g) is element k = kmap(h, g, t) of row / column x, and block j = k / 32 of row x takes
        // the scale that lane (x, j) supplies
        for (int h = 0; h < 3; ++h) for (int dl = 0; dl < 2; ++dl) {
            float A[16][128], B[128][16];
            for (int l = 0; l < 64; ++l) for (int t = 0; t < 32; ++t) {
                const int k = kmap(h, l >> 4, t), x = l & 15;
                A[x][k] = e4m3(ha[l * 32 + t]) * ldexpf(1.f, hsa[x + 16 * (k / 32)] - 127);
                B[k][x] = e4m3(hb[l * 32 + t]) * ldexpf(1.f, hsb[x + 16 * (k / 32)] - 127);
            }
            double err = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int row = dl == 0 ? (l >> 4) * 4 + r : l & 15, col = dl == 0 ? l & 15 : (l >> 4) * 4 + r;
                double s_ = 0;
                for (int k = 0; k < 128; ++k) s_ += (double)A[row][k] * B[k][col];
                err = fmax(err, fabs(s_ - out[l * 4 + r]));
            }
            if (err < 1e-3) { printf("pass %d: MATCH k-map H%d, D layout %s (max err %.2g)\n", pass, h, dl == 0 ? "row=(l>>4)*4+r col=l&15" : "row=l&15 col=(l>>4)*4+r", err); ++found; }
        }
        if (!found) printf("pass %d: no hypothesis matches; out[0..7] = %g %g %g %g %g %g %g %g\n", pass, out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7]);
    }
    return 0;
}
