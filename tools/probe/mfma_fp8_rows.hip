// Is a row of v_mfma_f32_16x16x32_fp8_fp8's result a function of that row of A alone?  (development probe, gfx950)
// The e4m3 cross-attention kernels carry a query as three e4m3 terms in rows 0..2 of the A operand; the beam-search form
// (attn_cross_mfma8_rows_kernel) fills the other rows with the terms of the other hypotheses.  Per trial: random e4m3 A (16 x 32)
// and B (32 x 16) bytes in the kernels' value ranges, D_full = A B (+ a second MFMA accumulating on top, as the kernels chain
// them), D_3 = the same with rows 3..15 of A zeroed; rows 0..2 are compared bit for bit.  The host also evaluates the exact sum
// of the (exact) e4m3 products in double and rounds it to f32: is the instruction's result the correctly rounded one?
// build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_fp8_rows.hip -o tools/probe/mfma_fp8_rows_bin && tools/probe/mfma_fp8_rows_bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [trial][2][16][32] bytes, B [trial][2][32][16] bytes (k-major rows of 16 columns); out [trial][variant 3][lane 64][4]
__global__ void probe(const unsigned char* A, const unsigned char* B, float* out, int keep_rows) {
    const int t = blockIdx.x, l = threadIdx.x, r = l & 15, g = l >> 4;
    unsigned long a[2], a3[2], b[2];
    for (int m = 0; m < 2; ++m) {
        a[m] = 0; b[m] = 0;
        for (int by = 0; by < 8; ++by) {
            const int k = 8 * g + by;
            a[m] |= (unsigned long)A[((size_t)(t * 2 + m) * 16 + r) * 32 + k] << (8 * by);
            b[m] |= (unsigned long)B[((size_t)(t * 2 + m) * 32 + k) * 16 + r] << (8 * by);
        }
        a3[m] = r < keep_rows ? a[m] : 0ul;
    }
    f32x4 c = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a[0], (long)b[0], c, 0, 0, 0);
    c1 = c;                                                         // one MFMA alone
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a[1], (long)b[1], c, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a3[0], (long)b[0], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a3[1], (long)b[1], c3, 0, 0, 0);
    for (int v = 0; v < 4; ++v) {
        out[(((size_t)t * 3 + 0) * 64 + l) * 4 + v] = c[v];
        out[(((size_t)t * 3 + 1) * 64 + l) * 4 + v] = c3[v];
        out[(((size_t)t * 3 + 2) * 64 + l) * 4 + v] = c1[v];
    }
}

static double e4m3_val(unsigned char x) {
    const int s = x >> 7, e = (x >> 3) & 15, m = x & 7;
    const double v = e == 0 ? ldexp((double)m, -9) : ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}
static unsigned char rnd_byte(int mode) {
    for (;;) {
        unsigned char x = (unsigned char)(rand() & 0xff);
        if ((x & 0x7f) == 0x7f) continue;                           // NaN encodings
        if (mode == 1 && ((x >> 3) & 15) > 11) continue;           // moderate exponents only
        if (mode == 2) x &= 0x7f;                                   // non-negative (probabilities x V of one sign: no cancellation)
        return x;
    }
}

int main() {
    const int T = 4096;
    std::vector<unsigned char> hA((size_t)T * 2 * 16 * 32), hB((size_t)T * 2 * 32 * 16);
    unsigned char *dA, *dB; float* dO;
    hipMalloc(&dA, hA.size()); hipMalloc(&dB, hB.size()); hipMalloc(&dO, (size_t)T * 3 * 64 * 4 * 4);
    std::vector<float> h((size_t)T * 3 * 64 * 4);
    for (int mode = 0; mode < 3; ++mode) {
        srand(1234 + mode);
        for (auto& x : hA) x = rnd_byte(mode);
        for (auto& x : hB) x = rnd_byte(mode);
        hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice);
        for (int keep : {3, 4}) {
            hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, dA, dB, dO, keep);
            hipMemcpy(h.data(), dO, h.size() * 4, hipMemcpyDeviceToHost);
            long n = 0, diff = 0, inexact1 = 0, inexact2 = 0, n1 = 0; double worst = 0, worst_rel1 = 0;
            for (int t = 0; t < T; ++t)
                for (int l = 0; l < 64; ++l)
                    for (int v = 0; v < 4; ++v) {
                        const int i = 4 * (l >> 4) + v, j = l & 15;
                        const float full = h[(((size_t)t * 3 + 0) * 64 + l) * 4 + v], three = h[(((size_t)t * 3 + 1) * 64 + l) * 4 + v];
                        const float one = h[(((size_t)t * 3 + 2) * 64 + l) * 4 + v];
                        double ex1 = 0, ex2 = 0;
                        for (int k = 0; k < 32; ++k) {
                            ex1 += e4m3_val(hA[((size_t)(t * 2 + 0) * 16 + i) * 32 + k]) * e4m3_val(hB[((size_t)(t * 2 + 0) * 32 + k) * 16 + j]);
                            ex2 += e4m3_val(hA[((size_t)(t * 2 + 1) * 16 + i) * 32 + k]) * e4m3_val(hB[((size_t)(t * 2 + 1) * 32 + k) * 16 + j]);
                        }
                        ++n1;
                        if ((float)ex1 != one) { ++inexact1; if (ex1 != 0) worst_rel1 = fmax(worst_rel1, fabs(one - ex1) / fabs(ex1)); }
                        if ((float)((double)(float)ex1 + ex2) != full && (float)(ex1 + ex2) != full) ++inexact2;
                        if (i < keep) { ++n; if (full != three) { ++diff; worst = fmax(worst, fabs((double)full - three) / fmax(fabs((double)full), 1e-30)); } }
                    }
            printf("mode %d (0 any e4m3, 1 exponents <= 11, 2 non-negative) keep rows 0..%d: rows compared %ld, DIFFER with the other rows zeroed: %ld (worst rel %.3e);  "
                   "single MFMA != correctly rounded exact sum: %ld of %ld (worst rel %.3e); chained pair != either rounding order: %ld\n",
                   mode, keep - 1, n, diff, worst, inexact1, n1, worst_rel1, inexact2);
        }
    }
    return 0;
}
