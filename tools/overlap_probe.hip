// Probe: dependent chain of GEMV-like kernels, (a) ordinary single-stream graph vs (b) two-stream graph where kernel
// i+1 is launched concurrently, prefetches its weights into registers and then spin-waits on a device counter that
// kernel i bumps at its end.  Prints microseconds per kernel for both.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 1280, ROWS = 8, NT = 256;

template <bool DEP>
__global__ __launch_bounds__(256) void gemv_like(const unsigned short* __restrict__ W, const float* __restrict__ x,
                                                 float* __restrict__ y, int N, const int* wait_ctr, int wait_val,
                                                 int* done_ctr, int* err) {
    __shared__ float xs[ROWS][K];
    __shared__ float red[4][ROWS][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int n = blockIdx.x * 16 + l15;
    // weights: column n, this lane covers k = (wave*?)...: 1280 = 10 steps of 128; wave handles steps wave, wave+4, wave+8
    u32x4 wq[3][4];
    const unsigned short* wrow = W + (size_t)n * K + g * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int step = wave + 4 * s; step = step < 10 ? step : 9;
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[s][j] = *(const u32x4*)(wrow + step * 128 + j * 32);
    }
    if (DEP) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_val) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 18)) { *err = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    for (int i = tid; i < ROWS * K / 4; i += 256) ((float4*)&xs[0][0])[i] = ((const float4*)x)[i];
    __syncthreads();
    float acc[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) acc[m] = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int step = wave + 4 * s;
        if (step < 10) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = step * 128 + j * 32 + g * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned u = wq[s][j][e];
                    const float w0 = __uint_as_float(u << 16), w1 = __uint_as_float(u & 0xffff0000u);
#pragma unroll
                    for (int m = 0; m < ROWS; ++m) acc[m] += w0 * xs[m][k0 + 2 * e] + w1 * xs[m][k0 + 2 * e + 1];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
        float v = acc[m];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (g == 0) red[wave][m][l15] = v;
    }
    __syncthreads();
    if (tid < ROWS * 16) {
        const int m = tid >> 4, c = tid & 15;
        const float v = red[0][m][c] + red[1][m][c] + red[2][m][c] + red[3][m][c];
        const int nn = blockIdx.x * 16 + c;
        if (nn < K) atomicAdd(y + m * K + nn, v * 1e-3f);
    }
    if (DEP) {
        __syncthreads();
        if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); atomicAdd(done_ctr, 1); }
    }
}

int main(int argc, char** argv) {
    const int chain = 256, N = argc > 1 ? atoi(argv[1]) : 5120;   // N columns -> N/16 blocks
    const int nblk = N / 16;
    unsigned short* W; float *x0, *x1; int *ctr, *err;
    const int NW = 64;   // distinct weight matrices (cycle) so that weights come from HBM, not L2
    CK(hipMalloc(&W, (size_t)NW * N * K * 2)); CK(hipMemset(W, 0x3c, (size_t)NW * N * K * 2));
    CK(hipMalloc(&x0, ROWS * K * 4)); CK(hipMalloc(&x1, ROWS * K * 4));
    CK(hipMemset(x0, 0, ROWS * K * 4)); CK(hipMemset(x1, 0, ROWS * K * 4));
    CK(hipMalloc(&ctr, (chain + 1) * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int mode = 0; mode < 2 && !(argc > 2); ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
        if (mode == 1) {
            CK(hipMemsetAsync(ctr, 0, (chain + 1) * 4, sa));
            CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0));
        }
        for (int i = 0; i < chain; ++i) {
            const unsigned short* Wi = W + (size_t)(i % NW) * N * K;
            float* xi = (i & 1) ? x1 : x0; float* yi = (i & 1) ? x0 : x1;
            if (mode == 0) hipLaunchKernelGGL((gemv_like<false>), dim3(nblk), dim3(256), 0, sa, Wi, xi, yi, N, nullptr, 0, nullptr, err);
            else hipLaunchKernelGGL((gemv_like<true>), dim3(nblk), dim3(256), 0, (i & 1) ? sb : sa, Wi, xi, yi, N, ctr + i,
                                    i == 0 ? 0 : nblk, ctr + i + 1, err);
        }
        if (mode == 1) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, sa)); CK(hipStreamSynchronize(sa));
        CK(hipEventRecord(e0, sa));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, sa));
        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per kernel (N=%d, %d blocks) err=%d\n", mode, mode ? "two-stream flags + weight prefetch" : "single-stream graph",
               ms * 1000.f / (5 * chain), N, nblk, herr);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    for (int mode = 2; mode < 4; ++mode) {   // eager launches: 2 = single stream, 3 = two streams + flags
        auto run = [&]() {
            if (mode == 3) CK(hipMemsetAsync(ctr, 0, (chain + 1) * 4, sa));
            if (mode == 3) { CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0)); }
            for (int i = 0; i < chain; ++i) {
                const unsigned short* Wi = W + (size_t)(i % NW) * N * K;
                float* xi = (i & 1) ? x1 : x0; float* yi = (i & 1) ? x0 : x1;
                if (mode == 2) hipLaunchKernelGGL((gemv_like<false>), dim3(nblk), dim3(256), 0, sa, Wi, xi, yi, N, nullptr, 0, nullptr, err);
                else hipLaunchKernelGGL((gemv_like<true>), dim3(nblk), dim3(256), 0, (i & 1) ? sb : sa, Wi, xi, yi, N, ctr + i,
                                        i == 0 ? 0 : nblk, ctr + i + 1, err);
            }
            if (mode == 3) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
        };
        run(); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(e0, sa));
        for (int r = 0; r < 5; ++r) run();
        CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per kernel (N=%d) err=%d\n", mode, mode == 3 ? "eager two-stream flags + prefetch" : "eager single stream",
               ms * 1000.f / (5 * chain), N, herr);
    }
    float h[4]; CK(hipMemcpy(h, x0, 16, hipMemcpyDeviceToHost)); printf("x0[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
    return 0;
}
