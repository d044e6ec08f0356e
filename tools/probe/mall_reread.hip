// Go / no-go probe for "cross-attention over the encoder states instead of the per-layer K/V cache" (round-4 review, item 7):
// all 32 decoder layers would re-read ONE buffer -- the encoder output of the batch: 30.7 MB at 8 sequences, 123 MB (e4m3) or
// 246 MB (bf16) at 64 -- with the layer's 57 MB weight stream going by between two reads.  Does the re-read come out of the
// 256 MB Infinity Cache fast enough (>= 12 TB/s was the bar) to beat streaming 492 MB of K/V per layer at 6.4 TB/s?
// Measures: per-kernel time of 32 x { read the shared buffer ; read a distinct 57 MB slice of a 1.8 GB pool }, by hipEvents around
// every kernel, for several buffer sizes and both load policies of the weight stream (default / non-temporal).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mall_reread.hip -o tools/probe/mall_reread_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u4* __restrict__ p, size_t n16, unsigned* sink) {
    u4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // eight loads in flight per lane
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int L = 32;
    const size_t wbytes = (size_t)57 << 20;
    unsigned char* pool; unsigned* sink;
    CK(hipMalloc(&pool, wbytes * L)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(pool, 1, wbytes * L));
    const size_t sizes[] = {(size_t)31 << 20, (size_t)62 << 20, (size_t)123 << 20, (size_t)185 << 20, (size_t)246 << 20, (size_t)492 << 20};
    std::vector<hipEvent_t> ev(2 * L + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    printf("# shared buffer re-read by 32 kernels, a distinct 57 MB weight slice streamed between two reads (2048 blocks x 256 threads, 16 B per lane)\n");
    printf("# %-10s %-8s %12s %12s %12s %14s\n", "buffer", "weights", "reread us", "reread TB/s", "weights us", "weights TB/s");
    for (size_t sz : sizes) {
        unsigned char* buf;
        CK(hipMalloc(&buf, sz)); CK(hipMemset(buf, 2, sz));
        for (int nt = 0; nt < 3; ++nt) {     // 0: weights by default-policy loads, 1: non-temporal, 2: no weight stream between the reads
            for (int rep = 0; rep < 2; ++rep) {     // second repetition is the one reported (steady state)
                CK(hipEventRecord(ev[0], 0));
                for (int l = 0; l < L; ++l) {
                    hipLaunchKernelGGL(read_kernel<false>, dim3(2048), dim3(256), 0, 0, (const u4*)buf, sz / 16, sink);
                    CK(hipEventRecord(ev[2 * l + 1], 0));
                    if (nt == 2) { }
                    else if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(2048), dim3(256), 0, 0, (const u4*)(pool + wbytes * l), wbytes / 16, sink);
                    else hipLaunchKernelGGL(read_kernel<false>, dim3(2048), dim3(256), 0, 0, (const u4*)(pool + wbytes * l), wbytes / 16, sink);
                    CK(hipEventRecord(ev[2 * l + 2], 0));
                }
                CK(hipDeviceSynchronize());
            }
            std::vector<float> tb, tw;
            for (int l = 4; l < L; ++l) {
                float a, b;
                CK(hipEventElapsedTime(&a, ev[2 * l], ev[2 * l + 1])); CK(hipEventElapsedTime(&b, ev[2 * l + 1], ev[2 * l + 2]));
                tb.push_back(a); tw.push_back(b);
            }
            std::sort(tb.begin(), tb.end()); std::sort(tw.begin(), tw.end());
            const double mb = tb[tb.size() / 2] * 1e3, mw = tw[tw.size() / 2] * 1e3;
            printf("  %-10zu %-8s %12.2f %12.2f %12.2f %14.2f\n", sz >> 20, nt == 2 ? "none" : nt ? "nt" : "default", mb, sz / mb / 1e6, mw, wbytes / mw / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
