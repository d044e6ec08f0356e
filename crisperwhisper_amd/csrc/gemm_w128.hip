// Encoder tile GEMM, round 4 (see the comment at the kernel).  Launched from gemm.hip: launch_gemm_epi.
#include "common.h"
#include "kernels.h"
#include "gemm_tiles.h"
#include <mutex>

namespace CW_NS {

#ifdef CW_EXPERIMENTS   // measured slower than the 8-phase schedule at every encoder shape (profiles/r04_gemm_w128_rejected_ab.txt)
__device__ static inline f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return cw_mfma_16x16x32(a, b, c); }

// ---------------------------------------------------------------------------------------------------
// Round 4: the 256 x 256 x 64 tile with FOUR waves of 128 x 128 each (one wave per SIMD).  The 8-wave schedules above are bound by
// their fragment reads (24 ds_read_b128 per wave and K-tile for 64 MFMAs, section 6d ablation): a 128 x 64 per-wave tile reads
// 0.375 fragments per MFMA.  With 128 x 128 per wave a k-step of 32 is 8 + 8 fragments for 64 MFMAs (0.25 per MFMA) and the
// 256 accumulator registers live in the AGPR half of the unified file.  One wave per SIMD means no partner wave covers a stall,
// so the wave pipelines itself:
// in half steps of 32 MFMAs (8 A fragments x 4 W fragments), each of which requests the operands of the next one first:
//     read W(t, k0, cols 64..127)                    | 32 MFMAs  A(t, k0) x W(t, k0, cols 0..63)
//     read A(t, k1), W(t, k1, cols 0..63)            | 32 MFMAs  A(t, k0) x W(t, k0, cols 64..127)
//     read W(t, k1, cols 64..127)                    | 32 MFMAs  A(t, k1) x W(t, k1, cols 0..63)
//     vmcnt(0) lgkmcnt(0), barrier                   -> tile t + 1 has landed for every wave, every wave holds the rest of tile t
//     DMA tile t + 2 -> stage t % 2; read A(t + 1, k0), W(t + 1, k0, cols 0..63)   | 32 MFMAs  A(t, k1) x W(t, k1, cols 64..127)
// one barrier per K-tile, every LDS read and DMA issue sits under MFMAs that do not depend on it; 256 accumulator + 96 fragment
// registers.  Its own translation unit because it is built with -mllvm -amdgpu-mfma-vgpr-form: with the default register
// classes hipcc shuffles the 256 accumulators between the VGPR and AGPR halves around every MFMA block (1360 v_accvgpr moves
// and 32 spills per K-tile); with the VGPR form the steady-state loop is MFMAs, ds_reads and DMA issues.  Same LDS image (source-side
// XOR swizzle), same epilogues and the same per-element summation order as the other 256-tile kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------------
#define W128_BK 32          // K columns per stage: 256 rows x 64 B of A and of W = 32 KB
#define W128_NS 4           // stages in flight: a tile is requested three tiles (3 x 64 MFMAs = 3072 clocks) before it is read
template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_w128_kernel(const bf16_t* __restrict__ A, int lda,
                                                             const bf16_t* __restrict__ W, int M, int N, int K,
                                                             EpiParams ep, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm5[];   // [4][A 16 KB | W 16 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM2, n0 = nt_ * BN2;

    // LDS image of a stage: row r of the tile = 64 contiguous bytes at r * 64 (four 16-byte k-chunks); a fragment read touches
    // 16 consecutive rows x 4 chunks = exactly 1 KB, so it is conflict-free without any swizzle.
    // DMA map: wave w, load q (0..3): rows w*64 + q*16 .. +15; lane -> (row lane >> 2, chunk lane & 3): 1 KB per instruction
    const int lrow = lane >> 2, csrc = (lane & 3) * 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    int va[4];                                                   // rows beyond M clamped: they feed output rows that are never stored
#pragma unroll
    for (int q = 0; q < 4; ++q) va[q] = (min(m0 + wave * 64 + q * 16 + lrow, M - 1) * lda + csrc) * 2;
    const int vw = ((n0 + wave * 64 + lrow) * K + csrc) * 2;
    auto issue_tile = [&](int k0, int st) {
        unsigned char* base = gsm5 + st * 32768 + (wave * 64) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, va[q], k0 * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + 16384 + q * 1024), 16, vw,
                                                     (k0 + q * 16 * K) * 2, 0, 0);
        }
    };
    const int aoff = (wr * 128 + l15) * 64 + g * 16, woff = 16384 + (wc * 128 + l15) * 64 + g * 16;
    auto read_a = [&](int st, bf16x8_t (&fa)[8]) {
        const unsigned char* sb = gsm5 + st * 32768 + aoff;
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8_t*)(sb + i * 1024);
    };
    auto read_w = [&](int st, int h, bf16x8_t (&fw)[4]) {
        const unsigned char* sb = gsm5 + st * 32768 + woff + h * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *(const bf16x8_t*)(sb + j * 1024);
    };
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#define CW_W128_MFMA32(FA, FW, H)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][(H) * 4 + j] = mfma16(FW[j], FA[i], acc[i][(H) * 4 + j]);

    // Tile t (32 k) sits in stage t % 4.  Steady state, per tile:
    //     read W(t, cols 64..127)                              | 32 MFMAs  A(t) x W(t, cols 0..63)
    //     vmcnt(<= 2 tiles in flight) lgkmcnt(0), barrier       -> tile t + 1 complete for every wave; everyone holds all of tile t
    //     DMA tile t + 4 -> stage t % 4; read A(t + 1), W(t + 1, cols 0..63)   | 32 MFMAs  A(t) x W(t, cols 64..127)
    const int nk = K / W128_BK;
    bf16x8_t fa0[8], fa1[8], fw0[4], fw1[4];
#pragma unroll
    for (int t = 0; t < W128_NS; ++t)
        if (t < nk) issue_tile(t * W128_BK, t);
    // tile 0 landed (the later ones stay in flight: 8 loads per tile and wave)
    if (nk >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(0, fa0); read_w(0, 0, fw0);
    for (int t = 0; t < nk; t += 2) {                            // two tiles per trip: the fragment registers alternate
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tt = t + u;
            if (tt < nk) {
                const int st = tt & 3;
                bf16x8_t (&fa)[8] = u ? fa1 : fa0;
                bf16x8_t (&fan)[8] = u ? fa0 : fa1;
                read_w(st, 1, fw1);
                CW_W128_MFMA32(fa, fw0, 0)
                // tile tt + 1 must have landed: at most the two youngest tiles (16 loads) of this wave stay in flight
                if (tt + 3 < nk) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tt + W128_NS < nk) issue_tile((tt + W128_NS) * W128_BK, st);
                if (tt + 1 < nk) { read_a((tt + 1) & 3, fan); read_w((tt + 1) & 3, 0, fw0); }
                CW_W128_MFMA32(fa, fw1, 1)
            }
        }
    }
#undef CW_W128_MFMA32

    const bool vec_ok = (EPI == EPI_HEADS || (ep.ldo & 3) == 0);
    const bool interior = m0 + BM2 <= M && vec_ok;               // block-uniform
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            int rows[4], cols[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rows[i] = m0 + wr * 128 + (ih * 4 + i) * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) cols[j] = n0 + wc * 128 + (jh * 4 + j) * 16 + g * 4;
            if (interior) {
                f32x4_t sub[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sub[i][j] = acc[ih * 4 + i][jh * 4 + j];
                epi_tile_interior<bf16_t, EPI>(ep, rows, cols, sub);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = rows[i];
                    if (m >= M) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t a4 = acc[ih * 4 + i][jh * 4 + j];
                        if (vec_ok) epi_store4<bf16_t, EPI>(ep, m, cols[j], a4[0], a4[1], a4[2], a4[3]);
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) epi_store1<bf16_t, EPI>(ep, m, cols[j] + r, a4[r]);
                        }
                    }
                }
            }
        }
}



template <int EPI>
static void launch_w128(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const EpiParams& ep, int tm2, int tn2, hipStream_t st) {
    static std::once_flag attr;
    std::call_once(attr, [] { (void)hipFuncSetAttribute((const void*)gemm_bf16_w128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); });
    hipLaunchKernelGGL((gemm_bf16_w128_kernel<EPI>), dim3(tm2 * tn2), dim3(256), 131072, st, A, lda, W, M, N, K, ep, tn2);
}

int cw_launch_gemm_w128(int epi, const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const EpiParams& ep, int tm2, int tn2, hipStream_t st) {
    if (N % BN2 || K % 32 || M < 1) return CW_ERR_INVALID;
    switch (epi) {
        case EPI_STORE: launch_w128<EPI_STORE>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        case EPI_GELU: launch_w128<EPI_GELU>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        case EPI_RESID_F32: launch_w128<EPI_RESID_F32>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        case EPI_GELU_POS_F32: launch_w128<EPI_GELU_POS_F32>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        case EPI_HEADS: launch_w128<EPI_HEADS>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        case EPI_STORE_F32: launch_w128<EPI_STORE_F32>(A, lda, W, M, N, K, ep, tm2, tn2, st); break;
        default: return CW_ERR_INVALID;
    }
    return CW_OK;
}

#else
int cw_launch_gemm_w128(int, const bf16_t*, int, const bf16_t*, int, int, int, const EpiParams&, int, int, hipStream_t) { return CW_ERR_INVALID; }
#endif

}  // namespace CW_NS
