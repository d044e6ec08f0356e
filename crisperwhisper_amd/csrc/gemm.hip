// GEMM family for the Whisper encoder/decoder on gfx950.
//
//   gemm_bf16_glds_kernel  C[M,N] = A[M,K] * W[N,K]^T -- 128x128x64 tiles, MFMA 16x16x32 bf16, operand tiles moved
//                      HBM -> LDS by global_load_lds_dwordx4 (double buffered, XOR swizzle on the DMA source).
//                      A rows are either plain (lda) or an im2col-free conv1d(k=3,pad=1) gather over a time-major
//                      activation (implicit GEMM for conv1 / conv2, TF/models/whisper/modeling_whisper.py:618-619).
//                      Operands are swapped in the MFMA (D = W_frag x A_frag) so every lane owns 4 *consecutive
//                      output columns* of one row -> vectorised epilogues (epi_store4).
//   gemm_bf16_256_kernel  the same with 256x256x64 tiles and 8 waves (128x64 per wave) for the large encoder shapes
//                      (>= 200 tiles): below the per-CU LDS-read and L1->LDS limits the 128 tile sits on.
//   gemm_bf16_pp_kernel   the 256x256x64 tile on a ping-pong schedule (plain row-major A, N % 256 == 0 -- the encoder's linear
//                      layers and the cross-K/V projection): the two waves of a SIMD run half a K-tile apart, one issuing its
//                      64 MFMAs from registers while the other reads the next tile's fragments and issues buffer_load-to-LDS DMA.
//   gemm_bf16_kernel   register-staged variant of the 128 tiling (fallback, CW_NO_GLDS=1).
//   gemm_f32_kernel    same contract in plain f32 VALU (parity mode + on-device reference).
//   gemv2_bf16_kernel  decode-time skinny GEMM (batch rows <= 16 per launch, row groups beyond): weights streamed once
//                      from HBM straight into MFMA B fragments, activations pulled to registers with a wave-local
//                      LayerNorm (TF modeling_whisper.py:470,485,498) and parked in LDS as bf16; in-place residual
//                      epilogue with K split + f32 atomics (exact on the 2^-12 residual grid); optional combination of
//                      split-attention partials; one or two 16-column tiles per block.
//   gemv_prep_kernel / gemv_mt_kernel  17..64 batch rows: activations laid out once in MFMA fragment-major order,
//                      one weight pass feeding up to 4 MFMA row tiles.
//   gemv_bf16_kernel   first-generation decode GEMV (any batch <= 64, K chunks through LDS); kept for shapes
//                      gemv2 does not take.
//   gemv_f32_kernel    f32 parity flavour of the same.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <mutex>
#include "gemm_tiles.h"

namespace CW_NS {

#define BM 128
#define BN 128
#define LDS_STRIDE 72  // bf16 per LDS row: 64 + 8 pad (144 B) keeps ds_read_b128 fragment reads spread

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ inline f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return cw_mfma_16x16x32(a, b, c); }

// Address of A row `m`, starting at K offset k0 (k0 % 64 == 0).  Returns nullptr for a zero row.
template <typename TA>
__device__ inline const TA* a_row_ptr(const AParams& ap, int m, int M, int k0) {
    if (m >= M) return nullptr;
    if (ap.amode == 0) return (const TA*)ap.A + (size_t)m * ap.lda + k0;
    int b = m / ap.T_out, t = m - b * ap.T_out;
    int tap = k0 / ap.C_in, c0 = k0 - tap * ap.C_in;
    int t_in = t * ap.stride + tap - 1;
    if (t_in < 0 || t_in >= ap.row_valid[b]) return nullptr;
    return (const TA*)ap.A + ((size_t)(ap.row_off[b] + t_in)) * ap.C_in + c0;
}


template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(AParams ap, const bf16_t* __restrict__ W, int M, int N,
                                                        int K, EpiParams ep, int tiles_n) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t sW[BN * LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    // staging: 128 rows x 64 cols bf16 = 1024 16-byte chunks per operand; 4 per thread.
    uint4 ra[4], rw[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + i * 256, row = c >> 3, col = (c & 7) * 8;
            const bf16_t* pa = a_row_ptr<bf16_t>(ap, m0 + row, M, k0);
            ra[i] = pa ? *(const uint4*)(pa + col) : make_uint4(0, 0, 0, 0);
            int n = n0 + row;
            rw[i] = (n < N) ? *(const uint4*)(W + (size_t)n * K + k0 + col) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + i * 256, row = c >> 3, col = (c & 7) * 8;
            *(uint4*)(sA + row * LDS_STRIDE + col) = ra[i];
            *(uint4*)(sW + row * LDS_STRIDE + col) = rw[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *(const bf16x8_t*)(sA + (wm * 64 + i * 16 + l15) * LDS_STRIDE + kk * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fw[j] = *(const bf16x8_t*)(sW + (wn * 64 + j * 16 + l15) * LDS_STRIDE + kk * 32 + g * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[j], fa[i], acc[i][j]);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue: acc[i][j][r] = C[m][n], m = m0 + wm*64 + i*16 + l15, n = n0 + wn*64 + j*16 + g*4 + r
    if (m0 + BM <= M && n0 + BN <= N && (EPI == EPI_HEADS || (ep.ldo & 3) == 0)) {   // interior tile (block-uniform)
        int rows[4], cols[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) cols[j] = n0 + wn * 64 + j * 16 + g * 4;
        epi_tile_interior<bf16_t, EPI>(ep, rows, cols, acc);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + wm * 64 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 64 + j * 16 + g * 4;
            if (n + 3 < N && (EPI == EPI_HEADS || (ep.ldo & 3) == 0)) {
                epi_store4<bf16_t, EPI>(ep, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[i][j][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA flavour of the tile GEMM: same 128x128x64 tiling and MFMA mapping, but the operand tiles go
// HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass), double
// buffered (2 x 32 KB).  The DMA destination is lane-linear (wave base + lane*16), so the bank-conflict
// fix is an XOR swizzle applied to the *source* address and mirrored on the fragment reads: 16-byte
// chunk c of row r lives at chunk position c ^ (r & 7).  Zero rows (M/N edges, conv padding, seek-window
// tail) are sourced from a zero page.
// ---------------------------------------------------------------------------------------------------
__device__ inline void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(AParams ap, const bf16_t* __restrict__ W, int M, int N,
                                                             int K, EpiParams ep, int tiles_n,
                                                             const bf16_t* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];   // [2][A 16 KB | W 16 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;

    // this lane stages rows r = ch*8 + (lane >> 3) of chunk ch = wave*4 + q, source chunk (lane & 7) ^ (r & 7)
    const int lrow = lane >> 3;
    const int csrc = ((lane & 7) ^ lrow) * 8;                       // r & 7 == lrow (ch*8 is a multiple of 8)
    // per-row address pieces are fixed across the K loop: hoist the (batch, time) decomposition of the conv gather
    size_t a_base[4]; int a_t[4], a_valid[4]; const bf16_t* w_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + lrow;
        const int m = m0 + r, n = n0 + r;
        w_row[q] = (n < N) ? W + (size_t)n * K + csrc : nullptr;
        a_valid[q] = -1;                                            // -1: zero row
        a_base[q] = 0; a_t[q] = 0;
        if (m < M) {
            if (ap.amode == 0) { a_base[q] = (size_t)m * ap.lda + csrc; a_valid[q] = 0x7fffffff; }
            else {
                const int b = m / ap.T_out, t = m - b * ap.T_out;
                a_t[q] = t * ap.stride - 1;                         // input row of tap 0 inside the window
                a_base[q] = (size_t)ap.row_off[b] * ap.C_in + csrc;
                a_valid[q] = ap.row_valid[b];
            }
        }
    }
    auto stage = [&](int k0, int buf) {
        unsigned char* base = gsm + buf * 32768;
        int tap = 0, c0 = k0;
        if (ap.amode != 0) { tap = k0 / ap.C_in; c0 = k0 - tap * ap.C_in; }   // k-tile uniform (scalar)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = wave * 4 + q;
            const bf16_t* pa = nullptr;
            if (ap.amode == 0) {
                if (a_valid[q] >= 0) pa = (const bf16_t*)ap.A + a_base[q] + k0;
            } else {
                const int t_in = a_t[q] + tap;
                if (t_in >= 0 && t_in < a_valid[q]) pa = (const bf16_t*)ap.A + a_base[q] + (size_t)t_in * ap.C_in + c0;
            }
            glds16(pa ? (const void*)pa : (const void*)zero_page, base + ch * 1024);
            glds16(w_row[q] ? (const void*)(w_row[q] + k0) : (const void*)zero_page, base + 16384 + ch * 1024);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage((kt + 1) * BK, cur ^ 1);
        const unsigned char* sA = gsm + cur * 32768;
        const unsigned char* sW = sA + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * 64 + i * 16 + l15;
                fa[i] = *(const bf16x8_t*)(sA + r * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wn * 64 + j * 16 + l15;
                fw[j] = *(const bf16x8_t*)(sW + r * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[j], fa[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (this wave's DMAs) ...
        __syncthreads();                                   // ... and everybody's; everyone done reading `cur`
    }

    if (m0 + BM <= M && n0 + BN <= N && (EPI == EPI_HEADS || (ep.ldo & 3) == 0)) {   // interior tile (block-uniform)
        int rows[4], cols[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) cols[j] = n0 + wn * 64 + j * 16 + g * 4;
        epi_tile_interior<bf16_t, EPI>(ep, rows, cols, acc);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + wm * 64 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 64 + j * 16 + g * 4;
            if (n + 3 < N && (EPI == EPI_HEADS || (ep.ldo & 3) == 0)) {
                epi_store4<bf16_t, EPI>(ep, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[i][j][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 256x256x64 flavour of the LDS-DMA GEMM for the large encoder shapes (M = batch*1500 rows): 8 waves as 2 (M) x 4 (N),
// each wave owns a 128x64 output patch (acc[8][4], 128 accumulator registers), two waves per SIMD.
// Why: with 128x128 tiles and 64x64 wave patches the kernel sits at BOTH per-CU limits at once -- fragment reads need
// 8 KB of LDS per 16 MFMAs (= 128 B/clk at full MFMA rate, the LDS limit) and the operand DMA needs 32 KB per K-tile
// (= 62 B/clk of the 64 B/clk L1 path).  256x256 / 128x64 brings that to ~94 B/clk and ~32 B/clk, and halves the
// L2/HBM re-fetch of the operand panels.  Same source-side XOR swizzle, staging map, raster and epilogues.
// LDS: 2 stages x (A 32 KB | W 32 KB) = 128 KB dynamic.
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_256_kernel(AParams ap, const bf16_t* __restrict__ W, int M, int N,
                                                            int K, EpiParams ep, int tiles_n,
                                                            const bf16_t* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm2[];   // [2][A 32 KB | W 32 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM2, n0 = nt_ * BN2;

    // staging map as in the 128 kernel: chunk ch = wave*4 + q (0..31) = rows ch*8 .. ch*8+7 of the 256-row tile
    const int lrow = lane >> 3;
    const int csrc = ((lane & 7) ^ lrow) * 8;
    size_t a_base[4]; int a_t[4], a_valid[4]; const bf16_t* w_row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + lrow;
        const int m = m0 + r, n = n0 + r;
        w_row[q] = (n < N) ? W + (size_t)n * K + csrc : nullptr;
        a_valid[q] = -1;
        a_base[q] = 0; a_t[q] = 0;
        if (m < M) {
            if (ap.amode == 0) { a_base[q] = (size_t)m * ap.lda + csrc; a_valid[q] = 0x7fffffff; }
            else {
                const int b = m / ap.T_out, t = m - b * ap.T_out;
                a_t[q] = t * ap.stride - 1;
                a_base[q] = (size_t)ap.row_off[b] * ap.C_in + csrc;
                a_valid[q] = ap.row_valid[b];
            }
        }
    }
    auto stage = [&](int k0, int buf) {
        unsigned char* base = gsm2 + buf * 65536;
        int tap = 0, c0 = k0;
        if (ap.amode != 0) { tap = k0 / ap.C_in; c0 = k0 - tap * ap.C_in; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = wave * 4 + q;
            const bf16_t* pa = nullptr;
            if (ap.amode == 0) {
                if (a_valid[q] >= 0) pa = (const bf16_t*)ap.A + a_base[q] + k0;
            } else {
                const int t_in = a_t[q] + tap;
                if (t_in >= 0 && t_in < a_valid[q]) pa = (const bf16_t*)ap.A + a_base[q] + (size_t)t_in * ap.C_in + c0;
            }
            glds16(pa ? (const void*)pa : (const void*)zero_page, base + ch * 1024);
            glds16(w_row[q] ? (const void*)(w_row[q] + k0) : (const void*)zero_page, base + 32768 + ch * 1024);
        }
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage((kt + 1) * BK, cur ^ 1);
        const unsigned char* sA = gsm2 + cur * 65536;
        const unsigned char* sW = sA + 32768;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[8], fw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wn * 64 + j * 16 + l15;
                fw[j] = *(const bf16x8_t*)(sW + r * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wm * 128 + i * 16 + l15;
                fa[i] = *(const bf16x8_t*)(sA + r * 128 + (((kk * 4 + g) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[j], fa[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const bool vec_ok = (EPI == EPI_HEADS || (ep.ldo & 3) == 0);
    if (m0 + BM2 <= M && n0 + BN2 <= N && vec_ok) {   // interior tile (block-uniform)
        int cols[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cols[j] = n0 + wn * 64 + j * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rows[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rows[i] = m0 + wm * 128 + (h * 4 + i) * 16 + l15;
            epi_tile_interior<bf16_t, EPI>(ep, rows, cols, *(const f32x4_t(*)[4][4])&acc[h * 4]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = m0 + wm * 128 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 64 + j * 16 + g * 4;
            if (n + 3 < N && vec_ok) {
                epi_store4<bf16_t, EPI>(ep, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[i][j][r]);
            }
        }
    }
}

#ifdef CW_EXPERIMENTS   // ping-pong schedule of the 256 tile: superseded by the 8-phase schedule (round 3), kept for A/B builds
// ---------------------------------------------------------------------------------------------------
// Ping-pong flavour of the 256x256x64 LDS-DMA GEMM (plain row-major A, N % 256 == 0): same tile, LDS image, swizzle,
// DMA map and epilogue as gemm_bf16_256_kernel, different time structure.  There all 8 waves move in lockstep through
// "DMA issue, fragment reads, 64 MFMAs, drain, barrier": the matrix pipes idle while the fragments come out of LDS and
// the LDS idles under the MFMAs (measured 41 % MFMA-busy in the steady state of fc2).  Here the two waves that share a
// SIMD belong to different groups (G0 = waves 0-3 = upper 128 rows, G1 = waves 4-7 = lower 128 rows) which run half a
// K-tile period apart: while one group issues its 64 MFMAs from registers, the other pulls the WHOLE next K-tile's
// fragments (24 x ds_read_b128 = 96 VGPRs) and issues its share of the LDS-DMA for the tile after.  Phases, separated by
// one s_barrier each (every wave executes the same number of barriers):
//      phase 2t   : G0 MFMA(t)                      | G1 READ(t), ISSUE(t+1), drain
//      phase 2t+1 : G0 READ(t+1), ISSUE(t+2)        | G1 MFMA(t)              (G0 drains at the end of its next MFMA phase)
// Stage (t+1)&1 is rewritten from phase 2t-1 on; its previous tile t-1 was last read in phases 2t-3 (G0) and 2t-2 (G1).
// Tile t+1 is complete when G0's pieces (issued in 2t-1) and G1's (issued in 2t) have been waited for by their issuers
// before the barrier that closes phase 2t; G0 reads it in 2t+1, G1 in 2t+2.  No LDS-DMA is in flight at any ds_read of
// the issuing wave (reads precede the issue inside a phase), so the compiler's conservative vmcnt(0) before LDS reads
// never fires.
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const bf16_t* __restrict__ A, int lda,
                                                           const bf16_t* __restrict__ W, int M, int N, int K,
                                                           EpiParams ep, int tiles_n, const bf16_t* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm3[];   // [2][A 32 KB | W 32 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // wm is also the group
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM2, n0 = nt_ * BN2;

    // DMA map as in the lockstep kernel: piece q of this wave = rows (wave*4 + q)*8 .. +7 of the A tile and of the W tile.
    // Issued as buffer_load_dwordx4 ... lds: SGPR descriptor + one 32-bit per-lane offset that never changes + a scalar
    // offset for the K position -- no 64-bit per-lane address arithmetic in the loop.  Rows beyond M are clamped to the
    // last row instead of zero-filled: they only feed output rows beyond M, which are never stored.
    const int lrow = lane >> 3;
    const int csrc = ((lane & 7) ^ lrow) * 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    int va[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) va[q] = (min(m0 + wave * 32 + q * 8 + lrow, M - 1) * lda + csrc) * 2;
    const int vw = ((n0 + wave * 32 + lrow) * K + csrc) * 2;
    auto issue = [&](int k0, int buf) {
        unsigned char* base = gsm3 + buf * 65536 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, va[q], k0 * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + 32768 + q * 1024), 16, vw,
                                                     (k0 + q * 8 * K) * 2, 0, 0);
        }
    };
    // fragment offsets inside a stage (chunk XOR by row & 7 == l15 & 7)
    const int sw = l15 & 7;
    const int aoff = (wm * 128 + l15) * 128, woff = 32768 + (wn * 64 + l15) * 128;
    bf16x8_t fa[2][8], fw[2][4];
    auto read_frags = [&](int buf) {
        const unsigned char* sb = gsm3 + buf * 65536;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = ((kk * 4 + g) ^ sw) << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) fw[kk][j] = *(const bf16x8_t*)(sb + woff + j * 2048 + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[kk][i] = *(const bf16x8_t*)(sb + aoff + i * 2048 + c);
        }
    };
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[kk][j], fa[kk][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nk = K / BK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 0) {   // loop rotated: a tile's fragments are read and consumed inside one iteration
        read_frags(0);
        if (nk > 1) issue(BK, 1);
        __builtin_amdgcn_s_barrier();                              // closes phase -1
        mfmas();                                                   // phase 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // own pieces of tile 1
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t + 1 < nk; ++t) {
            read_frags((t + 1) & 1);                               // phase 2t+1
            if (t + 2 < nk) issue((t + 2) * BK, t & 1);
            __builtin_amdgcn_s_barrier();
            mfmas();                                               // phase 2t+2: tile t+1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own pieces of tile t+2 (issued one phase ago)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                              // phase 2nk-1: G1's last MFMA phase
    } else {
        __builtin_amdgcn_s_barrier();                              // closes phase -1
        for (int t = 0; t < nk; ++t) {
            read_frags(t & 1);                                     // phase 2t
            if (t + 1 < nk) issue((t + 1) * BK, (t + 1) & 1);
            // own DMA pieces landed; own fragment reads returned -- G0 starts rewriting this stage right after the barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mfmas();                                               // phase 2t+1
            __builtin_amdgcn_s_barrier();
        }
    }

    const bool vec_ok = (EPI == EPI_HEADS || (ep.ldo & 3) == 0);
    if (m0 + BM2 <= M && vec_ok) {   // interior tile (block-uniform)
        int cols[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cols[j] = n0 + wn * 64 + j * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rows[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rows[i] = m0 + wm * 128 + (h * 4 + i) * 16 + l15;
            epi_tile_interior<bf16_t, EPI>(ep, rows, cols, *(const f32x4_t(*)[4][4])&acc[h * 4]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = m0 + wm * 128 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 64 + j * 16 + g * 4;
            if (vec_ok) {
                epi_store4<bf16_t, EPI>(ep, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[i][j][r]);
            }
        }
    }
}

#endif

// ---------------------------------------------------------------------------------------------------
// Quarter-tile ("8-phase") flavour of the same 256x256x64 LDS-DMA GEMM (round 3).  The ping-pong kernel above ends every
// phase with `s_waitcnt vmcnt(0)`: the DMA pieces a group issued in a phase must have landed before that phase's barrier, so
// a K-tile costs two memory round trips however few cycles its 64 MFMAs take (MFMA pipe busy 0.33-0.35).  Here
//   * the operand tiles are handled as four HALF tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255 of the K-tile, 16 KB
//     each) and a wave's 128 x 64 output is two 64-row x two 32-column pieces, one from each half, so that its four
//     quadrants touch the half tiles at different times:
//         phase 1: read A_h0 (8 ds_read_b128) + W_h0 (4)   -> 16 MFMAs, quadrant (0,0)
//         phase 2: read W_h1 (4)                            -> 16 MFMAs, quadrant (0,1)
//         phase 3: read A_h1 (8)                            -> 16 MFMAs, quadrant (1,1)
//         phase 4: (W_h0 fragments kept from phase 1)       -> 16 MFMAs, quadrant (1,0)
//   * every phase also issues the DMA of ONE half tile (2 x buffer_load ... lds per lane): W_h0, W_h1, A_h1 of the next
//     K-tile in phases 1-3 and A_h0 of the tile after next in phase 4 -- each into a half-tile slot whose last ds_read was
//     issued at least three phases earlier;
//   * the only wait on DMA is a COUNTED one, `s_waitcnt vmcnt(6)`: three half tiles stay in flight, a half tile is read
//     four phases (five for A_h0) after it was requested and never in the phase whose wait retired it;
//   * the two wave rows (wr = wave >> 2, which share the SIMDs pairwise) run one barrier apart: one does its 16 MFMAs while
//     the other reads fragments and issues DMA.
// Same LDS image, swizzle, epilogue and per-element summation order as the other 256-tile kernels (bit-identical results).
// ---------------------------------------------------------------------------------------------------
#ifdef CW_EXPERIMENTS
__constant__ int c_gemm_gm = 8;
#endif

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_8ph_kernel(const bf16_t* __restrict__ A, int lda,
                                                            const bf16_t* __restrict__ W, int M, int N, int K,
                                                            EpiParams ep, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm4[];   // [2][A 32 KB | W 32 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
#ifdef CW_EXPERIMENTS
    // tile-order A/B (option "gemm_gm"): g > 0 m-tiles per group; g < 0 the same groups of -g WITHOUT the XCD remap
    // (consecutive tiles dealt round-robin over the eight L2s).  profiles/r04_gemm_tile_order_ab.txt
    const int gm_ = c_gemm_gm;
    const int tile = gm_ < 0 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_, gm_ < 0 ? -gm_ : gm_);
#else
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
#endif
    const int m0 = mt_ * BM2, n0 = nt_ * BN2;

    // DMA map of a half tile (128 rows x 128 B): wave w, load q: rows w*16 + q*8 .. +7, lane -> (row lane >> 3, 16 B chunk
    // (lane & 7) ^ row): 1 KB per wave instruction, chunk-swizzled on the SOURCE side so the LDS image is lane-linear
    const int lrow = lane >> 3;
    const int csrc = ((lane & 7) ^ lrow) * 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    int va[2][2];                                                // rows beyond M clamped: they feed output rows that are never stored
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) va[h][q] = (min(m0 + h * 128 + wave * 16 + q * 8 + lrow, M - 1) * lda + csrc) * 2;
    const int vw = ((n0 + wave * 16 + lrow) * K + csrc) * 2;
    auto issue_a = [&](int k0, int buf, int h) {
#ifdef CW_8PH_NO_DMA
        return;
#endif
        unsigned char* base = gsm4 + buf * 65536 + (h * 128 + wave * 16) * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, va[h][q], k0 * 2, 0, 0);
    };
    auto issue_w = [&](int k0, int buf, int h) {
#ifdef CW_8PH_NO_DMA
        return;
#endif
        unsigned char* base = gsm4 + buf * 65536 + 32768 + (h * 128 + wave * 16) * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, vw,
                                                     (k0 + (h * 128 + q * 8) * K) * 2, 0, 0);
    };
    // fragment offsets (chunk XOR by row & 7 == l15 & 7): A half h: rows h*128 + wr*64 + i*16 + l15; W half h: rows h*128 + wn*32 + j*16 + l15
    const int sw = l15 & 7;
    const int aoff = (wr * 64 + l15) * 128, woff = 32768 + (wn * 32 + l15) * 128;
    bf16x8_t fa[2][4], fw[2][2][2];                             // fw[column half]: the W_h0 fragments serve phases 1 and 4
    auto read_a = [&](int buf, int h) {
#ifdef CW_8PH_NO_READS
        if (buf >= 0) return;
#endif
        const unsigned char* sb = gsm4 + buf * 65536 + aoff + h * 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = ((kk * 4 + g) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = *(const bf16x8_t*)(sb + i * 2048 + c);
        }
    };
    auto read_w = [&](int buf, int h) {
#ifdef CW_8PH_NO_READS
        if (buf >= 0) return;
#endif
        const unsigned char* sb = gsm4 + buf * 65536 + woff + h * 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = ((kk * 4 + g) ^ sw) << 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[h][kk][j] = *(const bf16x8_t*)(sb + j * 2048 + c);
        }
    };
    f32x4_t acc[2][4][4];                                        // [row half][i][column half * 2 + j]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // second half of a phase: the reads above have returned, 16 MFMAs on quadrant (hm, hn), closing barrier
#define CW_8PH_COMPUTE(hm, hn)                                                                              \
    do {                                                                                                    \
        __builtin_amdgcn_s_barrier();                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
                    acc[hm][i][(hn) * 2 + j] = mfma16(fw[hn][kk][j], fa[kk][i], acc[hm][i][(hn) * 2 + j]); \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
    } while (0)
    // counted wait on the LDS-DMA: three half tiles (6 loads) may stay in flight; once fewer are being issued (the last two
    // K-tiles) everything is drained instead
#ifndef CW_8PH_VMCNT
#define CW_8PH_VMCNT 6
#endif
#define CW_8PH_STR2(x) #x
#define CW_8PH_STR(x) CW_8PH_STR2(x)
#define CW_8PH_WAIT(steady)                                                                                 \
    do {                                                                                                    \
        if (steady) asm volatile("s_waitcnt vmcnt(" CW_8PH_STR(CW_8PH_VMCNT) ")" ::: "memory");             \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)

    const int nk = K / BK;
    // prologue: K-tile 0 complete, A_h0 of K-tile 1
    issue_a(0, 0, 0); issue_w(0, 0, 0); issue_w(0, 0, 1); issue_a(0, 0, 1);
    if (nk > 1) issue_a(BK, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                   // the lower wave row runs one barrier behind
    for (int t = 0; t < nk; ++t) {
        const int b = t & 1, nb = b ^ 1;
        const bool nxt = t + 1 < nk, nxt2 = t + 2 < nk;
        const bool steady = t + 2 < nk;                          // all four phases of this tile issue a half tile
        const int k1 = (t + 1) * BK, k2 = (t + 2) * BK;
        // phase 1
        read_w(b, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(b, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nxt) issue_w(k1, nb, 0);
        CW_8PH_WAIT(steady);
        CW_8PH_COMPUTE(0, 0);
        // phase 2
        read_w(b, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (nxt) issue_w(k1, nb, 1);
        CW_8PH_WAIT(steady);
        CW_8PH_COMPUTE(0, 1);
        // phase 3
        read_a(b, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (nxt) issue_a(k1, nb, 1);
        CW_8PH_WAIT(steady);
        CW_8PH_COMPUTE(1, 1);
        // phase 4
        if (nxt2) issue_a(k2, b, 0);
        CW_8PH_WAIT(steady);
        CW_8PH_COMPUTE(1, 0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                   // every wave executes the same number of barriers
#undef CW_8PH_COMPUTE
#undef CW_8PH_WAIT

    const bool vec_ok = (EPI == EPI_HEADS || (ep.ldo & 3) == 0);
    int cols[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cols[j] = n0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + g * 4;
    if (m0 + BM2 <= M && vec_ok) {   // interior tile (block-uniform)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rows[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rows[i] = m0 + h * 128 + wr * 64 + i * 16 + l15;
            epi_tile_interior<bf16_t, EPI>(ep, rows, cols, acc[h]);
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + h * 128 + wr * 64 + i * 16 + l15;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = cols[j];
                if (vec_ok) {
                    epi_store4<bf16_t, EPI>(ep, m, n, acc[h][i][j][0], acc[h][i][j][1], acc[h][i][j][2], acc[h][i][j][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[h][i][j][r]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// OCP e4m3 flavour of the ping-pong GEMM (opt-in encoder mode, BASELINE configs[3]): C[M,N] = (A8 * W8^T) * sa[m] * sw[n].
// A and W hold one byte per element with one f32 scale per row (quant_rows_fp8_kernel / layernorm_fp8_kernel); the MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (e8m0 127) -- on gfx950 the only fp8 form that runs at twice the
// bf16 rate.  A 256 x 128-byte operand tile is byte-identical in shape to the bf16 kernel's 256 x 64-element tile, so LDS
// image, swizzle, DMA map and schedule are the ping-pong kernel's; a lane's 32 operand bytes are the two 16-byte fragments
// the bf16 kernel feeds to two MFMAs (chunk g and chunk 4 + g of the row: k = g*16.. and 64 + g*16.., which is also the
// instruction's own k order -- tests/native/mfma_fp8_probe.hip), so one K-tile is 32 MFMAs of K = 128 instead of 64 of K = 32.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
// one operand fragment = 32 bytes of a row: 16-byte chunks g and 4 + g, read straight into the halves of one 8-register tuple
__device__ inline i32x8_t frag_fp8(const unsigned char* row, int c_lo, int c_hi) {
    const u32x4_t lo = *(const u32x4_t*)(row + c_lo), hi = *(const u32x4_t*)(row + c_hi);
    return (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_pp_kernel(const unsigned char* __restrict__ A, int lda,
                                                          const unsigned char* __restrict__ W, int M, int N, int K,
                                                          const float* __restrict__ sa, const float* __restrict__ sw,
                                                          EpiParams ep, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm4[];   // [2][A 32 KB | W 32 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // wm is also the group
    const int l15 = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int mt_, nt_;
    grouped_tile(tile, gridDim.x / tiles_n, tiles_n, mt_, nt_);
    const int m0 = mt_ * BM2, n0 = nt_ * BN2;

    const int lrow = lane >> 3;
    const int csrc = ((lane & 7) ^ lrow) * 16;                 // bytes
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    int va[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) va[q] = min(m0 + wave * 32 + q * 8 + lrow, M - 1) * lda + csrc;
    const int vw = (n0 + wave * 32 + lrow) * K + csrc;
    auto issue = [&](int k0, int buf) {
        unsigned char* base = gsm4 + buf * 65536 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, va[q], k0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + 32768 + q * 1024), 16, vw,
                                                     k0 + q * 8 * K, 0, 0);
        }
    };
    const int sw_ = l15 & 7;
    const int aoff = (wm * 128 + l15) * 128, woff = 32768 + (wn * 64 + l15) * 128;
    i32x8_t fa[8], fw[4];
    const int c_lo = (g ^ sw_) << 4, c_hi = ((4 + g) ^ sw_) << 4;
    auto read_frags = [&](int buf) {
        const unsigned char* sb = gsm4 + buf * 65536;
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = frag_fp8(sb + woff + j * 2048, c_lo, c_hi);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = frag_fp8(sb + aoff + i * 2048, c_lo, c_hi);
    };
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[j], fa[i], acc[i][j], 0, 0, 0, 127, 0, 127);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nk = K / 128;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 0) {   // same phases as gemm_bf16_pp_kernel; the loop is rotated so that a tile's fragments are read and consumed
                     // inside one iteration (carried across the back edge they cost ~150 spilled registers here)
        read_frags(0);
        if (nk > 1) issue(128, 1);
        __builtin_amdgcn_s_barrier();                              // closes phase -1
        mfmas();                                                   // phase 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t + 1 < nk; ++t) {
            read_frags((t + 1) & 1);                               // phase 2t+1
            if (t + 2 < nk) issue((t + 2) * 128, t & 1);
            __builtin_amdgcn_s_barrier();
            mfmas();                                               // phase 2t+2: tile t+1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                              // phase 2nk-1 (G1's last MFMA phase)
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            read_frags(t & 1);
            if (t + 1 < nk) issue((t + 1) * 128, (t + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mfmas();
            __builtin_amdgcn_s_barrier();
        }
    }

    // de-quantise: row scale x column scale, then the usual epilogues
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 cs = *(const float4*)(sw + n0 + wn * 64 + j * 16 + g * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float rs = sa[min(m0 + wm * 128 + i * 16 + l15, M - 1)];
            acc[i][j][0] *= rs * cs.x; acc[i][j][1] *= rs * cs.y; acc[i][j][2] *= rs * cs.z; acc[i][j][3] *= rs * cs.w;
        }
    }
    const bool vec_ok = (EPI == EPI_HEADS || (ep.ldo & 3) == 0);
    if (m0 + BM2 <= M && vec_ok) {   // interior tile (block-uniform)
        int cols[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cols[j] = n0 + wn * 64 + j * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rows[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rows[i] = m0 + wm * 128 + (h * 4 + i) * 16 + l15;
            epi_tile_interior<bf16_t, EPI>(ep, rows, cols, *(const f32x4_t(*)[4][4])&acc[h * 4]);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int m = m0 + wm * 128 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + wn * 64 + j * 16 + g * 4;
            if (vec_ok) {
                epi_store4<bf16_t, EPI>(ep, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) epi_store1<bf16_t, EPI>(ep, m, n + r, acc[i][j][r]);
            }
        }
    }
}

// Row-wise e4m3 quantisation of a 16-bit matrix (weights at option time; activations that no fused producer quantises):
// out8[r][k] = e4m3(x[r][k] / s_r), s_r = max_k |x[r][k]| / 448.  One wave per row, K % 8 == 0.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int rows, int K,
                                                             unsigned char* __restrict__ out, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * K;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        float v[8];
        h16_unpack8(*(const uint4*)(xr + k), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    for (int k = lane * 8; k < K; k += 512) {
        float v[8];
        h16_unpack8(*(const uint4*)(xr + k), v);
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        *(uint2*)(out + (size_t)row * K + k) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// ---------------------------------------------------------------------------------------------------
// f32 parity GEMM: 64x64 tile, 256 threads, 4x4 outputs per thread, BK = 16.
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(AParams ap, const float* __restrict__ W, int M, int N, int K,
                                                       EpiParams ep, int tiles_n) {
    __shared__ float sA[16][65];
    __shared__ float sW[16][65];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    const int tm = tid >> 4, tn = tid & 15;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        // each thread loads 4 A and 4 W elements: row = tid / 4, cols (tid % 4) * 4 .. +3
        int row = tid >> 2, kc = (tid & 3) * 4;
        const float* pa = a_row_ptr<float>(ap, m0 + row, M, (k0 / 64) * 64);
        int koff = k0 - (k0 / 64) * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sA[kc + c][row] = pa ? pa[koff + kc + c] : 0.f;
            int n = n0 + row;
            sW[kc + c][row] = (n < N) ? W[(size_t)n * K + k0 + kc + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k][tm * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = sW[k][tn * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + tm * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + tn * 4 + j;
            if (n < N) epi_store1<float, EPI>(ep, m, n, acc[i][j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Decode GEMV (skinny GEMM), bf16 weights.  One block = 16 output columns; its 4 waves split the K
// steps (128 k per step: each lane streams 64 contiguous bytes of its weight row) and reduce through
// LDS.  x is f32 [Mb][K] in global (decode activations are tiny and kept in f32), converted to bf16
// while being staged into LDS, with an optional LayerNorm prologue over the full row (K == d_model).
// ---------------------------------------------------------------------------------------------------
#define GV_MAXM 64

template <int EPI>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const float* __restrict__ x, int Mb, int K, int KC,
                                                        const bf16_t* __restrict__ W, int N,
                                                        const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Mpad = (Mb + 15) & ~15;
    const int xs_stride = KC + 8;                                  // bf16 elements
    bf16_t* xs = (bf16_t*)smem;                                     // [Mpad][KC+8]
    float* stats = (float*)(smem + (size_t)Mpad * xs_stride * 2);   // [Mpad][2] mean, rstd
    float* red = stats + 2 * GV_MAXM;                               // [4 waves][4 mtiles][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int mt_n = Mpad >> 4;

    if (ln_g) {  // LayerNorm statistics, two-pass, one wave per row
        for (int m = wave; m < Mb; m += 4) {
            const float* xr = x + (size_t)m * K;
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += xr[k];
            float mean = wave_sum(s) / (float)K;
            float v = 0.f;
            for (int k = lane; k < K; k += 64) { float d = xr[k] - mean; v += d * d; }
            float var = wave_sum(v) / (float)K;
            if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = 1.0f / sqrtf(var + 1e-5f); }
        }
    }
    __syncthreads();

    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int n = n0 + l15;
    const bf16_t* wrow = W + (size_t)(n < N ? n : 0) * K;
    const int steps_per_chunk = KC / 128;

    for (int kc0 = 0; kc0 < K; kc0 += KC) {
        // stage x[:, kc0 : kc0+KC] -> LDS bf16 (rows >= Mb are zero)
        for (int idx = tid; idx < Mpad * (KC / 4); idx += 256) {
            int m = idx / (KC / 4), k4 = (idx - m * (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < Mb) {
                v = *(const float4*)(x + (size_t)m * K + kc0 + k4);
                if (ln_g) {
                    float mean = stats[2 * m], rstd = stats[2 * m + 1];
                    if (ln_b) {
                        const float4 gg = *(const float4*)(ln_g + kc0 + k4);
                        const float4 bb = *(const float4*)(ln_b + kc0 + k4);
                        v.x = (v.x - mean) * rstd * gg.x + bb.x;
                        v.y = (v.y - mean) * rstd * gg.y + bb.y;
                        v.z = (v.z - mean) * rstd * gg.z + bb.z;
                        v.w = (v.w - mean) * rstd * gg.w + bb.w;
                    } else {   // affine part folded into the weights (engine.hip: fold_layernorm)
                        v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
                    }
                }
            }
            ushort4 o;
            o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
            *(ushort4*)(xs + (size_t)m * xs_stride + k4) = o;
        }
        __syncthreads();
        for (int step = wave; step < steps_per_chunk; step += 4) {
            const int kl = step * 128 + g * 32;                     // local k of this lane's 32 weights
            u32x4_t w4[4];
            if (n < N) {
                const u32x4_t* wp = (const u32x4_t*)(wrow + kc0 + kl);
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = __builtin_nontemporal_load(wp + j);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = (u32x4_t){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < mt_n) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x8_t a = *(const bf16x8_t*)(xs + (size_t)(t * 16 + l15) * xs_stride + kl + j * 8);
                        acc[t] = mfma16(a, __builtin_bit_cast(bf16x8_t, w4[j]), acc[t]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // cross-wave reduction: D[row = batch m = g*4 + r][col = n = l15]
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < mt_n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * 4 + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < mt_n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[((w * 4 + t) * 4 + r) * 64 + lane];
                    int m = t * 16 + g * 4 + r;
                    if (m < Mb && n < N) {
                        if (EPI == EPI_RESID_F32) {      // keep the residual stream on the 2^-12 grid (see resid_grid)
                            EpiParams e2 = ep; e2.bias = nullptr;
                            epi_store1<bf16_t, EPI>(e2, m, n, resid_grid(v + (ep.bias ? ep.bias[n] : 0.f)));
                        } else epi_store1<bf16_t, EPI>(ep, m, n, v);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Decode GEMV v2 (Mb <= 16, K slice per block <= 1280).  Latency-oriented rewrite of the kernel above:
//   * every global load is UNCONDITIONAL (indices are clamped to valid memory instead of predicated):
//     hipcc otherwise wraps each load in its own exec-masked block and serialises them with vmcnt waits;
//     garbage in unused MFMA rows/columns is harmless because it never reaches a stored output;
//   * issue order: bias, activation rows, LayerNorm parameters, then the weight stream -- vmcnt retires
//     in order, so the LayerNorm math overlaps the weight fetch;
//   * a row of activations lives in one wave (rows wave, wave+4, ...): LayerNorm statistics are wave-local
//     shuffles, no block barrier; rows are rounded to bf16 and parked in LDS;
//   * one barrier, MFMA 16x16x32 over the wave's K steps, cross-wave reduction through LDS, epilogue
//     spread over all 256 threads.
// grid = (ceil(N/16), KSPLIT).  KSPLIT > 1 is only used with the in-place residual epilogue, where the
// partial sums are accumulated with f32 atomics straight into the residual stream.
// NSLOT = ceil(steps / 4) weight steps per wave, PER_LANE = ceil(Kb / 256) float4 per lane per row.
// ---------------------------------------------------------------------------------------------------
// Development aid (make EXTRA=-DCW_PHASE_TIMING, tools/phase_probe.py): thread 0 of the first 512 blocks stamps the
// 100 MHz wall clock at the phase boundaries of the decode GEMV; cw_debug_phases copies the stamps out.  This is how the
// ds_bpermute-based LayerNorm reductions were found on the critical path (1.5 us of a 7.9 us kernel).
#if defined(CW_PHASE_TIMING) && !defined(CW_F16)
__device__ unsigned long long g_phase[512 * 8];
#define PH(i)                                                                                          \
    do {                                                                                               \
        const int bid_ = blockIdx.x + blockIdx.y * gridDim.x;                                          \
        if (threadIdx.x == 0 && bid_ < 512) g_phase[bid_ * 8 + (i)] = wall_clock64();                  \
    } while (0)
extern "C" int cw_debug_phases(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 512 * 8);
}
#else
#define PH(i) do { } while (0)
#endif

template <int EPI, int RPW, bool ATOMIC, bool COMBINE, int NSLOT, int PER_LANE, int NT = 1 /* 16-column tiles per block */,
          bool X16 = false /* activation rows arrive in the engine's 16-bit type (no LayerNorm, no combine) */>
__global__ __launch_bounds__(256) void gemv2_bf16_kernel(const float* __restrict__ x, int Mb, int K, int Kb,
                                                         const bf16_t* __restrict__ W, int N,
                                                         const float* __restrict__ ln_g,
                                                         const float* __restrict__ ln_b, EpiParams ep,
                                                         CombineParams cb, int m_base, int wpk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    if (gridDim.z > 1) {   // row groups inside one launch (the combining out-projection): block z takes rows z * rpb .. of the launch's Mb
        const int rpb = (Mb + (int)gridDim.z - 1) / (int)gridDim.z;
        const int mb0 = (int)blockIdx.z * rpb;
        m_base += mb0;
        Mb = min(rpb, Mb - mb0);
        if (Mb <= 0) return;
    }
    x += (size_t)m_base * K;                                  // this launch handles batch rows m_base .. m_base+Mb-1
    const int xs_stride = Kb + 8;
    bf16_t* xs = (bf16_t*)smem2;                              // [16][Kb+8]
    float* red = (float*)(smem2 + (size_t)16 * xs_stride * 2); // [4 waves][NT][4][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int kbase = blockIdx.y * Kb;
    const int steps = Kb >> 7;
    int nn[NT], ncl[NT];                                      // this lane's column in each tile, clamped (n >= N is dropped)
#pragma unroll
    for (int t = 0; t < NT; ++t) { nn[t] = n0 + t * 16 + l15; ncl[t] = nn[t] < N ? nn[t] : N - 1; }
    const int nvec = Kb >> 2;                                 // float4 per row slice
    const bool has_ln = ln_g != nullptr;
    const bool ln_affine = ln_b != nullptr;                   // false: gamma / beta were folded into W / bias at load time

    PH(0);                                                    // kernel entry
    float bias_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_v[t] = ep.bias ? ep.bias[ncl[t]] : 0.f;
    // EPI_QKV_CACHE: the cache row of the batch row this thread stores (m = g * 4 + wave), requested at the head of the queue
    int rpos_pre = 0;
    if (EPI == EPI_QKV_CACHE) rpos_pre = ep.row_pos[m_base + min(g * 4 + wave, Mb - 1)];

    // activation rows -> registers
    constexpr int PER8 = (PER_LANE + 1) / 2;                  // X16: 8 elements per 16-byte load
    u32x4_t xh[RPW][PER8];
    float4 xv[RPW][PER_LANE];
    if (X16) {
        const bf16_t* xb = (const bf16_t*)x;                  // (x was advanced by m_base * K floats: undo in 16-bit units below)
        xb = xb - (size_t)m_base * K * 2 + (size_t)m_base * K;
        const int nv8 = Kb >> 3;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            int row = wave + 4 * i;
            row = row < Mb ? row : Mb - 1;
#pragma unroll
            for (int c = 0; c < PER8; ++c) {
                int v8 = lane + 64 * c;
                v8 = v8 < nv8 ? v8 : nv8 - 1;
                xh[i][c] = *(const u32x4_t*)(xb + (size_t)row * K + kbase + v8 * 8);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < (X16 ? 0 : RPW); ++i) {
        int row = wave + 4 * i;
        row = row < Mb ? row : Mb - 1;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;
            const size_t off = (size_t)row * K + kbase + v4 * 4;
            if (!COMBINE) {
                xv[i][c] = *(const float4*)(x + off);
            } else {   // x = sum_s w_s o_s over the ATT_NS attention partials of head (k / 64)
                const int head = (kbase + v4 * 4) >> 6;
                const float* ml = cb.part_ml + ((size_t)(m_base + row) * cb.H + head) * ATT_NS * 2;
                float m[ATT_NS], l[ATT_NS];
                float4 o[ATT_NS];
#pragma unroll
                for (int sI = 0; sI < ATT_NS; ++sI) {
                    const float2 t = *(const float2*)(ml + 2 * sI);
                    m[sI] = t.x; l[sI] = t.y;
                    o[sI] = *(const float4*)(x + (size_t)sI * cb.plane + off);
                }
                float M = m[0];
#pragma unroll
                for (int sI = 1; sI < ATT_NS; ++sI) M = fmaxf(M, m[sI]);
                float L = 0.f;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sI = 0; sI < ATT_NS; ++sI) {
                    const float w = __expf(m[sI] - M);
                    L += l[sI] * w;
                    r.x += w * o[sI].x; r.y += w * o[sI].y; r.z += w * o[sI].z; r.w += w * o[sI].w;
                }
                const float inv = 1.0f / L;
                r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
                xv[i][c] = r;
            }
        }
    }
    // LayerNorm parameters (a valid dummy pointer is passed when there is no LayerNorm; loads are cheap)
    float4 gv[PER_LANE], bv[PER_LANE];
    if (has_ln && ln_affine) {   // kernel-uniform: the whole block of loads is either issued or not
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;
            gv[c] = *(const float4*)(ln_g + v4 * 4);
            bv[c] = *(const float4*)(ln_b + v4 * 4);
        }
    }
    // weight stream: steps wave, wave+4, wave+8 (clamped: a tail wave re-reads a step another wave owns)
    u32x4_t wq[NT][NSLOT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16_t* wrow = W + (size_t)ncl[t] * K + kbase + g * 8;   // MFMA j covers k = j*32 + g*8 .. +7
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int step = wave + 4 * s;
            step = step < steps ? step : steps - 1;
            // wpk: fragment-major weights (wfrag_pack_kernel): the 64 lanes' 16-byte fragments of one MFMA operand are 1 KB of
            // contiguous memory, i.e. 8 full cache lines per wave instruction instead of 16 half lines at a row stride
            const u32x4_t* wp = wpk ? (const u32x4_t*)(W + ((((size_t)(ncl[t] >> 4) * (K >> 5)) + (kbase >> 5) + step * 4) * 64 + g * 16 + (ncl[t] & 15)) * 8)
                                    : (const u32x4_t*)(wrow + step * 128);
            const int sj = wpk ? 64 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[t][s][j] = wp[j * sj];   // (non-temporal loads measured slower here: decode 436 vs 418 ms / step)
        }
    }

    PH(1);                                                    // every load accepted by the memory pipeline
    if (has_ln) {   // only launched with Kb == K
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < PER_LANE; ++c) {
                const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
                s += ok * ((xv[i][c].x + xv[i][c].y) + (xv[i][c].z + xv[i][c].w));
            }
            const float mean = wave_sum(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < PER_LANE; ++c) {
                const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
                float a = xv[i][c].x - mean, b = xv[i][c].y - mean, cc = xv[i][c].z - mean, d = xv[i][c].w - mean;
                q += ok * cw_sumsq4(a, b, cc, d);
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
            if (ln_affine) {
#pragma unroll
                for (int c = 0; c < PER_LANE; ++c) {
                    xv[i][c].x = (xv[i][c].x - mean) * rstd * gv[c].x + bv[c].x;
                    xv[i][c].y = (xv[i][c].y - mean) * rstd * gv[c].y + bv[c].y;
                    xv[i][c].z = (xv[i][c].z - mean) * rstd * gv[c].z + bv[c].z;
                    xv[i][c].w = (xv[i][c].w - mean) * rstd * gv[c].w + bv[c].w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < PER_LANE; ++c) {
                    xv[i][c].x = (xv[i][c].x - mean) * rstd; xv[i][c].y = (xv[i][c].y - mean) * rstd;
                    xv[i][c].z = (xv[i][c].z - mean) * rstd; xv[i][c].w = (xv[i][c].w - mean) * rstd;
                }
            }
        }
    }
    // rows -> bf16 -> LDS.  Rows >= Mb are never written: they only feed MFMA output rows that are dropped.
    if (X16) {
        const int nv8 = Kb >> 3;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 4 * i;
#pragma unroll
            for (int c = 0; c < PER8; ++c) {
                int v8 = lane + 64 * c;
                v8 = v8 < nv8 ? v8 : nv8 - 1;                   // clamped lanes rewrite identical data
                *(u32x4_t*)(xs + (size_t)row * xs_stride + v8 * 8) = xh[i][c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < (X16 ? 0 : RPW); ++i) {
        const int row = wave + 4 * i;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;                    // clamped lanes rewrite identical data
            const float4 v = xv[i][c];
            ushort4 o;
            o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
            *(ushort4*)(xs + (size_t)row * xs_stride + v4 * 4) = o;
        }
    }
    PH(2);                                                    // activations (LayerNorm) parked in LDS
    __syncthreads();
    PH(3);

    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step = wave + 4 * s;
        if (step < steps) {                                     // wave-uniform (scalar) branch, no loads inside
            const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t a = *(const bf16x8_t*)(xr + j * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, __builtin_bit_cast(bf16x8_t, wq[t][s][j]), acc[t]);
            }
        }
    }
    // D[row = batch g*4 + r][col = l15]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    PH(4);                                                    // weights arrived, MFMAs done
    __syncthreads();
    PH(5);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid >> 6;
        float v = red[((0 * NT + t) * 4 + r) * 64 + lane] + red[((1 * NT + t) * 4 + r) * 64 + lane] +
                  red[((2 * NT + t) * 4 + r) * 64 + lane] + red[((3 * NT + t) * 4 + r) * 64 + lane];
        const int m = g * 4 + r, n = nn[t];
        if (m < Mb && n < N) {
            if (ATOMIC) {
                atomicAdd(ep.outf + (size_t)(m_base + m) * ep.ldo + n, resid_grid(v + (blockIdx.y == 0 ? bias_v[t] : 0.f)));
            } else {
                EpiParams e2 = ep;
                e2.bias = nullptr;                       // bias was prefetched at kernel entry
                if (EPI == EPI_QKV_CACHE) { e2.row_pos = nullptr; e2.row_pos_pre = rpos_pre; }   // ... and so was the cache row
                epi_store1<bf16_t, EPI>(e2, m_base + m, n, EPI == EPI_RESID_F32 ? resid_grid(v + bias_v[t]) : v + bias_v[t]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Very wide LayerNorm GEMV (the 51866-column logits projection, Mb <= 16, K <= 1280, fragment-major weights): the same
// arithmetic as gemv2_bf16_kernel in a PERSISTENT column loop.  3242 column tiles as 1081 blocks of three are 2.1 rounds of
// resident blocks; every block redoes the LayerNorm of the rows and every round pays its own ramp (33 us for 133 MB = 4.0 TB/s).
// Here the grid is sized to be resident at once (<= 3 blocks per CU), a block normalises the rows ONCE, then walks the tiles
// b, b + G, b + 2G, ... with the next tile's weights requested before the current tile's MFMAs: one continuous stream per
// block.  Per tile: the wave's K steps on the matrix cores, cross-wave sum through one of two alternating LDS buffers (one
// barrier per tile), store.  Per-element summation order = gemv2's (wave partials over steps w, w + 4, w + 8, then waves
// 0..3): bit-identical logits.
// ---------------------------------------------------------------------------------------------------
template <int RPW, int NSLOT, int PER_LANE>
__global__ __launch_bounds__(256) void gemv_loop_kernel(const float* __restrict__ x, int Mb, int K, const bf16_t* __restrict__ W, int N,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int xs_stride = K + 8;
    bf16_t* xs = (bf16_t*)smem3;                               // [16][K+8]
    float* red = (float*)(smem3 + (size_t)16 * xs_stride * 2);  // [2][4 waves][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int steps = K >> 7, nvec = K >> 2, KS = K >> 5;
    const int ntiles = (N + 15) >> 4;
    const bool ln_affine = ln_b != nullptr;

    // first tile's weights and the rows: every request out before the LayerNorm arithmetic
    u32x4_t wq[NSLOT][4], wn[NSLOT][4];
    auto load_tile = [&](int tile, u32x4_t (&w)[NSLOT][4]) {
        const u32x4_t* base = (const u32x4_t*)W + (size_t)tile * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int step = wave + 4 * s;
            step = step < steps ? step : steps - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[s][j] = CW_STREAM_LD(base + (size_t)(step * 4 + j) * 64);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile, wq);
    float4 xv[RPW][PER_LANE], gv[PER_LANE], bv[PER_LANE];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int row = wave + 4 * i;
        row = row < Mb ? row : Mb - 1;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;
            xv[i][c] = *(const float4*)(x + (size_t)row * K + v4 * 4);
        }
    }
    if (ln_affine) {
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;
            gv[c] = *(const float4*)(ln_g + v4 * 4);
            bv[c] = *(const float4*)(ln_b + v4 * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            sx += ok * ((xv[i][c].x + xv[i][c].y) + (xv[i][c].z + xv[i][c].w));
        }
        const float mean = wave_sum(sx) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            float a = xv[i][c].x - mean, b = xv[i][c].y - mean, cc = xv[i][c].z - mean, d = xv[i][c].w - mean;
            q += ok * cw_sumsq4(a, b, cc, d);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
        const int row = wave + 4 * i;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            float4 v = xv[i][c];
            if (ln_affine) {
                v.x = (v.x - mean) * rstd * gv[c].x + bv[c].x; v.y = (v.y - mean) * rstd * gv[c].y + bv[c].y;
                v.z = (v.z - mean) * rstd * gv[c].z + bv[c].z; v.w = (v.w - mean) * rstd * gv[c].w + bv[c].w;
            } else {
                v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
            }
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;                    // clamped lanes rewrite identical data
            ushort4 o;
            o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
            *(ushort4*)(xs + (size_t)row * xs_stride + v4 * 4) = o;
        }
    }
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) load_tile(nxt, wn);                   // the next tile's stream runs under this tile's MFMAs and exchange
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int step = wave + 4 * s;
            if (step < steps) {
                const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = mfma16(*(const bf16x8_t*)(xr + j * 32), __builtin_bit_cast(bf16x8_t, wq[s][j]), acc);
            }
        }
        float* rb = red + buf * (4 * 4 * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) rb[(wave * 4 + r) * 64 + lane] = acc[r];
        __syncthreads();
        {
            const int r = tid >> 6;
            const float v = rb[(0 * 4 + r) * 64 + lane] + rb[(1 * 4 + r) * 64 + lane] + rb[(2 * 4 + r) * 64 + lane] + rb[(3 * 4 + r) * 64 + lane];
            const int m = g * 4 + r, n = tile * 16 + l15;
            if (m < Mb && n < N) out[(size_t)m * ldo + n] = v + (bias ? bias[n] : 0.f);
        }
        buf ^= 1;                                               // the buffer just read is written again two tiles (= one barrier) later
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[s][j] = wn[s][j];
    }
}

// ---------------------------------------------------------------------------------------------------
// Decode GEMV for 17..64 batch rows: weights streamed ONCE for all rows (the row-group loop above re-streams
// them per 16 rows).  Two launches:
//   gemv_prep_kernel   one block per batch row: attention-partial combine / LayerNorm (both optional), round to
//                      bf16 and park the row in MFMA *fragment-major* order
//                          xf[((mt * K/32 + ks) * 64 + lane) * 8 + e],  lane = g*16 + l15,
//                          row = mt*16 + l15, k = ks*32 + g*8 + e
//                      so that a wave's A operand is one fully coalesced 1 KB read;
//   gemv_mt_kernel     grid (ceil(N/16), KSPLIT): per wave K steps wave, wave+4, ..; per step 4 weight
//                      fragments, each used by MT MFMAs whose A fragments come straight from xf (L2 resident,
//                      no LDS staging, no barrier before the MFMAs); cross-wave reduction through LDS.
// ---------------------------------------------------------------------------------------------------
template <bool COMBINE>
__global__ __launch_bounds__(256) void gemv_prep_kernel(const float* __restrict__ x, int K,
                                                        const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, CombineParams cb,
                                                        bf16_t* __restrict__ xf) {
    __shared__ float s_red[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int nvec = K >> 2;
    float4 v[5];                                               // K <= 5120
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int v4 = tid + 256 * c;
        v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v4 < nvec) {
            const size_t off = (size_t)m * K + v4 * 4;
            if (!COMBINE) {
                v[c] = *(const float4*)(x + off);
            } else {
                const int head = (v4 * 4) >> 6;
                const float* ml = cb.part_ml + ((size_t)m * cb.H + head) * ATT_NS * 2;
                float mm[ATT_NS], ll[ATT_NS], M = -INFINITY;
#pragma unroll
                for (int sI = 0; sI < ATT_NS; ++sI) { mm[sI] = ml[2 * sI]; ll[sI] = ml[2 * sI + 1]; M = fmaxf(M, mm[sI]); }
                float L = 0.f;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sI = 0; sI < ATT_NS; ++sI) {
                    const float w = __expf(mm[sI] - M);
                    const float4 o = *(const float4*)(x + (size_t)sI * cb.plane + off);
                    L += ll[sI] * w;
                    r.x += w * o.x; r.y += w * o.y; r.z += w * o.z; r.w += w * o.w;
                }
                const float inv = 1.0f / L;
                v[c] = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
            }
        }
    }
    if (COMBINE && cb.cvec_out) {   // centre of the row for the column-owning out-projection behind this launch: mean(x1) from the stage's partial sums
        const float* ps = cb.pstats + ((size_t)(m >> 4) * cb.n_pstats * 16 + (m & 15)) * 2;
        const float s = tid < cb.n_pstats ? ps[(size_t)tid * 32] : 0.f;
        const float tot = block_sum(s, s_red);
        if (tid == 0) cb.cvec_out[m] = tot / (float)K;
    }
    if (ln_g) {
        float4 gq[5], bq[5];                                   // issued before the reductions
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int v4 = min(tid + 256 * c, nvec - 1);
            if (ln_b) { gq[c] = *(const float4*)(ln_g + v4 * 4); bq[c] = *(const float4*)(ln_b + v4 * 4); }
            else { gq[c] = make_float4(1.f, 1.f, 1.f, 1.f); bq[c] = make_float4(0.f, 0.f, 0.f, 0.f); }   // folded affine part
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);   // out-of-range slots hold zeros
        const float mean = block_sum(s, s_red) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (tid + 256 * c < nvec) {
                const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        const float rstd = 1.0f / sqrtf(block_sum(q, s_red) / (float)K + 1e-5f);
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int v4 = tid + 256 * c;
            if (v4 < nvec) {
                v[c].x = (v[c].x - mean) * rstd * gq[c].x + bq[c].x;
                v[c].y = (v[c].y - mean) * rstd * gq[c].y + bq[c].y;
                v[c].z = (v[c].z - mean) * rstd * gq[c].z + bq[c].z;
                v[c].w = (v[c].w - mean) * rstd * gq[c].w + bq[c].w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int v4 = tid + 256 * c;
        if (v4 < nvec) {
            const int k = v4 * 4;
            ushort4 o;
            o.x = f32_to_bf16(v[c].x); o.y = f32_to_bf16(v[c].y); o.z = f32_to_bf16(v[c].z); o.w = f32_to_bf16(v[c].w);
            *(ushort4*)(xf + frag_index(m, k, K)) = o;
        }
    }
}

// PREA (round 3): the activation fragments of all of the wave's K steps are requested up front as well (NSLOT x 4 x MT
// fragments = up to 192 VGPRs; one wave per SIMD, the register file is there).  Fetched on demand inside the MFMA loop they
// arrive a few at a time at L2 latency, and a 16-column block takes in three times as many activation bytes as weight bytes.
// NT / row groups (round 5): a block takes NT 16-column tiles and the MT row tiles blockIdx.z * MT .. of the activation rows.  At
// 64 rows a 16-column block reads 160 KB of (replicated) activation fragments for 40 KB of weights; two column tiles x two of the
// four row tiles is 80 + 80 KB for the same MFMAs.  Rows and column tiles are independent in the MFMA: every output element is
// the sum it was, in the order it was -- bit-identical to the one-tile block.
// OWN (round 6: the cross-attention out-projection of 33..64 rows without a K split -- grid (N / 32, 1, row tiles), MT = 1, NT = 2): the
// block owns its output elements, so besides x_new = resid + grid(acc + bias) it leaves what the NEXT LayerNorm projection needs and
// a preparation launch used to produce: y = x_new - c[row] in the 16-bit type in fragment-major order (c = the row's mean one stage
// earlier: y is centred to within the stage's own update, so its 16-bit rounding is that of the normalised row) and the partial sums
// (sum y, sum y^2) of the block's 32 columns per row, plain stores, one slot per (column pair, row).
// LNA (the consumer, fc1): the rows arrive as those y; the block sums the partials of its rows in a fixed order and applies the
// LayerNorm on the accumulator -- rstd (acc - mean_y wsum[n]) + bias[n], wsum[n] = row sum of the folded 16-bit weights -- exactly the
// algebra of gemv_stack_kernel / the <= 16-row fused stage.
struct MtExtra {
    const float* cvec;       // OWN: [rows] centre of every row
    bf16_t* xf_out;          // OWN: fragment-major 16-bit y rows [.. x ldo]
    float* stats_out;        // OWN: [N / 32][64][2] partial sums
    const float* stats_in;   // LNA: the same plane
    int n_stats;             // LNA: column pairs (N_producer / 32)
    const float* wsum;       // LNA: [N] row sums of the folded weights
};

template <int EPI, int MT, bool ATOMIC, int NSLOT, bool PREA = false, int NT = 1, int MODE2 = 0 /* 1: OWN, 2: LNA */>
__global__ __launch_bounds__(256) void gemv_mt_kernel(const bf16_t* __restrict__ xf, int Mb, int K, int Kb,
                                                      const bf16_t* __restrict__ W, int N, EpiParams ep, int wpk, MtExtra ex) {
    __shared__ float red[4 * NT * MT * 4 * 64];
    __shared__ float s_ln[MODE2 == 2 ? MT * 16 * 2 : 2];
    __shared__ float s_lnp[MODE2 == 2 ? 8 * MT * 16 * 2 : 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int t0 = blockIdx.z * MT;                           // first row tile of this block
    const int kbase = blockIdx.y * Kb;
    const int steps = Kb >> 7, KS = K >> 5;
    int n[NT], nc[NT];
    float bias_v[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        n[c] = n0 + c * 16 + l15;
        nc[c] = n[c] < N ? n[c] : N - 1;
        bias_v[c] = ep.bias ? ep.bias[nc[c]] : 0.f;
    }
    // operands of the OWN / LNA epilogues, requested at the head of the queue (left at their uses they are dependent loads behind the
    // MFMA loop: an L2 round trip on the tail of every block): the residual elements and row centres this thread updates, the row sums
    // of the folded weights of its columns
    float resid_v[NT][MT], cvec_v[MT], wsum_v[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) wsum_v[c] = (MODE2 == 2) ? ex.wsum[nc[c]] : 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int mr = min((t0 + t) * 16 + g * 4 + (tid >> 6), Mb - 1);
        cvec_v[t] = (MODE2 == 1) ? ex.cvec[mr] : 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) resid_v[c][t] = (MODE2 == 1) ? ep.resid[(size_t)mr * ep.ldo + nc[c]] : 0.f;
    }
    // EPI_QKV_CACHE: the cache rows of the batch rows this thread stores (m = (t0 + t) * 16 + g * 4 + tid / 64), requested at the head
    // of the queue -- the epilogue otherwise fetches each one behind a vmcnt(0): MT serial L2 round trips (found in the ISA, round 6)
    int rpos_pre[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) rpos_pre[t] = (EPI == EPI_QKV_CACHE) ? ep.row_pos[min((t0 + t) * 16 + g * 4 + (tid >> 6), Mb - 1)] : 0;

    // Fragment-major weights ONLY (round 6; the launchers refuse row-major ones and the engine then runs 17..64 rows as groups of 16 on
    // gemv2_bf16_kernel).  With the layout a run-time flag hipcc unswitched the unrolled request loop on it slot by slot -- every group
    // of four requests under its own pair of branches (ISA) -- and neither the flag by arithmetic nor one branch around the whole block
    // was as fast as no flag: batch 64 4.229 -> 4.177 ms per token step, beam step 2.389 -> 2.337.  156 instantiations: no second copy.
    (void)wpk;
    u32x4_t wq[NT][NSLOT][4];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int step = wave + 4 * s;
            step = step < steps ? step : steps - 1;               // clamped (unconditional) load, zeroed below
            const u32x4_t* wp = (const u32x4_t*)(W + ((((size_t)(nc[c] >> 4) * (K >> 5)) + (kbase >> 5) + step * 4) * 64 + g * 16 + (nc[c] & 15)) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[c][s][j] = wp[j * 64];
        }
    }
    f32x4_t acc[NT][MT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[c][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // LNA: thread (part = tid / (MT * 16), row = tid % (MT * 16)) takes column pairs part, part + PARTS, ... of its row (up to 12 loads
    // in flight with everything else); fixed order, so the statistics are the same bits in every block
    constexpr int LN_ROWS = MT * 16, LN_PARTS = 256 / LN_ROWS;
    float2 lnv[MODE2 == 2 ? 12 : 1];
    if (MODE2 == 2) {
        const int lrow = tid % LN_ROWS, lpart = tid / LN_ROWS;
        const float* sp = ex.stats_in + (size_t)(t0 * 16 + lrow) * 2;
#pragma unroll
        for (int i = 0; i < 12; ++i) lnv[i] = *(const float2*)(sp + (size_t)min(lpart + LN_PARTS * i, ex.n_stats - 1) * 128);
    }
    const u32x4_t* xq = (const u32x4_t*)xf;
    u32x4_t aq[PREA ? NSLOT : 1][4][MT];
    if (PREA) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int step = wave + 4 * s;
            step = step < steps ? step : steps - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < MT; ++t) aq[s][j][t] = xq[((size_t)(t0 + t) * KS + (kbase >> 5) + step * 4 + j) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);                     // every request is out before the first MFMA (hipcc otherwise sinks the loads to their uses)
    }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step_raw = wave + 4 * s;
        const bool live = step_raw < steps;                    // wave-uniform
        const int step = live ? step_raw : steps - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4_t w[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                w[c] = wq[c][s][j];
                if (!live) w[c] = (u32x4_t){0u, 0u, 0u, 0u};   // a dead slot contributes exactly zero
            }
            const int ks = (kbase >> 5) + step * 4 + j;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const u32x4_t a = PREA ? aq[PREA ? s : 0][j][t] : xq[((size_t)(t0 + t) * KS + ks) * 64 + lane];
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[c][t] = mfma16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, w[c]), acc[c][t]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wave * NT + c) * MT + t) * 4 + r) * 64 + lane] = acc[c][t][r];
    if (MODE2 == 2) {   // partial sums of the block's rows -> (mean_y, rstd) per row
        const int lrow = tid % LN_ROWS, lpart = tid / LN_ROWS;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float ok = (lpart + LN_PARTS * i < ex.n_stats) ? 1.f : 0.f;
            s1 += ok * lnv[i].x; s2 += ok * lnv[i].y;
        }
        s_lnp[(lpart * LN_ROWS + lrow) * 2] = s1; s_lnp[(lpart * LN_ROWS + lrow) * 2 + 1] = s2;
    }
    __syncthreads();
    if (MODE2 == 2) {
        if (tid < LN_ROWS) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < LN_PARTS; ++q) { s1 += s_lnp[(q * LN_ROWS + tid) * 2]; s2 += s_lnp[(q * LN_ROWS + tid) * 2 + 1]; }
            const float inv_k = 1.0f / (float)K;
            const float mean = s1 * inv_k;
            s_ln[tid * 2] = mean;
            s_ln[tid * 2 + 1] = 1.0f / sqrtf(fmaxf(s2 * inv_k - mean * mean, 0.f) + 1e-5f);
        }
        __syncthreads();
    }
    const int r = tid >> 6;
    static_assert(MODE2 != 2 || MT == 1 || MT == 2 || MT == 4, "LNA: the rows of a block must divide its 256 threads");
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = (t0 + t) * 16 + g * 4 + r;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(((w * NT + c) * MT + t) * 4 + r) * 64 + lane];
            const bool live = m < Mb && n[c] < N;
            if (MODE2 == 2) {
                const int lr = t * 16 + g * 4 + r;
                v = (v - s_ln[lr * 2] * wsum_v[c]) * s_ln[lr * 2 + 1];
            }
            if (live) {
                if (ATOMIC) {
                    atomicAdd(ep.outf + (size_t)m * ep.ldo + n[c], resid_grid(v + (blockIdx.y == 0 ? bias_v[c] : 0.f)));
                } else if (MODE2 == 1) {
                    const size_t o = (size_t)m * ep.ldo + n[c];
                    const float xn = resid_v[c][t] + resid_grid(v + bias_v[c]);
                    ep.outf[o] = xn;
                    const float y = xn - cvec_v[t];
                    ex.xf_out[frag_index(m, n[c], ep.ldo)] = f32_to_bf16(y);
                    s1 += y; s2 += y * y;
                } else {
                    EpiParams e2 = ep;
                    e2.bias = nullptr;
                    if (EPI == EPI_QKV_CACHE) { e2.row_pos = nullptr; e2.row_pos_pre = rpos_pre[t]; }
                    epi_store1<bf16_t, EPI>(e2, m, n[c], EPI == EPI_RESID_F32 ? resid_grid(v + bias_v[c]) : v + bias_v[c]);
                }
            }
        }
        if (MODE2 == 1) {   // sums over the block's NT x 16 columns: the lane's own tiles above, then the 16 lanes of the row
#pragma unroll
            for (int msk = 1; msk < 16; msk <<= 1) { s1 += __shfl_xor(s1, msk, 64); s2 += __shfl_xor(s2, msk, 64); }
            if (l15 == 0 && m < Mb) {
                float* so = ex.stats_out + ((size_t)blockIdx.x * 64 + m) * 2;
                so[0] = s1; so[1] = s2;
            }
        }
    }
}

// f32 parity GEMV: one wave per output column, x read from global (L2 resident); LN is applied by a
// separate kernel in f32 mode (ln_g must be null here).
template <int EPI>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ x, int Mb, int K,
                                                       const float* __restrict__ W, int N, EpiParams ep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* wr = W + (size_t)n * K;
    for (int mb = 0; mb < Mb; mb += 8) {
        float acc[8] = {};
        for (int k = lane; k < K; k += 64) {
            float w = wr[k];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (mb + i < Mb) acc[i] = fmaf(x[(size_t)(mb + i) * K + k], w, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = wave_sum(acc[i]);
            if (lane == 0 && mb + i < Mb) epi_store1<float, EPI>(ep, mb + i, n, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Host launchers
// ---------------------------------------------------------------------------------------------------
static const bf16_t* g_zero_page = nullptr;   // 256 B of zeros: DMA source for padded rows
static bool g_use_glds = true;
static bool g_use_256 = true;   // CW_NO_GEMM256=1: keep the 128x128 tiles everywhere
static int g_256_min_tiles = 200;
static int g_use_pp = -1;   // ping-pong schedule for the plain-A 256-tile shapes; -1: from the environment (CW_NO_GEMM_PP)
void cw_gemm_set_256_min_tiles(int n) { g_256_min_tiles = n; }
static int g_use_8ph = -1;  // quarter-tile (8-phase) schedule instead of ping-pong; -1: from the environment (CW_NO_GEMM_8PH)
void cw_gemm_set_pp(int on) { g_use_pp = on; }
void cw_gemm_set_8ph(int on) { g_use_8ph = on; }
void cw_gemm_set_gm(int gm) {   // experiments builds: tile order of the 8-phase kernel (see the kernel)
#ifdef CW_EXPERIMENTS
    if (gm == 0) gm = 8;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_gemm_gm), &gm, sizeof(int));
#else
    (void)gm;
#endif
}
static int g_gemv_loop = -1;  // persistent column loop for very wide LayerNorm GEMVs (logits); -1: from the environment (CW_NO_GEMV_LOOP)
void cw_gemv_set_loop(int on) { g_gemv_loop = on; }
static int g_comb_rowgroups = -1;  // combining out-projection of <= 8 rows as (N / 32, ksplit, row groups); -1: from the environment (CW_COMB_NO_ROWGROUPS)
void cw_gemv_set_comb_rowgroups(int on) { g_comb_rowgroups = on; }
static int g_use_w128 = -1;  // four waves of 128 x 128 (round 4, measured slower: -DCW_EXPERIMENTS builds only); -1: from the environment (CW_GEMM_W128=1)
void cw_gemm_set_w128(int on) { g_use_w128 = on; }

template <int EPI>
static void launch_gemm_epi(bool bf16, const AParams& ap, const void* W, int M, int N, int K, const EpiParams& ep,
                            hipStream_t st) {
    if (bf16) {
        int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
        static std::once_flag once;              // contexts may launch from several host threads
        std::call_once(once, [] {
            void* z = nullptr;
            if (hipMalloc(&z, 256) == hipSuccess) { hipMemset(z, 0, 256); g_zero_page = (const bf16_t*)z; }
            if (cw_sw::cw_switches().no_glds) g_use_glds = false;
            if (cw_sw::cw_switches().no_gemm256) g_use_256 = false;
        });
        // large shapes: 256x256 tiles once they fill most of the chip (>= 200 tiles); small M keeps the 128 tiles
        const int tm2 = (M + BM2 - 1) / BM2, tn2 = (N + BN2 - 1) / BN2;
        if (g_use_pp < 0) g_use_pp = !cw_sw::cw_switches().no_gemm_pp;
        const bool use_pp = g_use_pp != 0;
        if (g_use_8ph < 0) g_use_8ph = !cw_sw::cw_switches().no_gemm_8ph;
#ifdef CW_EXPERIMENTS
        if (g_use_w128 < 0) g_use_w128 = cw_sw::cw_switches().gemm_w128;
#else
        g_use_w128 = 0;
#endif
        if (g_use_glds && g_zero_page && g_use_256 && g_use_w128 && tm2 * tn2 >= g_256_min_tiles && ap.amode == 0 && N % BN2 == 0) {
            cw_launch_gemm_w128(EPI, (const bf16_t*)ap.A, ap.lda, (const bf16_t*)W, M, N, K, ep, tm2, tn2, st);
        } else if (g_use_glds && g_zero_page && g_use_256 && use_pp && g_use_8ph && tm2 * tn2 >= g_256_min_tiles && ap.amode == 0 && N % BN2 == 0) {
            static std::once_flag attr_8;
            std::call_once(attr_8, [] {
                hipFuncSetAttribute((const void*)gemm_bf16_8ph_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            });
            hipLaunchKernelGGL((gemm_bf16_8ph_kernel<EPI>), dim3(tm2 * tn2), dim3(512), 131072, st, (const bf16_t*)ap.A, ap.lda,
                               (const bf16_t*)W, M, N, K, ep, tn2);
#ifdef CW_EXPERIMENTS
        } else if (g_use_glds && g_zero_page && g_use_256 && use_pp && tm2 * tn2 >= g_256_min_tiles && ap.amode == 0 && N % BN2 == 0) {
            static std::once_flag attr_pp;
            std::call_once(attr_pp, [] {
                hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            });
            hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI>), dim3(tm2 * tn2), dim3(512), 131072, st, (const bf16_t*)ap.A, ap.lda,
                               (const bf16_t*)W, M, N, K, ep, tn2, g_zero_page);
#endif
        } else if (g_use_glds && g_zero_page && g_use_256 && tm2 * tn2 >= g_256_min_tiles) {
            static std::once_flag attr_once;     // one per EPI instantiation (function-local static in a template)
            std::call_once(attr_once, [] {
                hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            });
            hipLaunchKernelGGL((gemm_bf16_256_kernel<EPI>), dim3(tm2 * tn2), dim3(512), 131072, st, ap, (const bf16_t*)W, M,
                               N, K, ep, tn2, g_zero_page);
        } else if (g_use_glds && g_zero_page)
            hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI>), dim3(tm * tn), dim3(256), 65536, st, ap, (const bf16_t*)W, M,
                               N, K, ep, tn, g_zero_page);
        else
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(tm * tn), dim3(256), 0, st, ap, (const bf16_t*)W, M, N, K, ep,
                           tn);
    } else {
        int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((gemm_f32_kernel<EPI>), dim3(tm * tn), dim3(256), 0, st, ap, (const float*)W, M, N, K, ep,
                           tn);
    }
}

int cw_launch_gemm(bool bf16, int epi, const AParams& ap, const void* W, int M, int N, int K, const EpiParams& ep,
                   hipStream_t st) {
    if (K % 64 != 0 || M <= 0 || N <= 0) return CW_ERR_INVALID;
    switch (epi) {
        case EPI_STORE: launch_gemm_epi<EPI_STORE>(bf16, ap, W, M, N, K, ep, st); break;
        case EPI_GELU: launch_gemm_epi<EPI_GELU>(bf16, ap, W, M, N, K, ep, st); break;
        case EPI_RESID_F32: launch_gemm_epi<EPI_RESID_F32>(bf16, ap, W, M, N, K, ep, st); break;
        case EPI_GELU_POS_F32: launch_gemm_epi<EPI_GELU_POS_F32>(bf16, ap, W, M, N, K, ep, st); break;
        case EPI_HEADS: launch_gemm_epi<EPI_HEADS>(bf16, ap, W, M, N, K, ep, st); break;
        case EPI_STORE_F32: launch_gemm_epi<EPI_STORE_F32>(bf16, ap, W, M, N, K, ep, st); break;
        default: return CW_ERR_INVALID;
    }
    return CW_OK;
}

int cw_launch_quant_rows_fp8(const void* x, int rows, int K, void* out8, float* scale, hipStream_t st) {
    if (K % 8 != 0 || rows <= 0) return CW_ERR_INVALID;
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)x, rows, K, (unsigned char*)out8, scale);
    return CW_OK;
}

template <int EPI>
static void launch_gemm_fp8_epi(const void* A8, int lda, const void* W8, int M, int N, int K, const float* sa, const float* sw,
                                const EpiParams& ep, hipStream_t st) {
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        hipFuncSetAttribute((const void*)gemm_fp8_pp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    });
    const int tm2 = (M + BM2 - 1) / BM2, tn2 = N / BN2;
    hipLaunchKernelGGL((gemm_fp8_pp_kernel<EPI>), dim3(tm2 * tn2), dim3(512), 131072, st, (const unsigned char*)A8, lda,
                       (const unsigned char*)W8, M, N, K, sa, sw, ep, tn2);
}

// e4m3 x e4m3 GEMM with per-row scales of both operands; N % 256 == 0, K % 128 == 0, lda in bytes (= elements)
int cw_launch_gemm_fp8(int epi, const void* A8, int lda, const void* W8, int M, int N, int K, const float* sa, const float* sw,
                       const EpiParams& ep, hipStream_t st) {
    if (K % 128 != 0 || N % BN2 != 0 || M <= 0 || (size_t)M * lda >= 0x7fffffffull || (size_t)N * K >= 0x7fffffffull) return CW_ERR_INVALID;
    switch (epi) {
        case EPI_STORE: launch_gemm_fp8_epi<EPI_STORE>(A8, lda, W8, M, N, K, sa, sw, ep, st); break;
        case EPI_GELU: launch_gemm_fp8_epi<EPI_GELU>(A8, lda, W8, M, N, K, sa, sw, ep, st); break;
        case EPI_HEADS: launch_gemm_fp8_epi<EPI_HEADS>(A8, lda, W8, M, N, K, sa, sw, ep, st); break;
        case EPI_RESID_F32: launch_gemm_fp8_epi<EPI_RESID_F32>(A8, lda, W8, M, N, K, sa, sw, ep, st); break;
        default: return CW_ERR_INVALID;
    }
    return CW_OK;
}

// KC: K chunk staged in LDS per pass (bf16 path).  Mpad * (KC + 8) * 2 bytes must stay <= ~48 KB.
int cw_gemv_kc(int Mb, int K) {
    int Mpad = (Mb + 15) & ~15;
    for (int kc = K; kc >= 128; kc -= 128)
        if (K % kc == 0 && (size_t)Mpad * (kc + 8) * 2 <= 46 * 1024) return kc;
    return 128;
}

template <int EPI, int RPW, int NSLOT, int PER_LANE>
static void launch_gemv2_shape(dim3 grid, size_t lds, int ksplit, const float* x, int Mb, int K, int Kb, const void* W,
                               int N, const float* ln_g, const float* ln_b, const EpiParams& ep, hipStream_t st,
                               const CombineParams& cb, int m_base, int wpk) {
    if (EPI == EPI_RESID_F32 && cb.part_ml) {
        if (ksplit > 1)
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, true, NSLOT, PER_LANE>), grid, dim3(256), lds, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        else
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, false, true, NSLOT, PER_LANE>), grid, dim3(256), lds, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
    } else if (EPI == EPI_RESID_F32 && ksplit > 1) {
        if (N % 32 == 0 && (int)(grid.x * grid.y) > 256 && (int)(grid.x * grid.y) / 2 >= 128) {   // fc2: (80, 4) -> (40, 4)
            dim3 g2(grid.x / 2, grid.y);
            if (ep.x16)
                hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, false, NSLOT, PER_LANE, 2, true>), g2, dim3(256),
                                   lds + 4 * 4 * 64 * 4, st, x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
            else
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, false, NSLOT, PER_LANE, 2>), g2, dim3(256),
                               lds + 4 * 4 * 64 * 4, st, x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        } else if (ep.x16)
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, false, NSLOT, PER_LANE, 1, true>), grid, dim3(256), lds, st, x,
                               Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        else
        hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, false, NSLOT, PER_LANE>), grid, dim3(256), lds, st, x,
                           Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
    } else {
        // wide LayerNorm GEMVs (fc1: 320 16-column tiles on 256 CUs put two blocks on 64 CUs, whose 2 x 120 KB of loads are
        // the kernel's critical path): two column tiles per block share the activation rows and the LayerNorm work,
        // every CU gets at most one block
        // (column indices are clamped and stores masked per column, so N need not be a multiple of the block's columns.)  Very wide
        // outputs (the 51866-column logits: 3242 tiles, 12.7 per CU) take three tiles per block: every tile re-reads the 40 KB of
        // activation rows, which at one tile per block is half of all the bytes a CU takes in
        if (g_gemv_loop < 0) g_gemv_loop = !cw_sw::cw_switches().no_gemv_loop;   // A/B: 0 = three tiles per block instead of the persistent column loop
        const bool no_loop = g_gemv_loop == 0;
        if (EPI == EPI_STORE_F32 && ln_g && ksplit == 1 && grid.x >= 1024 && wpk && Kb == K && K % 128 == 0 && K <= 1280 && !cb.part_ml && !no_loop && m_base == 0) {
            // the logits projection: persistent column loop, grid resident at once (3 blocks per CU), tiles spread evenly
            const int cap = cw_sw::cw_switches().gemv_loop_cap;   // blocks in the grid per round of tiles (2 per CU measured best: 24.8 us; 3 per CU 25.9)
            const int tiles = (int)grid.x, rounds = (tiles + cap - 1) / cap, gsz = (tiles + rounds - 1) / rounds;
            const size_t lds3 = (size_t)16 * (K + 8) * 2 + 2 * 4 * 4 * 64 * 4;
            hipLaunchKernelGGL((gemv_loop_kernel<RPW, NSLOT, PER_LANE>), dim3(gsz), dim3(256), lds3, st, x, Mb, K, (const bf16_t*)W, N, ln_g, ln_b,
                               ep.bias, ep.outf, ep.ldo);
        } else if (ln_g && ksplit == 1 && grid.x >= 1024) {
            dim3 g3((grid.x + 2) / 3, 1);
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI, RPW, false, false, NSLOT, PER_LANE, 3>), g3, dim3(256), lds + 2 * 4 * 4 * 64 * 4, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        } else if (ln_g && ksplit == 1 && grid.x > 256 && grid.x / 2 >= 128) {
            dim3 g2((grid.x + 1) / 2, 1);
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI, RPW, false, false, NSLOT, PER_LANE, 2>), g2, dim3(256), lds + 4 * 4 * 64 * 4, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        } else
        hipLaunchKernelGGL((gemv2_bf16_kernel<EPI, RPW, false, false, NSLOT, PER_LANE>), grid, dim3(256), lds, st, x, Mb, K,
                           Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
    }
}

template <int EPI, int RPW>
static void launch_gemv2(const float* x, int Mb, int K, const void* W, int N, const float* ln_g, const float* ln_b,
                         const EpiParams& ep, hipStream_t st, const CombineParams* comb, int m_base, int wpk) {
    CombineParams cb{nullptr, 0, 0};
    if (comb) cb = *comb;
    // K split: only for the in-place residual epilogue (f32 atomics into the residual stream), sized so that
    // (N/16) * KSPLIT lands near the CU count and every block streams <= 1280 weights per column.
    int ksplit = 1;
    if (EPI == EPI_RESID_F32 && !ln_g && ep.outf == ep.resid) {
        const int tiles = (N + 15) / 16, steps = K / 128;
        while (tiles * ksplit < 256 && steps % (ksplit * 2) == 0 && steps / (ksplit * 2) >= 4) ksplit *= 2;
    }
    while (K / ksplit > 1280) ksplit *= 2;
    // fc2 (K = 5120, N = 1280, two column tiles per block): the largest K split that still gives every block its own CU --
    // grid (40, 5) = 200 blocks of 64 KB of weights instead of (40, 4) = 160 of 80 KB: 5.15 -> 4.98 us (8 slices: 5.47)
    const int fc2_ks = cw_sw::cw_switches().fc2_ksplit;
    if (EPI == EPI_RESID_F32 && !ln_g && !cb.part_ml && ep.outf == ep.resid && K > 1280 && N % 32 == 0) {
        if (fc2_ks > 0) { if (fc2_ks >= ksplit && K % (fc2_ks * 128) == 0) ksplit = fc2_ks; }
        else for (int ks = ksplit + 1; (N / 32) * ks <= 256; ++ks) if (K % (ks * 128) == 0) ksplit = ks;
    }
    // combining GEMV (cross-attention out-projection): a block's activation bytes are the ATT_NS partial planes of its K slice
    // (8 rows x 640 x 4 B x 6 = 123 KB at large-v3 against 20 KB of weights).  Two column tiles per block over 256-wide K
    // slices -- grid (40, 5) instead of (80, 2) -- bring the same weights with 49 KB of partials.
    // (off by default: -0.2 us per layer, but the regrouped partial sums move the residual stream by a few 2^-12 steps, and
    // one clip of the second-seed bf16 golden parts from transformers at a near-tie; CW_COMB_NT2=1)
    const bool comb_nt2 = cw_sw::cw_switches().comb_nt2;
    if (comb_nt2 && EPI == EPI_RESID_F32 && cb.part_ml && !ln_g && ep.outf == ep.resid && K % 256 == 0 && K >= 512 && N % 32 == 0) {
        const int ks = K / 256;
        const size_t lds2 = (size_t)16 * (256 + 8) * 2 + 2 * 4 * 4 * 64 * 4;
        hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, RPW, true, true, 1, 1, 2>), dim3(N / 32, ks), dim3(256), lds2, st,
                           x, Mb, K, 256, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
        return;
    }
    const int Kb = K / ksplit;
    const size_t lds = (size_t)16 * (Kb + 8) * 2 + 4 * 4 * 64 * 4;
    // combining GEMV of <= 8 rows (cross-attention out-projection): a block's bytes are the ATT_NS partial planes of its rows and K
    // slice -- 8 rows x 640 x 4 B x 6 = 123 KB against 20 KB of weights, on 160 of 256 CUs, and a CU takes in ~60 B / ns
    // (tools/gemv_stage_phase_probe.py: requests accepted at 1.9 us, reduce done at 3.1).  Same K slices (so every partial sum and
    // every rounding is the one of the (N / 16, ksplit) grid: bit-identical), but two column tiles per block and the rows in up to
    // three groups: grid (40, 2, 3) = 240 blocks of 46 KB of partials + 41 KB of weights.
    // 9..16 rows (round 6; RPW = 4 callers): the same -- 16 rows x 640 x 4 B x 6 = 246 KB of partials per 16-column block on 160 CUs
    // became grid (40, 2, 4) = 320 blocks of 61 + 41 KB: 8.18 us per launch at 16 rows before (the reference's batch_size)
    if (EPI == EPI_RESID_F32 && cb.part_ml && ksplit > 1 && (RPW == 2 || RPW == 4) && Mb > 1 && N % 32 == 0 && Kb > 256 && Kb <= 768 &&
        (g_comb_rowgroups < 0 ? !cw_sw::cw_switches().comb_no_rowgroups : g_comb_rowgroups != 0)) {
        int G = 256 / ((N / 32) * ksplit);
        G = G < 1 ? 1 : (G > Mb ? Mb : G);
        // 13..16 rows: three groups of <= 6 rows, two rows per wave (240 blocks, one per CU) rather than four groups of four
        // (320 blocks: 64 CUs would take two); CW_COMB_G4=1 keeps the four groups (A/B)
        if (RPW == 4 && Mb > 12 && (Mb + G - 1) / G <= 8 && !cw_sw::cw_switches().comb_g4) {
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, 2, true, true, 2, 3, 2>), dim3(N / 32, ksplit, G), dim3(256), lds + 4 * 4 * 64 * 4, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
            return;
        }
        while (G > 1 && (Mb + G - 1) / G > 4) ++G;             // one row per wave (RPW = 1): at most four rows per group
        if ((Mb + G - 1) / G <= 4) {
            hipLaunchKernelGGL((gemv2_bf16_kernel<EPI_RESID_F32, 1, true, true, 2, 3, 2>), dim3(N / 32, ksplit, G), dim3(256), lds + 4 * 4 * 64 * 4, st,
                               x, Mb, K, Kb, (const bf16_t*)W, N, ln_g, ln_b, ep, cb, m_base, wpk);
            return;
        }
    }
    dim3 grid((N + 15) / 16, ksplit);
    if (Kb <= 256) launch_gemv2_shape<EPI, RPW, 1, 1>(grid, lds, ksplit, x, Mb, K, Kb, W, N, ln_g, ln_b, ep, st, cb, m_base, wpk);
    else if (Kb <= 768) launch_gemv2_shape<EPI, RPW, 2, 3>(grid, lds, ksplit, x, Mb, K, Kb, W, N, ln_g, ln_b, ep, st, cb, m_base, wpk);
    else launch_gemv2_shape<EPI, RPW, 3, 5>(grid, lds, ksplit, x, Mb, K, Kb, W, N, ln_g, ln_b, ep, st, cb, m_base, wpk);
}

static bool gemv2_ok(int epi, int Mb, int K, const float* ln_g, const EpiParams& ep) {
    if (Mb > GV_MAXM || K % 128 != 0) return false;
    if (K <= 1280) return true;
    // larger K needs the K split, i.e. the in-place residual epilogue without LayerNorm
    if (epi != EPI_RESID_F32 || ln_g || ep.outf != ep.resid) return false;
    int ks = 1;
    while (K / ks > 1280) ks *= 2;
    return (K % ks == 0) && ((K / ks) % 128 == 0);
}

// which block shape the 33..64-row GEMV of an [N x K] projection takes (0 / 1 / 2 as below).  Measured at 64 and 40 rows through the
// step's own launches (tools/mt_variant_probe.py, profiles/r05_b64_mt_variants.txt): two row groups x two column tiles wins
// wherever that still leaves >= 160 blocks (q/k/v 10.6 -> 9.7 us, out-projection 6.3 -> 3.9, fc1 14.1 -> 11.1, fc2 9.9 -> 7.7);
// the 80-tile cross-attention query projection takes the row groups alone (9.9 -> 7.3)
static int cw_gemv_mt_pick(int N, int K, int ksplit) {
    (void)K;
    if (N % 32 == 0 && (N / 32) * ksplit * 2 >= 160) return 2;
    return 1;
}
static int g_mt_variant = -2;   // -2: from the environment (CW_MT_VARIANT); -1: heuristic; 0: one tile per block; 1: two row groups; 2: two row groups x two column tiles
void cw_gemv_set_mt_variant(int v) { g_mt_variant = v; }

template <int EPI, int MT>
static void launch_gemv_mt(const bf16_t* xf, int Mb, int K, const void* W, int N, const EpiParams& ep, bool allow_split,
                           hipStream_t st, int wpk) {
    int ksplit = 1;
    if (EPI == EPI_RESID_F32 && allow_split) {
        const int tiles = (N + 15) / 16, steps = K / 128;
        while (tiles * ksplit < 256 && steps % (ksplit * 2) == 0 && steps / (ksplit * 2) >= 4) ksplit *= 2;
    }
    while (K / ksplit > 1280) ksplit *= 2;
    const int Kb = K / ksplit;
    dim3 grid((N + 15) / 16, ksplit);
    const bool atomic = EPI == EPI_RESID_F32 && ksplit > 1;
    const bool prea = !cw_sw::cw_switches().mt_no_prea;   // A/B: activation fragments fetched on demand
    const int steps = Kb / 128;
    // 33..64 rows: row groups / column-tile pairs (see the kernel).  Chosen per shape from the launch table of
    // profiles/r05_b64_mt_variants.txt; CW_MT_VARIANT=0|1|2 forces one for A/B.
    if (g_mt_variant == -2) g_mt_variant = cw_sw::cw_switches().mt_variant;
    int variant = g_mt_variant;
    if (variant < 0) variant = cw_gemv_mt_pick(N, K, ksplit);
    if (MT >= 3 && prea && variant >= 1 && (variant == 1 || N % 32 == 0)) {
        constexpr int MH = 2;                                  // row tiles per group: 33..64 rows in two groups
        const int nt = variant == 2 ? 2 : 1;
        dim3 g2((unsigned)((N + 16 * nt - 1) / (16 * nt)), (unsigned)ksplit, 2);
#define CW_MT_LAUNCH2(NS, NTT)                                                                                        \
        do {                                                                                                          \
            if (atomic)                                                                                               \
                hipLaunchKernelGGL((gemv_mt_kernel<EPI_RESID_F32, MH, true, NS, true, NTT>), g2, dim3(256), 0, st, xf, Mb, K, Kb, \
                                   (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                     \
            else                                                                                                      \
                hipLaunchKernelGGL((gemv_mt_kernel<EPI, MH, false, NS, true, NTT>), g2, dim3(256), 0, st, xf, Mb, K, Kb, \
                                   (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                     \
        } while (0)
        if (nt == 2) { if (steps <= 4) CW_MT_LAUNCH2(1, 2); else if (steps <= 8) CW_MT_LAUNCH2(2, 2); else CW_MT_LAUNCH2(3, 2); }
        else { if (steps <= 4) CW_MT_LAUNCH2(1, 1); else if (steps <= 8) CW_MT_LAUNCH2(2, 1); else CW_MT_LAUNCH2(3, 1); }
#undef CW_MT_LAUNCH2
        return;
    }
#define CW_MT_LAUNCH(NS)                                                                                              \
    do {                                                                                                              \
        if (atomic && prea)                                                                                           \
            hipLaunchKernelGGL((gemv_mt_kernel<EPI_RESID_F32, MT, true, NS, true>), grid, dim3(256), 0, st, xf, Mb, K, Kb, \
                               (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                         \
        else if (atomic)                                                                                              \
            hipLaunchKernelGGL((gemv_mt_kernel<EPI_RESID_F32, MT, true, NS, false>), grid, dim3(256), 0, st, xf, Mb, K, Kb, \
                               (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                         \
        else if (prea)                                                                                                \
            hipLaunchKernelGGL((gemv_mt_kernel<EPI, MT, false, NS, true>), grid, dim3(256), 0, st, xf, Mb, K, Kb,      \
                               (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                         \
        else                                                                                                          \
            hipLaunchKernelGGL((gemv_mt_kernel<EPI, MT, false, NS, false>), grid, dim3(256), 0, st, xf, Mb, K, Kb,     \
                               (const bf16_t*)W, N, ep, wpk, MtExtra{});                                                         \
    } while (0)
    if (steps <= 4) CW_MT_LAUNCH(1);
    else if (steps <= 8) CW_MT_LAUNCH(2);
    else CW_MT_LAUNCH(3);
#undef CW_MT_LAUNCH
}

// Mb in 17..64, bf16: prep (combine / LayerNorm -> bf16 fragments in `scratch`) + one weight pass for all rows
template <int EPI>
static int launch_gemv_large(const float* x, int Mb, int K, const void* W, int N, const float* ln_g, const float* ln_b,
                             const EpiParams& ep, hipStream_t st, const CombineParams* comb, void* scratch, int wpk) {
    if (K > 5120 || K % 128 != 0) return CW_ERR_INVALID;
    if (!wpk) return CW_ERR_INVALID;                           // gemv_mt_kernel reads fragment-major weights only (see the kernel)
    bf16_t* xf = (bf16_t*)scratch;
    CombineParams cb{nullptr, 0, 0};
    if (comb) cb = *comb;
    if (!x) {                                                  // producer already wrote the fragments into `scratch`
        if (ln_g || cb.part_ml) return CW_ERR_INVALID;
    } else if (cb.part_ml) hipLaunchKernelGGL((gemv_prep_kernel<true>), dim3(Mb), dim3(256), 0, st, x, K, ln_g, ln_b, cb, xf);
    else hipLaunchKernelGGL((gemv_prep_kernel<false>), dim3(Mb), dim3(256), 0, st, x, K, ln_g, ln_b, cb, xf);
    // the K split needs the in-place residual epilogue (partials accumulate into the residual stream)
    const bool allow_split = EPI == EPI_RESID_F32 && ep.outf == ep.resid;
    int ks = 1;
    while (K / ks > 1280) ks *= 2;
    if (ks > 1 && (!allow_split || K % ks != 0 || (K / ks) % 128 != 0)) return CW_ERR_INVALID;
    const int MT = (Mb + 15) / 16;
    if (MT == 2) launch_gemv_mt<EPI, 2>(xf, Mb, K, W, N, ep, allow_split, st, wpk);
    else if (MT == 3) launch_gemv_mt<EPI, 3>(xf, Mb, K, W, N, ep, allow_split, st, wpk);
    else launch_gemv_mt<EPI, 4>(xf, Mb, K, W, N, ep, allow_split, st, wpk);
    return CW_OK;
}

template <int EPI>
static int launch_gemv_epi(bool bf16, const float* x, int Mb, int K, const void* W, int N, const float* ln_g,
                           const float* ln_b, const EpiParams& ep, hipStream_t st, const CombineParams* comb,
                           void* scratch, int wpk) {
    // (row-major weights: gemv_mt_kernel reads fragment-major ones only -- the groups of 16 rows below)
    if (bf16 && Mb > 16 && scratch && wpk) return launch_gemv_large<EPI>(x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
    if (bf16 && Mb > 16 && !x) return CW_ERR_INVALID;          // fragment-major activations have no other consumer
    if (comb && !(bf16 && gemv2_ok(EPI, Mb, K, ln_g, ep) && EPI == EPI_RESID_F32)) return CW_ERR_INVALID;
    if (bf16) {
        if (gemv2_ok(EPI, Mb, K, ln_g, ep)) {
            // batches beyond one MFMA tile (16 rows) run as row groups: the weights of the 2nd.. group come from
            // L2 / Infinity Cache (the layer's weights were just streamed by the first group)
            for (int m_base = 0; m_base < Mb; m_base += 16) {
                const int rows = Mb - m_base < 16 ? Mb - m_base : 16;
                if (rows <= 8) launch_gemv2<EPI, 2>(x, rows, K, W, N, ln_g, ln_b, ep, st, comb, m_base, wpk);
                else launch_gemv2<EPI, 4>(x, rows, K, W, N, ln_g, ln_b, ep, st, comb, m_base, wpk);
            }
            return CW_OK;
        }
        if (wpk) return CW_ERR_INVALID;                         // the first-generation kernel reads row-major weights only
        int kc = cw_gemv_kc(Mb, K);
        if (kc % 128 != 0 || K % kc != 0) return CW_ERR_INVALID;
        int Mpad = (Mb + 15) & ~15;
        size_t lds = (size_t)Mpad * (kc + 8) * 2 + 2 * GV_MAXM * 4 + 4 * 4 * 4 * 64 * 4;
        hipLaunchKernelGGL((gemv_bf16_kernel<EPI>), dim3((N + 15) / 16), dim3(256), lds, st, x, Mb, K, kc,
                           (const bf16_t*)W, N, ln_g, ln_b, ep);
    } else {
        if (ln_g || wpk) return CW_ERR_INVALID;
        hipLaunchKernelGGL((gemv_f32_kernel<EPI>), dim3((N + 3) / 4), dim3(256), 0, st, x, Mb, K, (const float*)W, N,
                           ep);
    }
    return CW_OK;
}

// 33..64 rows (round 6): x_new = resid + W a + b WITHOUT a K split or atomics -- gemv_mt_kernel OWN, grid (N / 32, 1, row tiles): two
// column tiles x one row tile per block (80 KB of weights + 40 KB of activation fragments at K = 1280) -- leaving the next LayerNorm
// projection's 16-bit rows and partial sums behind (see the kernel).  The consumer is cw_launch_gemv_lna.
// column tiles per block of the column-owning GEMV = partial-sum slots per row N / (16 nt); CW_OWN_NT=1|2 (A/B)
int cw_gemv_own_nt(int N) {
    (void)N;   // one tile (80 slots at d = 1280, grid (80, 1, row tiles)): 2.366 ms per beam step at 8 x 5 rows against 2.411 at two, batch 40 3.227 against 3.260
    return cw_sw::cw_switches().own_nt == 2 ? 2 : 1;
}
int cw_launch_gemv_own(const void* xf, int Mb, int K, const void* W, int N, const EpiParams& ep, const float* cvec, void* xf_out,
                       float* stats_out, hipStream_t st, bool wpacked) {
    if (Mb < 17 || Mb > GV_MAXM || K % 128 || K > 1536 || N % 32 || N / 16 > 96 || !xf || !cvec || !xf_out || !stats_out || !wpacked) return CW_ERR_INVALID;
    if (!ep.outf || !ep.resid || ep.ldo != N || xf == xf_out) return CW_ERR_INVALID;
    const int steps = K / 128, wpk = wpacked ? 1 : 0;
    const MtExtra ex{cvec, (bf16_t*)xf_out, stats_out, nullptr, 0, nullptr};
    const int nt = cw_gemv_own_nt(N);
    const dim3 grid((unsigned)(N / (16 * nt)), 1, (unsigned)((Mb + 15) / 16));
#define CW_OWN_LAUNCH(NS, NTT) hipLaunchKernelGGL((gemv_mt_kernel<EPI_RESID_F32, 1, false, NS, true, NTT, 1>), grid, dim3(256), 0, st, (const bf16_t*)xf, Mb, K, K, (const bf16_t*)W, N, ep, wpk, ex)
    if (nt == 2) { if (steps <= 4) CW_OWN_LAUNCH(1, 2); else if (steps <= 8) CW_OWN_LAUNCH(2, 2); else CW_OWN_LAUNCH(3, 2); }
    else { if (steps <= 4) CW_OWN_LAUNCH(1, 1); else if (steps <= 8) CW_OWN_LAUNCH(2, 1); else CW_OWN_LAUNCH(3, 1); }
#undef CW_OWN_LAUNCH
    return CW_OK;
}

// ... and the LayerNorm projection behind it (fc1 + GELU -> fragment-major rows): the rows arrive as y = x - c in 16 bits, the
// LayerNorm is applied on the accumulator from the producer's partial sums (gemv_mt_kernel LNA); block shape of the 33..64-row GEMVs
int cw_launch_gemv_lna(const void* xf, int Mb, int K, const void* W, int N, const EpiParams& ep, const float* stats_in, int n_stats,
                       const float* wsum, hipStream_t st, bool wpacked) {
    if (Mb < 33 || Mb > GV_MAXM || K % 128 || K > 1536 || N % 32 || !xf || !stats_in || !wsum || n_stats < 1 || n_stats > 96 || !ep.out || !wpacked) return CW_ERR_INVALID;
    const int steps = K / 128, wpk = wpacked ? 1 : 0;
    const dim3 grid((unsigned)(N / 32), 1, 2);
    const MtExtra ex{nullptr, nullptr, nullptr, stats_in, n_stats, wsum};
    if (steps <= 4) hipLaunchKernelGGL((gemv_mt_kernel<EPI_GELU_FRAG, 2, false, 1, true, 2, 2>), grid, dim3(256), 0, st, (const bf16_t*)xf, Mb, K, K, (const bf16_t*)W, N, ep, wpk, ex);
    else if (steps <= 8) hipLaunchKernelGGL((gemv_mt_kernel<EPI_GELU_FRAG, 2, false, 2, true, 2, 2>), grid, dim3(256), 0, st, (const bf16_t*)xf, Mb, K, K, (const bf16_t*)W, N, ep, wpk, ex);
    else hipLaunchKernelGGL((gemv_mt_kernel<EPI_GELU_FRAG, 2, false, 3, true, 2, 2>), grid, dim3(256), 0, st, (const bf16_t*)xf, Mb, K, K, (const bf16_t*)W, N, ep, wpk, ex);
    return CW_OK;
}

// 17..64 rows: combine the key-split cross-attention partials into 16-bit fragment-major rows (the consumer is a
// gemv_rows_kernel producer launched separately, decfuse.hip)
int cw_launch_rows_combine(const float* part_o, int Mb, int K, const CombineParams& cb, void* xf, hipStream_t st) {
    if (Mb < 1 || Mb > GV_MAXM || K % 128 || K > 5120 || !cb.part_ml || !xf) return CW_ERR_INVALID;
    hipLaunchKernelGGL((gemv_prep_kernel<true>), dim3(Mb), dim3(256), 0, st, part_o, K, (const float*)nullptr, (const float*)nullptr, cb, (bf16_t*)xf);
    return CW_OK;
}

int cw_launch_gemv(bool bf16, int epi, const float* x, int Mb, int K, const void* W, int N, const float* ln_g,
                   const float* ln_b, const EpiParams& ep, hipStream_t st, const CombineParams* comb, void* scratch, bool wpacked) {
    const int wpk = wpacked ? 1 : 0;
    // 16-bit activation rows: only the K-split residual GEMV of <= 16 rows takes them (fc2 behind an EPI_GELU fc1)
    if (ep.x16 && !(bf16 && epi == EPI_RESID_F32 && !ln_g && !comb && Mb <= 16 && K > 1280 && ep.outf == ep.resid)) return CW_ERR_INVALID;
    if (Mb <= 0 || Mb > GV_MAXM || K % 128 != 0) return CW_ERR_INVALID;
    if (!x && !(bf16 && Mb > 16 && scratch)) return CW_ERR_INVALID;
    switch (epi) {
        case EPI_GELU_F32: return launch_gemv_epi<EPI_GELU_F32>(bf16, x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        case EPI_GELU:                                           // 16-bit row-major output [Mb][ldo] for a consumer launched with ep.x16
            if (!bf16 || Mb > 16) return CW_ERR_INVALID;
            return launch_gemv_epi<EPI_GELU>(bf16, x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        case EPI_GELU_FRAG:
            if (!(bf16 && Mb > 16 && scratch)) return CW_ERR_INVALID;
            return launch_gemv_large<EPI_GELU_FRAG>(x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        case EPI_RESID_F32: return launch_gemv_epi<EPI_RESID_F32>(bf16, x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        case EPI_STORE_F32: return launch_gemv_epi<EPI_STORE_F32>(bf16, x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        case EPI_QKV_CACHE: return launch_gemv_epi<EPI_QKV_CACHE>(bf16, x, Mb, K, W, N, ln_g, ln_b, ep, st, comb, scratch, wpk);
        default: return CW_ERR_INVALID;
    }
}


// ---------------------------------------------------------------------------------------------------
// Fragment-major weight layout of the decode GEMVs (round 3).  A wave instruction of the row-major weight stream fetches, for
// 16 output columns, 64 contiguous bytes of each row: 16 half cache lines at a stride of one row; the stream then runs at
// 3.7 TB/s at best (the 133 MB logits GEMV: 36 us).  Packed, element (n, k) of a [N][K] matrix sits at
//     (((n >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (n & 15)) * 8 + (k & 7)          (= frag_index(n, k, K))
// so the 64 lanes' 16-byte MFMA fragments of one (16-column tile, 32-wide k step) are 1 KB of contiguous memory: 8 full lines
// per instruction.  Rows are padded to a multiple of 16 (the pad rows repeat the last row and are never stored).  The engine
// packs the decoder's GEMV matrices once, after folding; the kernels take either layout (`wpk`).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wfrag_pack_kernel(const bf16_t* __restrict__ src, int N, int K, bf16_t* __restrict__ dst,
                                                         long long n_chunks) {
    const long long ci = (long long)blockIdx.x * 256 + threadIdx.x;   // one 16-byte chunk (8 elements) of the packed image
    if (ci >= n_chunks) return;
    const int lane = (int)(ci & 63);
    const long long tk = ci >> 6;
    const int KS = K >> 5;
    const int kj = (int)(tk % KS);
    const int tile = (int)(tk / KS);
    int n = tile * 16 + (lane & 15);
    n = n < N ? n : N - 1;
    const int k = kj * 32 + (lane >> 4) * 8;
    *(u32x4_t*)(dst + ci * 8) = *(const u32x4_t*)(src + (size_t)n * K + k);
}

size_t cw_wfrag_elems(int N, int K) { return (size_t)((N + 15) & ~15) * K; }
int cw_launch_wfrag_pack(const void* src, int N, int K, void* dst, hipStream_t st) {
    if (K % 32 || N < 1) return CW_ERR_INVALID;
    const long long n_chunks = (long long)(cw_wfrag_elems(N, K) / 8);
    hipLaunchKernelGGL(wfrag_pack_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)src, N, K,
                       (bf16_t*)dst, n_chunks);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm affine folding (bf16 decode engine): a projection fed by LN(x) = n(x) * g + beta is rewritten as
//     W LN(x) + b = (W diag(g)) n(x) + (b + W beta)
// once, at weight-load time, from the f32 checkpoint values (single bf16 rounding of W diag(g)), so the decode GEMVs
// neither fetch the LayerNorm parameters nor apply them (measured 0.3-0.56 us per launch on a 5-7 us latency chain).
// One wave per output row: w_out[n][:] = bf16(scale * W[n][:] * g[:]),  bias[n] += scale * sum_k W[n][k] beta[k].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ Wf, int N, int K,
                                                             const float* __restrict__ g, const float* __restrict__ beta,
                                                             float scale, bf16_t* __restrict__ w_out,
                                                             float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* wr = Wf + (size_t)n * K;
    float dot = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *(const float4*)(wr + k), gg = *(const float4*)(g + k), bb = *(const float4*)(beta + k);
        dot += (w.x * bb.x + w.y * bb.y) + (w.z * bb.z + w.w * bb.w);
        ushort4 o;
        o.x = f32_to_bf16(scale * w.x * gg.x); o.y = f32_to_bf16(scale * w.y * gg.y);
        o.z = f32_to_bf16(scale * w.z * gg.z); o.w = f32_to_bf16(scale * w.w * gg.w);
        *(ushort4*)(w_out + (size_t)n * K + k) = o;
    }
    dot = wave_sum(dot);
    if (lane == 0) bias[n] += scale * dot;
}
int cw_launch_fold_layernorm(const float* Wf, int N, int K, const float* g, const float* beta, float scale, void* w_out,
                             float* bias, hipStream_t st) {
    if (K % 4) return CW_ERR_INVALID;
    hipLaunchKernelGGL(fold_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, st, Wf, N, K, g, beta, scale, (bf16_t*)w_out, bias);
    return CW_OK;
}

}  // namespace CW_NS
