// Decoder linear layers for 17..64 rows (batch 17..64 greedy, items x beams under beam search): skinny-M MFMA GEMM.
//
// Shape of the problem (TF modeling_whisper.py:448-505 at large-v3): M = 17..64 rows against weight matrices of 3.3-13 MB that
// must stream from HBM once per token step.  A 16-column GEMV block re-reads every row's activations for its K range (164 KB
// of 16-bit rows against 40 KB of weights at 64 rows): the chip then moves four times the weight bytes and each CU's intake,
// not HBM, bounds the launch.  Here a block owns a 64-column x Kb tile for ALL rows:
//   * weights: fragment-major (gemm.hip: wfrag_pack_kernel), every wave streams its own 16-column tile, all requests up front,
//     non-temporal, one 1 KB line group per instruction;
//   * activations: staged ONCE per block in LDS as 16-bit MFMA A fragments (MT x Kb/32 KB) and read by the four waves;
//   * K is split over grid.y so that (N / 64) x S lands near the CU count with 30-130 KB per block.
// The K split makes every block's result a partial sum, so the kernel has exactly two outputs:
//   OUT_ATOMIC   linear epilogues (the three residual projections): f32 atomics into the residual stream on its 2^-12 grid
//                (common.h: resid_grid; exact, hence order-independent and bit-reproducible);
//   OUT_PLANE    projections that feed a non-linearity (q/k/v -> cache + softmax, fc1 -> GELU): the partial goes to plane
//                [slice][row][col] and skinny_finish_kernel sums the planes in fixed order.
// LayerNorm never runs as its own pass: with the affine part folded into W' / b' at load time (gemm.hip:
// fold_layernorm_kernel),  W LN(x) = rstd (W' x - mean W'1) + b'  is applied by the finish kernel from the row statistics it
// takes of x itself (two-pass, f32) -- the GEMM needs no statistics and reads the f32 residual rows directly (ACT_F32).
// Rounding x to 16 bits BEFORE the mean is removed would lose |mean| / std of precision in the cancellation; so a block
// rounds x - c, c = the row's mean over the block's own K slice (what it has in registers), and adds c * (W' 1 over the slice)
// back in f32 -- that slice row sum comes out of the same weight fragments through one extra MFMA against a row of ones.
// The rounded operand is then as well conditioned as the rounded LN(x) of a prepared row.
#include "common.h"
#include "kernels.h"
#include <mutex>

namespace CW_NS {

#ifdef CW_EXPERIMENTS   // measured against the round-3 path in the step and rejected (profiles/r04_b64_*): A/B builds only
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ static inline f32x4_t mfma16s(bf16x8_t a, bf16x8_t b, f32x4_t c) { return cw_mfma_16x16x32(a, b, c); }
__device__ static inline void sk_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

enum { SK_ACT_F32 = 0, SK_ACT_FRAG = 1 };
enum { SK_OUT_PLANE = 0, SK_OUT_ATOMIC = 1 };

// grid (ceil(N / 64), S), 256 threads.  Wave w owns the 16-column tile blockIdx.x * 4 + w for all MT row tiles and the block's
// nks = Kb / 32 <= NKS k-steps (slots past nks multiply by a zero weight fragment: the loop stays branch-free so that the LDS
// reads of a k-step are issued under the MFMAs of the previous one).
// LDS: max([MT][nks][64 lanes] 16-byte A fragments, [MT * 16][65] f32 output tile) + 64 floats of row centres.
__host__ __device__ static inline int sk_lds_main(int MT, int nks) {
    const int a = MT * nks * 1024, t = MT * 16 * 65 * 4;
    return a > t ? a : t;
}

template <int MT, int NKS, int ACT, int OUT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    u32x4_t* afr = (u32x4_t*)sk_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nks = p.Kb >> 5, KS = p.K >> 5;
    const int ks0 = blockIdx.y * nks;
    const int ntiles = (p.N + 15) >> 4;
    const int tile_raw = blockIdx.x * 4 + wave;
    const bool tile_ok = tile_raw < ntiles;                       // wave-uniform
    const int tile = tile_ok ? tile_raw : ntiles - 1;
    float* cs = (float*)(sk_smem + sk_lds_main(MT, nks));         // [MT * 16] slice means of the rows (ACT_F32)

    // ---- weight stream: every request of the wave is out before anything else
    u32x4_t wq[NKS];
    {
        const u32x4_t* wp = (const u32x4_t*)p.W + ((size_t)tile * KS + ks0) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NKS; ++s) wq[s] = CW_STREAM_LD(wp + (size_t)(s < nks ? s : nks - 1) * 64);
    }
#ifdef CW_SK_DEBUG
    const int dbg = p.dbg;
    if (dbg & 8) for (int s = 0; s < NKS; ++s) wq[s] = (u32x4_t){0u, 0u, 0u, 0u};
#else
    const int dbg = 0;
#endif

    // ---- activations -> LDS fragments, once per block
    if (dbg & 1) {
    } else if (ACT == SK_ACT_FRAG) {
        // the source is already fragment-major: chunk (mt, s) = 1 KB contiguous on both sides, moved by LDS-DMA
        const u32x4_t* xq = (const u32x4_t*)p.xf;
        const int items = MT * nks;
        for (int it = wave; it < items; it += 4) {
            const int mt = it / nks, s = it - mt * nks;
            sk_glds16(xq + ((size_t)mt * KS + ks0 + s) * 64 + lane, afr + (size_t)it * 64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wave < MT) {
        // wave mt: the 16 rows of row tile mt over the block's K slice.  One instruction = 8 rows x one full 128-byte line
        // (8 lanes x 16 B per row): lane -> (row lane / 8 of the half, 4 floats oct * 4 .. of the k-step)
        const int oct = lane & 7, r8 = lane >> 3;
        float4 v[2 * NKS];
        bool live[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = wave * 16 + h * 8 + r8;
            live[h] = row < p.Mb;
            const float* xr = p.x + (size_t)(live[h] ? row : 0) * p.K + (size_t)ks0 * 32 + oct * 4;
#pragma unroll
            for (int s = 0; s < NKS; ++s) v[2 * s + h] = *(const float4*)(xr + (s < nks ? s : nks - 1) * 32);
        }
        float c[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s)
                if (s < nks) sum += (v[2 * s + h].x + v[2 * s + h].y) + (v[2 * s + h].z + v[2 * s + h].w);
            sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);   // the row's 8 lanes
            c[h] = live[h] ? sum / (float)p.Kb : 0.f;
            if (oct == 0) cs[wave * 16 + h * 8 + r8] = c[h];
            // slice statistics of the row for the consumer's LayerNorm (combined over the S slices in fixed order): mean and
            // the sum of squares ABOUT that mean -- no E[x^2] - mean^2 cancellation.  One column tile writes them.
            if (p.stats && blockIdx.x == 0) {
                float m2 = 0.f;
#pragma unroll
                for (int s = 0; s < NKS; ++s)
                    if (s < nks) {
                        const float4 q = v[2 * s + h];
                        const float a0 = q.x - c[h], a1 = q.y - c[h], a2 = q.z - c[h], a3 = q.w - c[h];
                        m2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                    }
                m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64);
                if (oct == 0 && live[h]) *(float2*)(p.stats + ((size_t)blockIdx.y * p.Mb + wave * 16 + h * 8 + r8) * 2) = make_float2(c[h], m2);
            }
        }
        // element (row16, k) of a fragment sits at lane' = (k / 8) * 16 + row16, slot k % 8: 8 bytes per lane
        unsigned char* dst0 = sk_smem + (size_t)wave * nks * 1024 + ((oct >> 1) * 16 + r8) * 16 + (oct & 1) * 8;
#pragma unroll
        for (int s = 0; s < NKS; ++s)
            if (s < nks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 q = v[2 * s + h];
                    uint2 o = make_uint2(0u, 0u);
                    if (live[h]) {
                        o.x = (unsigned)f32_to_bf16(q.x - c[h]) | ((unsigned)f32_to_bf16(q.y - c[h]) << 16);
                        o.y = (unsigned)f32_to_bf16(q.z - c[h]) | ((unsigned)f32_to_bf16(q.w - c[h]) << 16);
                    }
                    *(uint2*)(dst0 + s * 1024 + h * 128) = o;
                }
    }
    __syncthreads();

    // ---- products: MT row tiles share every weight fragment
    f32x4_t acc[MT], accw = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned one2 = (unsigned)f32_to_bf16(1.0f) * 0x10001u;
    const u32x4_t ones = (u32x4_t){one2, one2, one2, one2};
    if (dbg & 2) {
#pragma unroll
        for (int s = 0; s < NKS; ++s) acc[0] += __builtin_bit_cast(f32x4_t, wq[s]);
    } else {
        // the A fragments of k-step s + 1 are requested before the MFMAs of k-step s are issued (MT LDS reads in flight)
        u32x4_t an[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) an[t] = afr[((size_t)t * nks) * 64 + lane];
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            u32x4_t ac[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) ac[t] = an[t];
            if (s + 1 < NKS) {
                const int sn = s + 1 < nks ? s + 1 : 0;            // a dead slot reads slot 0: finite values times a zero weight fragment
#pragma unroll
                for (int t = 0; t < MT; ++t) an[t] = afr[((size_t)t * nks + sn) * 64 + lane];
            }
            u32x4_t bw = wq[s];
            if (s >= nks) bw = (u32x4_t){0u, 0u, 0u, 0u};          // block-uniform: a dead slot contributes exactly zero
            const bf16x8_t b = __builtin_bit_cast(bf16x8_t, bw);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = mfma16s(__builtin_bit_cast(bf16x8_t, ac[t]), b, acc[t]);
            if (ACT == SK_ACT_F32) accw = mfma16s(__builtin_bit_cast(bf16x8_t, ones), b, accw);   // every row: sum_k W'[n][k] over the slice
        }
    }
    if ((dbg & 4) && acc[0][0] != 12345.678f) return;

    // ---- D[row = t*16 + g*4 + r][col = tile*16 + l15]
    if (OUT == SK_OUT_ATOMIC) {
        const int n = tile * 16 + l15;
        if (!tile_ok || n >= p.N) return;
        const float bias_v = (p.bias && blockIdx.y == 0) ? p.bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = t * 16 + g * 4 + r;
                if (m < p.Mb) atomicAdd(p.outf + (size_t)m * p.ldo + n, resid_grid(acc[t][r] + bias_v));
            }
        return;
    }
    // planes: through an LDS tile so that a store instruction covers 4 rows x 256 contiguous bytes
    float cv[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[t][r] = ACT == SK_ACT_F32 ? cs[t * 16 + g * 4 + r] : 0.f;
    __syncthreads();                                               // every wave is done with the A fragments
    float* tl = (float*)sk_smem;                                   // [MT * 16][65]
    if (tile_ok) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[(t * 16 + g * 4 + r) * 65 + wave * 16 + l15] = fmaf(cv[t][r], accw[r], acc[t][r]);
    }
    __syncthreads();
    float* dst = p.planes + (size_t)blockIdx.y * p.Mb * p.N;
    for (int idx = tid; idx < MT * 16 * 16; idx += 256) {
        const int m = idx >> 4, c4 = (idx & 15) * 4;
        const int n = blockIdx.x * 64 + c4;
        if (m < p.Mb && n < p.N) {
            const float* q = tl + m * 65 + c4;
            *(float4*)(dst + (size_t)m * p.N + n) = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
}

// Planes -> destination: one block per (row, 1024 columns).  The block takes the LayerNorm statistics of its row of x (two
// passes over K <= 5120 floats held in registers), sums the S partial planes in slice order and stores
//     EPI(rstd (sum_s P_s - mean wsum) + bias)        EPI in {STORE_F32, QKV_CACHE, GELU_FRAG}
// x == null: no LayerNorm (mean 0, rstd 1).
template <int EPI>
__global__ __launch_bounds__(256) void skinny_finish_kernel(SkinnyFinishParams p) {
    __shared__ float s_red[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int n = (blockIdx.y * 256 + tid) * 4;
    const bool col_ok = n < p.N;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), ws = acc, bs = acc;
    if (col_ok) {                                                  // requested before the reductions below
        for (int s = 0; s < p.S; ++s) {
            const float4 q = *(const float4*)(p.planes + ((size_t)s * p.Mb + m) * p.N + n);
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
        }
        if (p.wsum) ws = *(const float4*)(p.wsum + n);
        if (p.ep.bias) bs = *(const float4*)(p.ep.bias + n);
    }
    float mean = 0.f, rstd = 1.f;
    if (p.stats) {
        // LayerNorm statistics of the row from the S slice records the GEMM left (mean_s, M2_s over Kb values each):
        // mean = avg mean_s,  M2 = sum M2_s + Kb sum (mean_s - mean)^2   -- every thread the same S small loads, no reduction
        float ms[16], m2 = 0.f;
        const int S = p.S < 16 ? p.S : 16;
        for (int s = 0; s < S; ++s) {
            const float2 r = *(const float2*)(p.stats + ((size_t)s * p.Mb + m) * 2);
            ms[s] = r.x; mean += r.x; m2 += r.y;
        }
        mean /= (float)S;
        const float kb = (float)p.K / (float)S;
        for (int s = 0; s < S; ++s) m2 += kb * (ms[s] - mean) * (ms[s] - mean);
        rstd = 1.0f / sqrtf(m2 / (float)p.K + 1e-5f);
    } else if (p.x) {
        const int nvec = p.K >> 2;
        float4 v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int v4 = tid + 256 * c;
            v[c] = v4 < nvec ? *(const float4*)(p.x + (size_t)m * p.K + v4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        mean = block_sum(s, s_red) / (float)p.K;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (tid + 256 * c < nvec) {
                const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        rstd = 1.0f / sqrtf(block_sum(q, s_red) / (float)p.K + 1e-5f);
    }
    if (!col_ok) return;
    float o[4] = {rstd * (acc.x - mean * ws.x) + bs.x, rstd * (acc.y - mean * ws.y) + bs.y,
                  rstd * (acc.z - mean * ws.z) + bs.z, rstd * (acc.w - mean * ws.w) + bs.w};
    const EpiParams& ep = p.ep;
    if (EPI == EPI_STORE_F32) {
        *(float4*)(ep.outf + (size_t)m * ep.ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (EPI == EPI_GELU_FRAG) {                             // n % 4 == 0: the four elements are contiguous in a fragment
        Pack4<bf16_t>::st((bf16_t*)ep.out + frag_index(m, n, ep.ldo), gelu_erf(o[0]), gelu_erf(o[1]), gelu_erf(o[2]), gelu_erf(o[3]));
    } else if (EPI == EPI_QKV_CACHE) {
        const int which = n / ep.d_model, r = n - which * ep.d_model;   // d_model % 4 == 0: the four columns share `which` and head
        if (which == 0) {
            *(float4*)(ep.outf + (size_t)m * ep.d_model + r) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            const int h = r >> 6, dd = r & 63;
            bf16_t* base = (bf16_t*)(which == 1 ? ep.out1 : ep.out2);
            Pack4<bf16_t>::st(base + (((size_t)m * ep.H + h) * ep.S_pad + ep.row_pos[m]) * 64 + dd, o[0], o[1], o[2], o[3]);
        }
    }
}

__global__ void skinny_empty_kernel(int) {}
void cw_launch_skinny_empty(hipStream_t st) { hipLaunchKernelGGL(skinny_empty_kernel, dim3(256), dim3(256), 0, st, 0); }

// ---- host side -------------------------------------------------------------------------------------------------------------
// K split of a launch: nks = k-steps (of 32) per block, a divisor of K / 32 that is <= 16; S = K / (32 nks) slices.  The
// (N / 64) x S blocks should land near the CU count (one or two light blocks per CU); planes are capped by `s_max`.
int cw_skinny_pick_nks(int N, int K, int s_max) {
    const int KS = K >> 5, ctiles = (N + 63) / 64;
    int best = 0;
    long best_cost = 0;
    for (int nks = 1; nks <= 16 && nks <= KS; ++nks) {
        if (KS % nks) continue;
        const int S = KS / nks;
        if (s_max > 0 && S > s_max) continue;
        const long blocks = (long)ctiles * S;
        // distance from ~256 blocks, over-subscription (two blocks on a CU) weighted like under-subscription; very thin
        // slices (one or two k-steps) pay the fixed block cost too often
        long cost = blocks < 256 ? (256 - blocks) * 2 : (blocks - 256);
        if (nks < 4) cost += 64 * (4 - nks);
        if (!best || cost < best_cost) { best = nks; best_cost = cost; }
    }
    return best;
}

template <int MT, int ACT, int OUT>
static int launch_skinny_mt(const SkinnyParams& p, int nks, hipStream_t st) {
    const dim3 grid((p.N + 63) / 64, (p.K >> 5) / nks);
    const size_t lds = (size_t)sk_lds_main(MT, nks) + 64 * 4;
#define CW_SK_LAUNCH(NKS)                                                                                                   \
    do {                                                                                                                    \
        static std::once_flag attr;                                                                                         \
        std::call_once(attr, [] {                                                                                           \
            (void)hipFuncSetAttribute((const void*)skinny_kernel<MT, NKS, ACT, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
        });                                                                                                                 \
        hipLaunchKernelGGL((skinny_kernel<MT, NKS, ACT, OUT>), grid, dim3(256), lds, st, p);                                \
    } while (0)
    if (nks <= 4) CW_SK_LAUNCH(4);
    else if (nks <= 8) CW_SK_LAUNCH(8);
    else if (nks <= 10) CW_SK_LAUNCH(10);
    else CW_SK_LAUNCH(16);
#undef CW_SK_LAUNCH
    return CW_OK;
}

// mode 0: f32 rows -> planes (the consumer is cw_launch_skinny_finish); mode 1: 16-bit fragment-major rows -> residual atomics.
// nks: k-steps per block (cw_skinny_pick_nks), must divide K / 32.
int cw_launch_skinny(int mode, const SkinnyParams& p0, int nks, hipStream_t st) {
    SkinnyParams p = p0;
    if (p.Mb < 1 || p.Mb > 64 || p.K % 32 || p.N < 1 || nks < 1 || nks > 16 || (p.K >> 5) % nks || !p.W) return CW_ERR_INVALID;
    if (mode == 0 && (!p.x || !p.planes || p.N % 4)) return CW_ERR_INVALID;
    if (mode == 1 && (!p.xf || !p.outf)) return CW_ERR_INVALID;
    p.Kb = nks * 32;
    const int MT = (p.Mb + 15) / 16;
#define CW_SK_MODE(MTV)                                                                                \
    (mode == 0 ? launch_skinny_mt<MTV, SK_ACT_F32, SK_OUT_PLANE>(p, nks, st)                         \
               : launch_skinny_mt<MTV, SK_ACT_FRAG, SK_OUT_ATOMIC>(p, nks, st))
    if (MT == 1) return CW_SK_MODE(1);
    if (MT == 2) return CW_SK_MODE(2);
    if (MT == 3) return CW_SK_MODE(3);
    return CW_SK_MODE(4);
#undef CW_SK_MODE
}

int cw_launch_skinny_finish(int epi, const SkinnyFinishParams& p, hipStream_t st) {
    if (p.Mb < 1 || p.N % 4 || p.S < 1 || !p.planes || (p.x && (p.K % 4 || p.K > 5120)) || (p.stats && p.S > 16)) return CW_ERR_INVALID;
    const dim3 grid(p.Mb, (p.N + 1023) / 1024);
    switch (epi) {
        case EPI_STORE_F32: hipLaunchKernelGGL((skinny_finish_kernel<EPI_STORE_F32>), grid, dim3(256), 0, st, p); break;
        case EPI_GELU_FRAG: hipLaunchKernelGGL((skinny_finish_kernel<EPI_GELU_FRAG>), grid, dim3(256), 0, st, p); break;
        case EPI_QKV_CACHE:
            if (p.ep.d_model % 64) return CW_ERR_INVALID;
            hipLaunchKernelGGL((skinny_finish_kernel<EPI_QKV_CACHE>), grid, dim3(256), 0, st, p);
            break;
        default: return CW_ERR_INVALID;
    }
    return CW_OK;
}

#else
int cw_skinny_pick_nks(int, int, int) { return 0; }
int cw_launch_skinny(int, const SkinnyParams&, int, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_skinny_finish(int, const SkinnyFinishParams&, hipStream_t) { return CW_ERR_INVALID; }
void cw_launch_skinny_empty(hipStream_t) {}
#endif

}  // namespace CW_NS
