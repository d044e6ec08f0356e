// Six-launch decoder layer of the 16-bit engine (batch rows <= 16; 17..64 greedy rows run the same stage in groups of 16 rows,
// gemv_stack_kernel's grid y, inside their ten-launch layer): LayerNorm is linear up to two per-row scalars,
//     W LN(x) = rstd * (W' x - mean * (W' 1)) + b',      W' = W diag(gamma),  b' = b + W beta,
// so a projection that follows "residual add -> LayerNorm" can be applied to the *summands* of the residual before
// the statistics exist, and the statistics are applied by the consumer.  With x1 = x + Wo a + bo (self-attention
// out-projection, TF modeling_whisper.py:471-476) and q_c = Wq_c LN_c(x1) (:485-493)
//     q_c = rstd(x1) * ( W'q_c x + (W'q_c Wo) a + W'q_c bo - mean(x1) * (W'q_c 1) ) + b'q_c
// the two dependent GEMVs "out-projection" and "cross-attention query" become ONE stage over the stacked matrix
// [W'q_c ; W'q_c Wo ; Wo] (inputs x, a, a), and in the same way cross-attention out-projection + fc1 (:498-502) become one
// stage over [W'1 ; W'1 Wo_c ; Wo_c].  The product matrices are formed once at weight-load time in f32
// (fold_product_kernel) and rounded to the engine's 16-bit type once.  A decoder layer is then 6 dependent launches
// instead of 8:  qkv | self-attention | X1 | cross-attention (finishes q_c) | X2 | fc2 (finishes gelu(fc1)).
// Cost: the product matrices add 16.4 MB of weight stream per layer at large-v3 (56.6 -> 73 MB); what it buys is two
// launch boundaries + kernel ramps per layer on a chain that is latency-bound.
//
//   fold_product_kernel / fold_rowvec_kernel   load-time: C = (A diag(s) scale) B in f32 -> 16 bit;  c = A' v,  w = W16 1
//   gemv_stack_kernel    one launch over a row-stacked weight matrix: up to 3 segments, each with its own f32 input rows
//                        and epilogue (plain store, or residual add on the 2^-12 grid into a second buffer); NT 16-column
//                        tiles per block share the activation rows; same weight streaming as gemv2_bf16_kernel (gemm.hip)
//   gemv_fc2x_kernel     fc2 whose activation load finishes fc1: mid = gelu(rstd (u_a + u_b - mean w1) + b1') from the two
//                        partial projections and the LayerNorm statistics of the residual row it computes wave-locally
#include "common.h"
#include "kernels.h"

namespace CW_NS {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ static inline f32x4_t mfma16x(bf16x8_t a, bf16x8_t b, f32x4_t c) { return cw_mfma_16x16x32(a, b, c); }

// ---------------------------------------------------------------------------------------------------
// load-time products (f32 VALU, 64 x 64 output tile per block, run once per checkpoint)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_product_kernel(const float* __restrict__ A, const float* __restrict__ s,
                                                           float scale, const float* __restrict__ B, int N, int J, int K,
                                                           bf16_t* __restrict__ C) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int j0 = 0; j0 < J; j0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int n = idx >> 4, j = idx & 15;
            As[j][n] = A[(size_t)(n0 + n) * J + j0 + j] * (s ? s[j0 + j] : 1.0f) * scale;
            const int jb = idx >> 6, k = idx & 63;
            Bs[jb][k] = B[(size_t)(j0 + jb) * K + k0 + k];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[j][ty * 4 + i]; b[i] = Bs[j][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[i][jj] = fmaf(a[i], b[jj], acc[i][jj]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) C[(size_t)(n0 + ty * 4 + i) * K + k0 + tx * 4 + jj] = f32_to_bf16(acc[i][jj]);
}

// one wave per output row n:  c[n] = sum_j A[n][j] s[j] scale v[j]  (f32; optional),  w[n] = sum_j float(W16[n][j])
__global__ __launch_bounds__(256) void fold_rowvec_kernel(const float* __restrict__ A, const float* __restrict__ s, float scale,
                                                          const float* __restrict__ v, const bf16_t* __restrict__ W16, int N,
                                                          int J, float* __restrict__ c_out, float* __restrict__ w_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float c = 0.f, w = 0.f;
    for (int j = lane; j < J; j += 64) {
        if (c_out) c = fmaf(A[(size_t)n * J + j] * (s ? s[j] : 1.0f) * scale, v[j], c);
        if (w_out) w += bf16_to_f32(W16[(size_t)n * J + j]);
    }
    c = wave_sum(c);
    w = wave_sum(w);
    if (lane == 0) {
        if (c_out) c_out[n] = c;
        if (w_out) w_out[n] = w;
    }
}

int cw_launch_fold_product(const float* A, const float* s, float scale, const float* B, int N, int J, int K, void* C16,
                           hipStream_t st) {
    if (N % 64 || K % 64 || J % 16) return CW_ERR_INVALID;
    hipLaunchKernelGGL(fold_product_kernel, dim3(K / 64, N / 64), dim3(256), 0, st, A, s, scale, B, N, J, K, (bf16_t*)C16);
    return CW_OK;
}
int cw_launch_fold_rowvec(const float* A, const float* s, float scale, const float* v, const void* W16, int N, int J,
                          float* c_out, float* w_out, hipStream_t st) {
    hipLaunchKernelGGL(fold_rowvec_kernel, dim3((N + 3) / 4), dim3(256), 0, st, A, s, scale, v, (const bf16_t*)W16, N, J,
                       c_out, w_out);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// gemv_stack_kernel: Mb <= 16 rows, K <= 1280 (one K slice per block), up to NT column tiles per block (per segment).
// Structure and MFMA fragment conventions are those of gemv2_bf16_kernel (gemm.hip): unconditional clamped loads, a row of
// activations lives in one wave, weights go straight from HBM into B fragments, cross-wave reduction through LDS.
// ---------------------------------------------------------------------------------------------------
#define SEG_PICK(field) (si == 0 ? p.seg[0].field : (si == 1 ? p.seg[1].field : p.seg[2].field))

template <int RPW, int NSLOT, int PER_LANE, int NT>
__global__ __launch_bounds__(256) void gemv_stack_kernel(StackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    // 17..64 rows (round 5): blockIdx.y = group of 16 rows; a group is a 16-row launch of its own over the same weights (its
    // blocks follow the first group's on the same XCD: 240 % 8 == 0, so the tile's weights come out of that L2 from the second
    // group on), per-row arithmetic unchanged -- rows m0 .. m0 + 15 of every operand, pstats one plane per group
    const int K = p.K, m0 = blockIdx.y * 16, Mb = min(16, p.Mb - m0);
    const int xs_stride = K + 8;
    bf16_t* xs = (bf16_t*)smem_s;                               // [16][K+8]
    float* red = (float*)(smem_s + (size_t)16 * xs_stride * 2);  // [4 waves][NT][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int bid = blockIdx.x;
    int si = 0;                                                  // block-uniform segment
    if (p.nseg > 1 && bid >= p.seg[1].block0) si = 1;
    if (p.nseg > 2 && bid >= p.seg[2].block0) si = 2;
    const float* __restrict__ x = SEG_PICK(x) + (size_t)m0 * K;
    const float* __restrict__ bias = SEG_PICK(bias);
    const int tile0 = SEG_PICK(tile0), n_tiles = SEG_PICK(n_tiles), block0 = SEG_PICK(block0), snt = SEG_PICK(nt);
    const int steps = K >> 7, nvec = K >> 2;
    const int tl0 = (bid - block0) * snt;                        // first tile of this block inside the segment
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;

    if (p.zero && blockIdx.y == 0) {                             // clear a buffer that a later launch accumulates into
        const int zi = bid * 256 + tid;
        if (zi < p.zero_n4) ((float4*)p.zero)[zi] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* __restrict__ wsum = SEG_PICK(wsum);
    float* smean = red + 4 * NT * 4 * 64;                       // [16] row means (segments with wsum)
    int nloc[NT];                                                // this lane's output column inside the segment (or -1)
    float bias_v[NT], wsum_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tl = tl0 + t;
        const bool ok = t < snt && tl < n_tiles;
        nloc[t] = ok ? tl * 16 + l15 : -1;
        const int ncl = (ok ? tl : n_tiles - 1) * 16 + l15;
        bias_v[t] = bias ? bias[ncl] : 0.f;
        wsum_v[t] = wsum ? wsum[ncl] : 0.f;
    }
    // activation rows -> registers
    float4 xv[RPW][PER_LANE];
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
    // weight stream: steps wave, wave+4, wave+8 (clamped: a tail wave re-reads a step another wave owns)
    u32x4_t wq[NT][NSLOT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < snt) {                                           // block-uniform: a lighter segment streams fewer tiles
            const int tl = tl0 + t < n_tiles ? tl0 + t : n_tiles - 1;
            const bf16_t* wrow = W + ((size_t)(tile0 + tl) * 16 + l15) * K + g * 8;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                int step = wave + 4 * s;
                step = step < steps ? step : steps - 1;
                const u32x4_t* wp = p.wpk ? (const u32x4_t*)(W + ((((size_t)(tile0 + tl) * (K >> 5)) + step * 4) * 64 + lane) * 8)
                                          : (const u32x4_t*)(wrow + step * 128);
                const int sj = p.wpk ? 64 : 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) wq[t][s][j] = wp[j * sj];
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSLOT; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) wq[t][s][j] = (u32x4_t){0u, 0u, 0u, 0u};
        }
    }
    // every load above is in flight before the first one is waited for: without the fence hipcc interleaves the weight loads
    // with the conversion of the activation rows as those return, which delays the weight stream by one L2 round trip
    __builtin_amdgcn_sched_barrier(0);
    // Rows that feed a LayerNorm consumer (W' x, the consumer applies rstd (. - mean W'1) + b'): rounding x itself to 16 bits
    // spends the significand on the row's mean, which the consumer subtracts again -- a factor sqrt(1 + mean^2 / var) on the
    // rounding noise of rows with an offset (outlier channels / drifting residual streams of trained checkpoints).  The row
    // lives in this wave, so its moments are a few DPP steps away: rows with |mean| >= std are rounded as x - mean and
    // mean * (W' 1) is added back in f32 in the epilogue.  Below that the factor is < 1.41 (half a bit) and the row is rounded
    // as it is: bit-identical to the uncentred stage, whose parity record the goldens hold (the rows of the synthetic models
    // sit at |mean| / std ~ 0.3-0.9, where the two roundings differ by noise only -- and that noise moves near-tie argmaxes).
    if (wsum) {                                                  // block-uniform
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < PER_LANE; ++c)
                if (lane + 64 * c < nvec) {
                    sx += (xv[i][c].x + xv[i][c].y) + (xv[i][c].z + xv[i][c].w);
                    sq += cw_sumsq4(xv[i][c].x, xv[i][c].y, xv[i][c].z, xv[i][c].w);
                }
            float mu = wave_sum(sx) / (float)K;
            if (2.f * mu * mu < wave_sum(sq) / (float)K) mu = 0.f;    // |mean| < std  <=>  2 mean^2 < E[x^2]
#pragma unroll
            for (int c = 0; c < PER_LANE; ++c) { xv[i][c].x -= mu; xv[i][c].y -= mu; xv[i][c].z -= mu; xv[i][c].w -= mu; }
            if (lane == 0) smean[wave + 4 * i] = mu;
        }
    }
    // rows -> 16 bit -> LDS
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = wave + 4 * i;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            int v4 = lane + 64 * c;
            v4 = v4 < nvec ? v4 : nvec - 1;
            const float4 v = xv[i][c];
            ushort4 o;
            o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
            *(ushort4*)(xs + (size_t)row * xs_stride + v4 * 4) = o;
        }
    }
    __syncthreads();
    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step = wave + 4 * s;
        if (step < steps) {
            const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t a = *(const bf16x8_t*)(xr + j * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16x(a, __builtin_bit_cast(bf16x8_t, wq[t][s][j]), acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    const int epi = SEG_PICK(epi);
    const int ldo = n_tiles * 16;
    float* __restrict__ out = SEG_PICK(out) + (size_t)m0 * ldo;
    float* __restrict__ out2 = SEG_PICK(out2);
    if (out2) out2 += (size_t)m0 * ldo;
    const float* __restrict__ resid = SEG_PICK(resid);
    if (resid) resid += (size_t)m0 * ldo;
    float* __restrict__ pstats = SEG_PICK(pstats);
    if (pstats) pstats += (size_t)blockIdx.y * ((n_tiles + snt - 1) / snt) * 32;   // [group][blocks of the segment][16][2]
    float ps1 = 0.f, ps2 = 0.f;                                 // this lane's share of (sum, sum of squares) of row g*4 + wave
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid >> 6;
        const float v = red[((0 * NT + t) * 4 + r) * 64 + lane] + red[((1 * NT + t) * 4 + r) * 64 + lane] +
                        red[((2 * NT + t) * 4 + r) * 64 + lane] + red[((3 * NT + t) * 4 + r) * 64 + lane];
        const int m = g * 4 + r, n = nloc[t];
        if (m < Mb && n >= 0) {
            const size_t o = (size_t)m * ldo + n;
            const float back = wsum ? smean[m] * wsum_v[t] : 0.f;
            if (epi == 0) {
                out[o] = v + bias_v[t] + back;
            } else if (epi == 2) {
                atomicAdd(out + o, v + bias_v[t] + back);
            } else {
                const float rv = resid[o] + resid_grid(v + bias_v[t]);
                out[o] = rv;
                if (out2) out2[o] = rv;
                ps1 += rv; ps2 += rv * rv;
            }
        }
    }
    if (epi == 1 && pstats) {
        // per-block LayerNorm partial sums of the rows just written (the 16 lanes of a DPP row hold the 16 columns of one
        // output row): the consumer adds the blocks' partials in a fixed order instead of re-reading the rows
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
            ps1 += sft == 0 ? dpp_mov<0x111, 0xf>(0.f, ps1) : sft == 1 ? dpp_mov<0x112, 0xf>(0.f, ps1) : sft == 2 ? dpp_mov<0x114, 0xf>(0.f, ps1) : dpp_mov<0x118, 0xf>(0.f, ps1);
            ps2 += sft == 0 ? dpp_mov<0x111, 0xf>(0.f, ps2) : sft == 1 ? dpp_mov<0x112, 0xf>(0.f, ps2) : sft == 2 ? dpp_mov<0x114, 0xf>(0.f, ps2) : dpp_mov<0x118, 0xf>(0.f, ps2);
        }
        const int m = g * 4 + (tid >> 6);
        if (l15 == 15) *(float2*)(pstats + ((size_t)(bid - block0) * 16 + m) * 2) = make_float2(ps1, ps2);
    }
}

template <int RPW, int NT>
static int launch_stack_nt(StackParams& p, hipStream_t st) {
    int blocks = 0;
    for (int s = 0; s < p.nseg; ++s) {
        if (p.seg[s].nt <= 0 || p.seg[s].nt > NT) p.seg[s].nt = NT;
        p.seg[s].block0 = blocks;
        blocks += (p.seg[s].n_tiles + p.seg[s].nt - 1) / p.seg[s].nt;
    }
    if (p.zero && (long long)blocks * 256 < p.zero_n4) return CW_ERR_INVALID;
    const size_t lds = (size_t)16 * (p.K + 8) * 2 + (size_t)4 * NT * 4 * 64 * 4 + 16 * 4;
    const dim3 grid(blocks, (p.Mb + 15) / 16);                  // y: groups of 16 rows
    if (p.K <= 256) hipLaunchKernelGGL((gemv_stack_kernel<RPW, 1, 1, NT>), grid, dim3(256), lds, st, p);
    else if (p.K <= 768) hipLaunchKernelGGL((gemv_stack_kernel<RPW, 2, 3, NT>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemv_stack_kernel<RPW, 3, 5, NT>), grid, dim3(256), lds, st, p);
    return CW_OK;
}

// nt: most 16-column tiles per block (1..3; segments may ask for fewer through StackSeg::nt); 0 = chosen so that the launch
// is one block per CU at most where possible
int cw_launch_gemv_stack(const StackParams& p_in, int nt, hipStream_t st) {
    StackParams p = p_in;
    if (p.Mb < 1 || p.Mb > 64 || p.K % 128 || p.K > 1280 || p.nseg < 1 || p.nseg > 3) return CW_ERR_INVALID;
    if (nt <= 0) {
        nt = 1;
        for (;;) {
            int blocks = 0;
            for (int s = 0; s < p.nseg; ++s) {
                const int snt = p.seg[s].nt > 0 && p.seg[s].nt < nt ? p.seg[s].nt : nt;
                blocks += (p.seg[s].n_tiles + snt - 1) / snt;
            }
            if (blocks <= 256 || nt == 3) break;
            ++nt;
        }
    }
    if (p.Mb > 8) {                                              // 9..16 rows: four rows per wave
        if (nt == 1) return launch_stack_nt<4, 1>(p, st);
        if (nt == 2) return launch_stack_nt<4, 2>(p, st);
        return launch_stack_nt<4, 3>(p, st);
    }
    if (nt == 1) return launch_stack_nt<2, 1>(p, st);
    if (nt == 2) return launch_stack_nt<2, 2>(p, st);
    return launch_stack_nt<2, 3>(p, st);
}

// ===================================================================================================
// Measured and rejected: kept as published A/Bs behind -DCW_EXPERIMENTS, not part of the default library.
// ===================================================================================================
#ifdef CW_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------
// gemv_fc2x_kernel: x3 = x2 + W2 gelu(fc1) + b2 with fc1 finished on load,
//     mid[m][k] = gelu(rstd_m (u[m][k] - mean_m w1sum[k]) + b1[k]),
// (mean, rstd) = LayerNorm statistics of the rows of `xstat`, wave-local (wave w: rows w, w + 4), shared through LDS.
// grid (N / (16 NT), F / D): one K slice of
// d_model columns per block, partial sums into the residual stream with f32 atomics (exact on the 2^-12 grid, like gemv2's K
// split).  The slice is staged column-wise -- a thread owns float4 column `tid` of all rows (NMAIN passes of 256 float4), the
// remaining columns are dealt out as (row, column) items, NTAIL per thread -- so the per-column constants are fetched once
// per block instead of once per wave.
// ---------------------------------------------------------------------------------------------------
template <int NSLOT, int NMAIN, int NTAIL, int NT>
__global__ __launch_bounds__(256) void gemv_fc2x_kernel(Fc2xParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
    const int Kb = p.D, Mb = p.Mb, K = p.F, N = p.D;
    const int xs_stride = Kb + 8;
    bf16_t* xs = (bf16_t*)smem_f;
    float* red = (float*)(smem_f + (size_t)16 * xs_stride * 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int kbase = blockIdx.y * Kb;
    const int steps = Kb >> 7, nvec = Kb >> 2;
    const int rem = nvec - NMAIN * 256;                          // float4 columns beyond the main passes (0 .. 255)
    const bf16_t* __restrict__ W = (const bf16_t*)p.W2;
    int nn[NT], ncl[NT];
    float bias_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        nn[t] = n0 + t * 16 + l15;
        ncl[t] = nn[t] < N ? nn[t] : N - 1;
        bias_v[t] = p.b2 ? p.b2[ncl[t]] : 0.f;
    }
    // LayerNorm statistics of the residual rows: wave w takes rows w and w + 4 (requested first: they come back first)
    __shared__ float s_stat[16];
    float4 sv[2][5];
    const int dvec = p.D >> 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = min(wave + 4 * i, Mb - 1);
#pragma unroll
        for (int c = 0; c < 5; ++c) sv[i][c] = *(const float4*)(p.xstat + (size_t)row * p.D + (size_t)min(lane + 64 * c, dvec - 1) * 4);
    }
    float4 um[NMAIN > 0 ? NMAIN : 1][8], w1m[NMAIN > 0 ? NMAIN : 1], b1m[NMAIN > 0 ? NMAIN : 1];
#pragma unroll
    for (int q = 0; q < NMAIN; ++q) {
        const int c4 = q * 256 + tid;
        w1m[q] = *(const float4*)(p.w1sum + kbase + c4 * 4);
        b1m[q] = *(const float4*)(p.b1 + kbase + c4 * 4);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int mc = m < Mb ? m : Mb - 1;
            um[q][m] = *(const float4*)(p.u + (size_t)mc * K + kbase + c4 * 4);
        }
    }
    float4 ut[NTAIL > 0 ? NTAIL : 1], w1t[NTAIL > 0 ? NTAIL : 1], b1t[NTAIL > 0 ? NTAIL : 1];
    int trow[NTAIL > 0 ? NTAIL : 1], tc4[NTAIL > 0 ? NTAIL : 1];
#pragma unroll
    for (int i = 0; i < NTAIL; ++i) {
        int j = tid + 256 * i;
        const int nitem = 8 * rem;
        j = j < nitem ? j : nitem - 1;                          // clamped items rewrite identical data
        trow[i] = j / rem;
        tc4[i] = NMAIN * 256 + j % rem;
        const int mc = trow[i] < Mb ? trow[i] : Mb - 1;
        w1t[i] = *(const float4*)(p.w1sum + kbase + tc4[i] * 4);
        b1t[i] = *(const float4*)(p.b1 + kbase + tc4[i] * 4);
        ut[i] = *(const float4*)(p.u + (size_t)mc * K + kbase + tc4[i] * 4);
    }
    u32x4_t wq[NT][NSLOT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16_t* wrow = W + (size_t)ncl[t] * K + kbase + g * 8;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            int step = wave + 4 * s;
            step = step < steps ? step : steps - 1;
            const u32x4_t* wp = p.wpk ? (const u32x4_t*)(W + ((((size_t)(ncl[t] >> 4) * (K >> 5)) + (kbase >> 5) + step * 4) * 64 + g * 16 + (ncl[t] & 15)) * 8)
                                      : (const u32x4_t*)(wrow + step * 128);
            const int sj = p.wpk ? 64 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) wq[t][s][j] = wp[j * sj];
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // all loads issued before the first wait (see gemv_stack_kernel)
    float mean[8], rstd[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float ok = (lane + 64 * c < dvec) ? 1.f : 0.f;
            sx += ok * ((sv[i][c].x + sv[i][c].y) + (sv[i][c].z + sv[i][c].w));
        }
        const float mu = wave_sum(sx) / (float)p.D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float ok = (lane + 64 * c < dvec) ? 1.f : 0.f;
            const float a = sv[i][c].x - mu, b = sv[i][c].y - mu, cc = sv[i][c].z - mu, d = sv[i][c].w - mu;
            sq += ok * cw_sumsq4(a, b, cc, d);
        }
        const float rs = 1.0f / sqrtf(wave_sum(sq) / (float)p.D + 1e-5f);
        if (lane == 0) { s_stat[2 * (wave + 4 * i)] = mu; s_stat[2 * (wave + 4 * i) + 1] = rs; }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) { mean[m] = s_stat[2 * m]; rstd[m] = s_stat[2 * m + 1]; }
#define FC2X_MID(uu, ww, bb, mu, rs) f32_to_bf16(gelu_fast(((uu) - (mu) * (ww)) * (rs) + (bb)))
#pragma unroll
    for (int q = 0; q < NMAIN; ++q) {
        const int c4 = q * 256 + tid;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            ushort4 o;
            o.x = FC2X_MID(um[q][m].x, w1m[q].x, b1m[q].x, mean[m], rstd[m]);
            o.y = FC2X_MID(um[q][m].y, w1m[q].y, b1m[q].y, mean[m], rstd[m]);
            o.z = FC2X_MID(um[q][m].z, w1m[q].z, b1m[q].z, mean[m], rstd[m]);
            o.w = FC2X_MID(um[q][m].w, w1m[q].w, b1m[q].w, mean[m], rstd[m]);
            *(ushort4*)(xs + (size_t)m * xs_stride + c4 * 4) = o;
        }
    }
#pragma unroll
    for (int i = 0; i < NTAIL; ++i) {
        float mu = mean[0], rs = rstd[0];
#pragma unroll
        for (int m = 1; m < 8; ++m) if (trow[i] == m) { mu = mean[m]; rs = rstd[m]; }
        ushort4 o;
        o.x = FC2X_MID(ut[i].x, w1t[i].x, b1t[i].x, mu, rs);
        o.y = FC2X_MID(ut[i].y, w1t[i].y, b1t[i].y, mu, rs);
        o.z = FC2X_MID(ut[i].z, w1t[i].z, b1t[i].z, mu, rs);
        o.w = FC2X_MID(ut[i].w, w1t[i].w, b1t[i].w, mu, rs);
        *(ushort4*)(xs + (size_t)trow[i] * xs_stride + tc4[i] * 4) = o;
    }
#undef FC2X_MID
    __syncthreads();
    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step = wave + 4 * s;
        if (step < steps) {
            const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t a = *(const bf16x8_t*)(xr + j * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16x(a, __builtin_bit_cast(bf16x8_t, wq[t][s][j]), acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid >> 6;
        const float v = red[((0 * NT + t) * 4 + r) * 64 + lane] + red[((1 * NT + t) * 4 + r) * 64 + lane] +
                        red[((2 * NT + t) * 4 + r) * 64 + lane] + red[((3 * NT + t) * 4 + r) * 64 + lane];
        const int m = g * 4 + r, n = nn[t];
        if (m < Mb && n < N) atomicAdd(p.x + (size_t)m * N + n, resid_grid(v + (blockIdx.y == 0 ? bias_v[t] : 0.f)));
    }
}

template <int NSLOT, int NMAIN, int NTAIL>
static void launch_fc2x_shape(const Fc2xParams& p, bool two, hipStream_t st) {
    const int nt = two ? 2 : 1;
    const int tiles = (p.D + 16 * nt - 1) / (16 * nt);
    dim3 grid(tiles, p.F / p.D);
    const size_t lds = (size_t)16 * (p.D + 8) * 2 + (size_t)4 * nt * 4 * 64 * 4;
    if (two) hipLaunchKernelGGL((gemv_fc2x_kernel<NSLOT, NMAIN, NTAIL, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemv_fc2x_kernel<NSLOT, NMAIN, NTAIL, 1>), grid, dim3(256), lds, st, p);
}

int cw_launch_gemv_fc2x(const Fc2xParams& p, hipStream_t st) {
    if (p.Mb < 1 || p.Mb > 8 || p.D % 128 || p.D > 1280 || p.F % p.D) return CW_ERR_INVALID;
    const int ks = p.F / p.D;
    const bool two = p.D % 32 == 0 && (p.D / 16) * ks > 256 && (p.D / 32) * ks >= 128;   // (80, 4) -> (40, 4), as gemv2
    // float4 columns of a K slice: nvec = D / 4 = NMAIN * 256 + rem;  the rem columns x 8 rows are NTAIL items per thread
    switch (p.D) {
        case 128:  launch_fc2x_shape<1, 0, 1>(p, two, st); break;    // nvec 32
        case 256:  launch_fc2x_shape<1, 0, 2>(p, two, st); break;    // 64
        case 384:  launch_fc2x_shape<1, 0, 3>(p, two, st); break;    // 96
        case 512:  launch_fc2x_shape<1, 0, 4>(p, two, st); break;    // 128
        case 640:  launch_fc2x_shape<2, 0, 5>(p, two, st); break;    // 160
        case 768:  launch_fc2x_shape<2, 0, 6>(p, two, st); break;    // 192
        case 896:  launch_fc2x_shape<2, 0, 7>(p, two, st); break;    // 224
        case 1024: launch_fc2x_shape<2, 1, 0>(p, two, st); break;    // 256
        case 1152: launch_fc2x_shape<3, 1, 1>(p, two, st); break;    // 288
        case 1280: launch_fc2x_shape<3, 1, 2>(p, two, st); break;    // 320
        default: return CW_ERR_INVALID;
    }
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// mlp_pair_kernel: LN + fc1 + GELU, group barrier, fc2 + residual in one launch (grid F / 32, 256 threads; every block must be
// resident at once: F / 32 <= CUs).  Block p computes fc1 columns [32 p, 32 p + 32) -- they lie in K slice kq = p / (D / 32) of
// fc2 -- writes them as 16-bit values with agent-scope (write-through) stores and arrives at its group's barrier; once the
// D / 32 blocks of the group are in, the slice mid[:, kq D .. (kq + 1) D) is complete and block p continues as fc2 block
// (column pair p % (D / 32), K slice kq) exactly like gemv2's K-split launch.  fc2's 80 KB of weights per block are requested at
// kernel entry, behind fc1's: they stream while fc1 computes and the group gathers, which is what a second launch cannot do.
// The barrier is a sense-reversing arrival counter per group (reusable without re-initialisation: graph replays carry no
// per-launch arguments); spins are bounded and report through p.err instead of hanging.
// MEASURED AND REJECTED (round 3, MI355X, B = 8, profiles/r03_c_*): 23.3 us per launch against 6.5 + 6.2 us for the two launches
// it replaces -- correct (parity tests green), but an arrival barrier among 40 blocks spread over 8 XCDs, the write-through
// hand-over of the 20 KB slice and its agent-scope re-read cost more than a kernel boundary (~2 us) does.  Kept behind
// CW_MLP_PAIR=1 as the A/B for "replace a boundary by an in-kernel barrier"; the decode step does not use it.
// ---------------------------------------------------------------------------------------------------
#define MLP_SPIN_LIMIT 400000
template <int NSLOT, int PER_LANE>
__global__ __launch_bounds__(256) void mlp_pair_kernel(MlpPairParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
    __shared__ int s_ok;
    constexpr int NT = 2, RPW = 2;
    const int D = p.D, F = p.F, Mb = p.Mb;
    const int xs_stride = D + 8;
    bf16_t* xs = (bf16_t*)smem_m;                               // [16][D+8]: LayerNorm(x) for fc1, then the mid slice for fc2
    float* red = (float*)(smem_m + (size_t)16 * xs_stride * 2);  // [4 waves][NT][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int per_group = D / 32;
    const int kq = blockIdx.x / per_group, gi = blockIdx.x - kq * per_group;
    const int steps = D >> 7, nvec = D >> 2;
    const bf16_t* __restrict__ W1 = (const bf16_t*)p.W1;
    const bf16_t* __restrict__ W2 = (const bf16_t*)p.W2;
    const int n1 = blockIdx.x * 32;                              // fc1 columns of this block
    const int n2 = gi * 32;                                      // fc2 columns
    const int kbase = kq * D;                                    // fc2 K slice

    float b1v[NT], b2v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { b1v[t] = p.b1[n1 + t * 16 + l15]; b2v[t] = p.b2[n2 + t * 16 + l15]; }
    float4 xv[RPW][PER_LANE];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = min(wave + 4 * i, Mb - 1);
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) xv[i][c] = *(const float4*)(p.x + (size_t)row * D + (size_t)min(lane + 64 * c, nvec - 1) * 4);
    }
    u32x4_t w1q[NT][NSLOT][4], w2q[NT][NSLOT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16_t* wrow = W1 + (size_t)(n1 + t * 16 + l15) * D + g * 8;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int step = min(wave + 4 * s, steps - 1);
            const u32x4_t* wp = p.wpk ? (const u32x4_t*)(W1 + ((((size_t)((n1 >> 4) + t) * (D >> 5)) + step * 4) * 64 + lane) * 8)
                                      : (const u32x4_t*)(wrow + step * 128);
            const int sj = p.wpk ? 64 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) w1q[t][s][j] = wp[j * sj];
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16_t* wrow = W2 + (size_t)(n2 + t * 16 + l15) * F + kbase + g * 8;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int step = min(wave + 4 * s, steps - 1);
            const u32x4_t* wp = p.wpk ? (const u32x4_t*)(W2 + ((((size_t)((n2 >> 4) + t) * (F >> 5)) + (kbase >> 5) + step * 4) * 64 + lane) * 8)
                                      : (const u32x4_t*)(wrow + step * 128);
            const int sj = p.wpk ? 64 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) w2q[t][s][j] = wp[j * sj];
        }
    }
    __builtin_amdgcn_sched_barrier(0);                          // every load is out before the first wait
    // ---- fc1: wave-local LayerNorm (gamma / beta folded into W1 / b1), rows -> 16 bit -> LDS
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            sx += ok * ((xv[i][c].x + xv[i][c].y) + (xv[i][c].z + xv[i][c].w));
        }
        const float mean = wave_sum(sx) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            const float a = xv[i][c].x - mean, b = xv[i][c].y - mean, cc = xv[i][c].z - mean, d = xv[i][c].w - mean;
            sq += ok * cw_sumsq4(a, b, cc, d);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
        const int row = wave + 4 * i;
#pragma unroll
        for (int c = 0; c < PER_LANE; ++c) {
            const int v4 = min(lane + 64 * c, nvec - 1);
            ushort4 o;
            o.x = f32_to_bf16((xv[i][c].x - mean) * rstd); o.y = f32_to_bf16((xv[i][c].y - mean) * rstd);
            o.z = f32_to_bf16((xv[i][c].z - mean) * rstd); o.w = f32_to_bf16((xv[i][c].w - mean) * rstd);
            *(ushort4*)(xs + (size_t)row * xs_stride + v4 * 4) = o;
        }
    }
    __syncthreads();
    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step = wave + 4 * s;
        if (step < steps) {
            const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t a = *(const bf16x8_t*)(xr + j * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16x(a, __builtin_bit_cast(bf16x8_t, w1q[t][s][j]), acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    bf16_t* mid = (bf16_t*)p.mid;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid >> 6;
        const float v = red[((0 * NT + t) * 4 + r) * 64 + lane] + red[((1 * NT + t) * 4 + r) * 64 + lane] +
                        red[((2 * NT + t) * 4 + r) * 64 + lane] + red[((3 * NT + t) * 4 + r) * 64 + lane];
        const unsigned int h = f32_to_bf16(gelu_erf(v + b1v[t]));
        const unsigned int other = (unsigned int)__shfl_xor((int)h, 1, 64);        // neighbouring column
        const int m = g * 4 + r, n = n1 + t * 16 + l15;
        if (m < Mb && !(l15 & 1))                               // two columns per 4-byte write-through store
            __hip_atomic_store((unsigned int*)(mid + (size_t)m * F + n), h | (other << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- group barrier: the stores above have completed (and so have the fc2 weight loads) before the block arrives
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned int* cnt = p.bar + 2 * kq;
        unsigned int* gen = p.bar + 2 * kq + 1;
        const unsigned int g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (ticket == (unsigned int)per_group - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, g0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int it = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
                __builtin_amdgcn_s_sleep(1);
                if (++it > MLP_SPIN_LIMIT) { ok = 0; break; }
            }
        }
        if (!ok) *p.err = 1;
        s_ok = ok;
    }
    __syncthreads();
    // no acquire fence: the slice is read with sc1 loads below, which are coherent at the memory side on their own; an agent-scope
    // acquire here is an L2 invalidate (round 4, profiles/r04_grid_barrier.txt: the fences are 3 of a barrier's 5 us)
    if (p.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // ---- fc2: the group's mid slice [Mb][D] -> LDS (agent-scope loads: written by blocks on other XCDs)
    {
        const unsigned long long* src = (const unsigned long long*)mid;
        const int n8 = D >> 2;                                   // 8-byte words per row of the slice
        for (int idx = tid; idx < 8 * n8; idx += 256) {
            const int row = idx / n8, w = idx - row * n8;
            const int rc = row < Mb ? row : Mb - 1;
            const unsigned long long v = __hip_atomic_load((unsigned long long*)src + ((size_t)rc * F + kbase) / 4 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(unsigned long long*)(xs + (size_t)row * xs_stride + w * 4) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step = wave + 4 * s;
        if (step < steps) {
            const bf16_t* xr = xs + (size_t)l15 * xs_stride + step * 128 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8_t a = *(const bf16x8_t*)(xr + j * 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma16x(a, __builtin_bit_cast(bf16x8_t, w2q[t][s][j]), acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid >> 6;
        const float v = red[((0 * NT + t) * 4 + r) * 64 + lane] + red[((1 * NT + t) * 4 + r) * 64 + lane] +
                        red[((2 * NT + t) * 4 + r) * 64 + lane] + red[((3 * NT + t) * 4 + r) * 64 + lane];
        const int m = g * 4 + r, n = n2 + t * 16 + l15;
        if (m < Mb) atomicAdd(p.x + (size_t)m * D + n, resid_grid(v + (kq == 0 ? b2v[t] : 0.f)));
    }
}

int cw_launch_mlp_pair(const MlpPairParams& p, hipStream_t st) {
    if (p.Mb < 1 || p.Mb > 8 || p.D % 128 || p.D > 1280 || p.F % p.D || p.F / 32 > 256 || !p.bar || !p.err || !p.mid) return CW_ERR_INVALID;
    const size_t lds = (size_t)16 * (p.D + 8) * 2 + (size_t)4 * 2 * 4 * 64 * 4;
    const dim3 grid(p.F / 32);
    if (p.D <= 256) hipLaunchKernelGGL((mlp_pair_kernel<1, 1>), grid, dim3(256), lds, st, p);
    else if (p.D <= 768) hipLaunchKernelGGL((mlp_pair_kernel<2, 3>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((mlp_pair_kernel<3, 5>), grid, dim3(256), lds, st, p);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// gemv_rows_kernel: 17..64 rows (kernels.h: RowsParams).  grid N / 16, NW waves; wave w takes the 128-wide K steps w, w + NW, ..
// of the WHOLE K (no K split, no atomics): all of the block's weights are requested before the first wait (NSLOT x 4
// fragments of 16 B per lane), A fragments come straight from the fragment-major activations (L2 resident), the waves'
// partial tiles meet in LDS.
// ---------------------------------------------------------------------------------------------------
template <int EPI, int MT, int NSLOT, int NW, bool PRODUCE, bool HILO>
__global__ __launch_bounds__(NW * 64) void gemv_rows_kernel(RowsParams p) {
    __shared__ float red[NW * MT * 4 * 64];
    __shared__ float s_st[PRODUCE ? 1 : NW * 64 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int K = p.K, N = p.N;
    const int n0 = blockIdx.x * 16;
    const int steps = K >> 7, KS = K >> 5;
    const int n = n0 + l15;
    const int nc = n < N ? n : N - 1;
    const bf16_t* W = (const bf16_t*)p.W;
    const bool ln = !PRODUCE && p.ln_pstats != nullptr;
    // consumer: this thread's share of the LayerNorm partial sums (row = lane, blocks wave, wave + NW, ..), requested first
    float s1 = 0.f, s2 = 0.f;
    if (ln) {                                                  // all requests out before the first use (a rolled loop serialises them)
        constexpr int PER = 80 / NW;                           // <= 80 producer blocks (d_model <= 1280)
        float2 t[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) t[i] = *(const float2*)(p.ln_pstats + ((size_t)min(wave + NW * i, p.ln_nblk - 1) * 64 + lane) * 2);
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (wave + NW * i < p.ln_nblk) { s1 += t[i].x; s2 += t[i].y; }
    }
    u32x4_t wq[NSLOT][4];
    const bf16_t* wrow = W + (size_t)nc * K + g * 8;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        int step = wave + NW * s;
        step = step < steps ? step : steps - 1;               // clamped (unconditional) load, zeroed below
        const u32x4_t* wp = p.wpk ? (const u32x4_t*)(W + ((((size_t)(nc >> 4) * KS) + step * 4) * 64 + g * 16 + (nc & 15)) * 8)
                                  : (const u32x4_t*)(wrow + step * 128);
        const int sj = p.wpk ? 64 : 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[s][j] = wp[j * sj];
    }
    const float bias_v = p.ep.bias ? p.ep.bias[nc] : 0.f;
    const float wsum_v = ln ? p.ln_wsum[nc] : 0.f;
    if (ln) { s_st[(wave * 64 + lane) * 2] = s1; s_st[(wave * 64 + lane) * 2 + 1] = s2; }
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const u32x4_t* xq = (const u32x4_t*)p.xf;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int step_raw = wave + NW * s;
        const bool live = step_raw < steps;                    // wave-uniform
        const int step = live ? step_raw : steps - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4_t w = wq[s][j];
            if (!live) w = (u32x4_t){0u, 0u, 0u, 0u};          // a dead slot contributes exactly zero
            const int ks = step * 4 + j;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const u32x4_t a = xq[((size_t)t * KS + ks) * 64 + lane];
                if (HILO) {                                     // low halves first: the small terms meet before the large ones
                    const u32x4_t al = xq[(size_t)p.lo_off + ((size_t)t * KS + ks) * 64 + lane];
                    acc[t] = mfma16x(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, w), acc[t]);
                }
                acc[t] = mfma16x(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, w), acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * MT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    const float inv_k = 1.0f / (float)K;
    for (int idx = wave; idx < 4 * MT; idx += NW) {           // (row tile t, row r of the lane's group) pairs over the waves
        const int t = idx >> 2, r = idx & 3;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[((w * MT + t) * 4 + r) * 64 + lane];
        const int m = t * 16 + g * 4 + r;
        const bool live = m < p.Mb && n < N;
        if (!PRODUCE) {
            if (ln) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { a1 += s_st[(w * 64 + m) * 2]; a2 += s_st[(w * 64 + m) * 2 + 1]; }
                const float mean = a1 * inv_k;
                const float rstd = 1.0f / sqrtf(fmaxf(a2 * inv_k - mean * mean, 0.f) + 1e-5f);
                v = (v - mean * wsum_v) * rstd;
            }
            if (live) {
                EpiParams e2 = p.ep;
                e2.bias = nullptr;
                epi_store1<bf16_t, EPI>(e2, m, n, v + bias_v);
            }
        } else {
            float xn = 0.f;
            if (live) {
                const size_t o = (size_t)m * p.ep.ldo + n;
                xn = p.ep.resid[o] + (v + bias_v);
                p.ep.outf[o] = xn;
                const bf16_t hi = f32_to_bf16(xn);
                ((bf16_t*)p.xf_out)[frag_index(m, n, N)] = hi;
                if (p.lo_off) ((bf16_t*)p.xf_out)[(size_t)p.lo_off * 8 + frag_index(m, n, N)] = f32_to_bf16(xn - bf16_to_f32(hi));
            }
            float ps1 = xn, ps2 = xn * xn;                      // over the block's 16 columns: lane 15 of each row of lanes
            ps1 += dpp_mov<0x111, 0xf>(0.f, ps1); ps2 += dpp_mov<0x111, 0xf>(0.f, ps2);
            ps1 += dpp_mov<0x112, 0xf>(0.f, ps1); ps2 += dpp_mov<0x112, 0xf>(0.f, ps2);
            ps1 += dpp_mov<0x114, 0xf>(0.f, ps1); ps2 += dpp_mov<0x114, 0xf>(0.f, ps2);
            ps1 += dpp_mov<0x118, 0xf>(0.f, ps1); ps2 += dpp_mov<0x118, 0xf>(0.f, ps2);
            if (l15 == 15 && m < p.Mb) *(float2*)(p.pstats_out + ((size_t)blockIdx.x * 64 + m) * 2) = make_float2(ps1, ps2);
        }
    }
}

// first rows of a decode step (token + position embedding, f32): 16-bit fragment-major copy + whole-row sums as ONE partial
__global__ __launch_bounds__(256) void rows_prep_kernel(const float* __restrict__ x, int K, bf16_t* __restrict__ xf,
                                                        float* __restrict__ pstats, int lo_off) {
    __shared__ float s_red[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
        const float4 v = *(const float4*)(x + (size_t)m * K + k);
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        ushort4 o;
        o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
        *(ushort4*)(xf + frag_index(m, k, K)) = o;
        if (lo_off) {
            ushort4 l;
            l.x = f32_to_bf16(v.x - bf16_to_f32(o.x)); l.y = f32_to_bf16(v.y - bf16_to_f32(o.y));
            l.z = f32_to_bf16(v.z - bf16_to_f32(o.z)); l.w = f32_to_bf16(v.w - bf16_to_f32(o.w));
            *(ushort4*)(xf + (size_t)lo_off * 8 + frag_index(m, k, K)) = l;
        }
    }
    s1 = block_sum(s1, s_red);
    s2 = block_sum(s2, s_red);
    if (tid == 0) *(float2*)(pstats + (size_t)m * 2) = make_float2(s1, s2);
}

int cw_launch_rows_prep(const float* x, int Mb, int K, void* xf, float* pstats, int lo_off, hipStream_t st) {
    if (Mb < 1 || Mb > 64 || K % 32) return CW_ERR_INVALID;
    hipLaunchKernelGGL(rows_prep_kernel, dim3(Mb), dim3(256), 0, st, x, K, (bf16_t*)xf, pstats, lo_off);
    return CW_OK;
}

template <int EPI, int MT, bool PRODUCE>
static int launch_rows_mt(const RowsParams& p, hipStream_t st) {
    const int steps = p.K / 128;
    const dim3 grid((p.N + 15) / 16);
    const bool hilo = !PRODUCE && p.lo_off != 0;                 // consumers of a residual copy carried as hi + lo halves
#define CW_ROWS(NS, NW)                                                                                                 \
    do {                                                                                                                \
        if (hilo) hipLaunchKernelGGL((gemv_rows_kernel<EPI, MT, NS, NW, PRODUCE, !PRODUCE>), grid, dim3(NW * 64), 0, st, p); \
        else hipLaunchKernelGGL((gemv_rows_kernel<EPI, MT, NS, NW, PRODUCE, false>), grid, dim3(NW * 64), 0, st, p);    \
    } while (0)
    if (steps <= 4) CW_ROWS(1, 4);
    else if (steps <= 8) CW_ROWS(2, 4);
    else if (steps <= 12) CW_ROWS(3, 4);
    else if (steps <= 16) CW_ROWS(2, 8);
    else if (steps <= 24) CW_ROWS(3, 8);
    else if (steps <= 32) CW_ROWS(4, 8);
    else if (steps <= 40) CW_ROWS(5, 8);
    else return CW_ERR_INVALID;
#undef CW_ROWS
    return CW_OK;
}
template <int EPI, bool PRODUCE>
static int launch_rows(const RowsParams& p, hipStream_t st) {
    const int MT = (p.Mb + 15) / 16;
    if (MT == 2) return launch_rows_mt<EPI, 2, PRODUCE>(p, st);
    if (MT == 3) return launch_rows_mt<EPI, 3, PRODUCE>(p, st);
    return launch_rows_mt<EPI, 4, PRODUCE>(p, st);
}

int cw_launch_gemv_rows(int epi, bool produce, const RowsParams& p, hipStream_t st) {
    if (p.Mb <= 16 || p.Mb > 64 || p.K % 128 || p.K > 5120 || p.N < 1 || !p.xf || !p.W) return CW_ERR_INVALID;
    if (produce) {
        if (epi != EPI_RESID_F32 || !p.xf_out || !p.pstats_out || !p.ep.outf || !p.ep.resid || p.N % 16) return CW_ERR_INVALID;
        return launch_rows<EPI_RESID_F32, true>(p, st);
    }
    if (p.ln_pstats && (!p.ln_wsum || p.ln_nblk < 1 || p.ln_nblk > 80)) return CW_ERR_INVALID;
    switch (epi) {
        case EPI_STORE_F32: return launch_rows<EPI_STORE_F32, false>(p, st);
        case EPI_QKV_CACHE: return launch_rows<EPI_QKV_CACHE, false>(p, st);
        case EPI_GELU_FRAG: return launch_rows<EPI_GELU_FRAG, false>(p, st);
        default: return CW_ERR_INVALID;
    }
}

#else   // !CW_EXPERIMENTS: the measured-and-rejected stages are not in the library (make EXTRA=-DCW_EXPERIMENTS builds them)
int cw_launch_gemv_fc2x(const Fc2xParams&, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_mlp_pair(const MlpPairParams&, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_rows_prep(const float*, int, int, void*, float*, int, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_gemv_rows(int, bool, const RowsParams&, hipStream_t) { return CW_ERR_INVALID; }
#endif

}  // namespace CW_NS
