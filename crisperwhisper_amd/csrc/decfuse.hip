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

// UNI (round 6): every segment streams NT tiles per block from fragment-major weights -- the shape of every launch of the default
// engine.  The weight requests then sit under no branch (`t < snt` and `wpk` are run-time values otherwise: hipcc wraps each tile's
// loads in a block-uniform branch and must answer the first wait behind them with vmcnt(0)).  Same arithmetic.
template <int RPW, int NSLOT, int PER_LANE, int NT, bool UNI = false>
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
    // the residual elements the x1 segment's epilogue adds to (epi == 1), requested at the head of the queue: at their use they are a
    // dependent load behind the MFMA loop -- an L2 round trip on the tail of the blocks that end the launch.  Unconditional (other
    // segments read an element of their own input instead: a load under a branch would make every later counted wait conservative)
    float resid_pre[NT];
    {
        const float* rsd = SEG_PICK(resid);
        const int ldo_ = n_tiles * 16, mr = min(g * 4 + (tid >> 6), Mb - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nl = nloc[t] >= 0 ? nloc[t] : 0;
            const float* rp = rsd ? rsd + (size_t)(m0 + mr) * ldo_ + nl : x;
            resid_pre[t] = *rp;
        }
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
        if (UNI || t < snt) {                                    // block-uniform: a lighter segment streams fewer tiles
            const int tl = tl0 + t < n_tiles ? tl0 + t : n_tiles - 1;
            const bf16_t* wrow = W + ((size_t)(tile0 + tl) * 16 + l15) * K + g * 8;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                int step = wave + 4 * s;
                step = step < steps ? step : steps - 1;
                const bool wpk = UNI || p.wpk;
                const u32x4_t* wp = wpk ? (const u32x4_t*)(W + ((((size_t)(tile0 + tl) * (K >> 5)) + step * 4) * 64 + lane) * 8)
                                        : (const u32x4_t*)(wrow + step * 128);
                const int sj = wpk ? 64 : 4;
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
                const float rv = resid_pre[t] + resid_grid(v + bias_v[t]);
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
    else {
        bool uni = p.wpk != 0;
        for (int s = 0; s < p.nseg; ++s) uni = uni && p.seg[s].nt == NT;
        if (uni) hipLaunchKernelGGL((gemv_stack_kernel<RPW, 3, 5, NT, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((gemv_stack_kernel<RPW, 3, 5, NT>), grid, dim3(256), lds, st, p);
    }
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
#include "decfuse_experiments.inc"   // 600 lines of published A/B kernels, outside the default build
#else   // !CW_EXPERIMENTS: the measured-and-rejected stages are not in the library (make EXTRA=-DCW_EXPERIMENTS builds them)
int cw_launch_gemv_fc2x(const Fc2xParams&, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_mlp_pair(const MlpPairParams&, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_rows_prep(const float*, int, int, void*, float*, int, hipStream_t) { return CW_ERR_INVALID; }
int cw_launch_gemv_rows(int, bool, const RowsParams&, hipStream_t) { return CW_ERR_INVALID; }
#endif

}  // namespace CW_NS
