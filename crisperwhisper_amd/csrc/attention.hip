// Attention kernels (head_dim = 64 for every Whisper size).
//
//   attn_encoder_q64    (default) 64 queries per wave, LDS-DMA staging, transpose reads for V, q-tile software pipeline --
//                       see the comment block in front of it; attn_encoder_bf16 below is the first-generation kernel it replaced
//                       (kept for A/B runs, CW_ATTN_V1=1).
//   attn_encoder_bf16   flash-style fused softmax(Q K^T) V over S = 1500 frames, no mask
//                       (TF/models/whisper/modeling_whisper.py:215-238, 284-356; q is pre-scaled by
//                       the projection).  Never materialises the S x S map the HF eager path keeps.
//                       Computes S^T = K Q^T with MFMA 16x16x32 so that each lane owns one query column:
//                       softmax statistics are lane-local + 2 cross-lane shuffles, the probabilities are
//                       already in B-operand layout for O^T = V^T P^T (no LDS round trip for P), and the
//                       running rescale of O is a per-lane scalar.
//   attn_encoder_f32    straightforward f32 version (parity mode + on-device reference).
//   attn_decode         one query per (batch, head) against a KV cache (decoder self-attention with
//                       n_keys = t+1, cross-attention with n_keys = 1500).  HBM-bound: coalesced 16-byte
//                       loads, 8 lanes per key row.  For alignment heads the normalised probability row
//                       is written straight into the [B, H_a, L, S] alignment buffer consumed by the DTW
//                       stage (replaces HF's per-step retention of all 32x20 heads,
//                       TF/generation/utils.py:2909-2910).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <mutex>

namespace CW_NS {

__device__ inline f32x4_t mfma16a(bf16x8_t a, bf16x8_t b, f32x4_t c) { return cw_mfma_16x16x32(a, b, c); }

#define KT 64          // keys per LDS tile
#define RESCALE_THR 8.0f
#define KS_STRIDE 72   // bf16 per K row in LDS (64 + 8)
#define VS_STRIDE 66   // bf16 per V^T row in LDS (64 keys + 2): 33 dwords -> spreads the transposing writes

#ifdef CW_EXPERIMENTS   // round-1 encoder attention (32 queries per wave), A/B only (CW_ATTN_V1=1)
// grid: (ceil(S/128), H, B), 256 threads; wave w owns queries q0 + w*32 .. +31 (two 16-query tiles).
__global__ __launch_bounds__(256) void attn_encoder_bf16_kernel(const bf16_t* __restrict__ Q,
                                                                const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V,
                                                                bf16_t* __restrict__ out, int H, int S, int S_pad) {
    __shared__ __attribute__((aligned(16))) bf16_t sK[KT * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t sVt[64 * VS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware: the q-blocks of one (batch, head) are consecutive logical ids and therefore share
    // an XCD's L2 for their K/V re-reads (round-robin placement re-fetched K/V once per XCD: 5.7x over-fetch).
    const int nqb = (S + 127) / 128;
    int lid;
    {
        const int nwg = gridDim.x, nx = 8, xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
        const int q = nwg / nx, r = nwg % nx;
        lid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / H, h = bh - b * H;
    const size_t head_off = ((size_t)b * H + h) * S_pad * 64;
    const bf16_t* Qh = Q + head_off;
    const bf16_t* Kh = K + head_off;
    const bf16_t* Vh = V + head_off;
    const int qbase = qblk * 128 + wave * 32;

    // Q fragments (B operand of S^T = K Q^T): lane supplies Q[q = l15][d = kk*32 + g*8 .. +7]
    bf16x8_t fq[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = qbase + qt * 16 + l15;
        if (q >= S) q = S - 1;  // clamp: rows beyond S are computed but never stored
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fq[qt][kk] = *(const bf16x8_t*)(Qh + (size_t)q * 64 + kk * 32 + g * 8);
    }

    f32x4_t o[2][4];
    float mrow[2], lrow[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        mrow[qt] = -INFINITY; lrow[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (S + KT - 1) / KT;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int k0 = tile * KT;
        __syncthreads();  // previous tile fully consumed
        // stage K tile: 64 keys x 64 d = 512 16-byte chunks, 2 per thread (rows < S_pad are allocated/zeroed)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int c = tid + i * 256, row = c >> 3, col = (c & 7) * 8;
            uint4 kv = *(const uint4*)(Kh + (size_t)(k0 + row) * 64 + col);
            *(uint4*)(sK + row * KS_STRIDE + col) = kv;
            // V: transpose while staging -> sVt[d][key]
            uint4 vv = *(const uint4*)(Vh + (size_t)(k0 + row) * 64 + col);
            const bf16_t* ve = (const bf16_t*)&vv;
#pragma unroll
            for (int e = 0; e < 8; ++e) sVt[(col + e) * VS_STRIDE + row] = ve[e];
        }
        __syncthreads();

        // S^T tiles: st[qt][kt][r] = score(key = k0 + kt*16 + g*4 + r, query = qbase + qt*16 + l15)
        f32x4_t st[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16x8_t fk0 = *(const bf16x8_t*)(sK + (kt * 16 + l15) * KS_STRIDE + g * 8);
            bf16x8_t fk1 = *(const bf16x8_t*)(sK + (kt * 16 + l15) * KS_STRIDE + 32 + g * 8);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                z = mfma16a(fk0, fq[qt][0], z);
                z = mfma16a(fk1, fq[qt][1], z);
                st[qt][kt] = z;
            }
        }
        // mask keys beyond S (only the last tile), online softmax per query column
        bf16x8_t fp[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int key = k0 + kt * 16 + g * 4 + r;
                    float s = (key < S) ? st[qt][kt][r] : -INFINITY;
                    st[qt][kt][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = xor32_max(xor16_max(mx));
            // deferred rescale: the running max (and with it the O / l accumulators) is only moved when some row
            // of the wave grew by more than RESCALE_THR; otherwise P = exp(s - m_old) <= e^THR, still exact in the
            // final O / l ratio (f32 accumulators).  The decision is taken before this tile's P exists.
            float mnew = mrow[qt];
            if (__any(mx > mrow[qt] + RESCALE_THR)) {
                mnew = fmaxf(mrow[qt], mx);
                const float alpha = __expf(mrow[qt] - mnew);   // exp(-inf) = 0 on the first tile
                mrow[qt] = mnew;
                lrow[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
            }
            float psum = 0.f;
            bf16_t pb[16];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __expf(st[qt][kt][r] - mnew);
                    psum += pv;
                    pb[kt * 4 + r] = f32_to_bf16(pv);
                }
            lrow[qt] += psum;
            // B operand of O^T = V^T P^T for k-step kp: contraction index j<4 -> key (2kp)*16 + g*4 + j,
            // j>=4 -> key (2kp+1)*16 + g*4 + (j-4); the A operand below uses the same mapping.
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                bf16x8_t f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] = (short)pb[(2 * kp) * 4 + j];
                    f[4 + j] = (short)pb[(2 * kp + 1) * 4 + j];
                }
                fp[qt][kp] = f;
            }
        }
        // O^T[d][q] += sum_key V^T[d][key] P^T[key][q]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const bf16_t* vr = sVt + (dt * 16 + l15) * VS_STRIDE;
                bf16x8_t fv;
                const uint32_t* a0 = (const uint32_t*)(vr + (2 * kp) * 16 + g * 4);
                const uint32_t* a1 = (const uint32_t*)(vr + (2 * kp + 1) * 16 + g * 4);
                uint32_t w0 = a0[0], w1 = a0[1], w2 = a1[0], w3 = a1[1];
                fv[0] = (short)(w0 & 0xffff); fv[1] = (short)(w0 >> 16);
                fv[2] = (short)(w1 & 0xffff); fv[3] = (short)(w1 >> 16);
                fv[4] = (short)(w2 & 0xffff); fv[5] = (short)(w2 >> 16);
                fv[6] = (short)(w3 & 0xffff); fv[7] = (short)(w3 >> 16);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o[qt][dt] = mfma16a(fv, fp[qt][kp], o[qt][dt]);
            }
        }
    }

    // finalise: l is a per-lane partial over this lane's keys -> sum the 4 lane groups of each column
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = lrow[qt];
        l = xor32_sum(xor16_sum(l));
        const float inv = 1.0f / l;
        const int q = qbase + qt * 16 + l15;
        if (q < S) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                ushort4 pk;
                pk.x = f32_to_bf16(o[qt][dt][0] * inv); pk.y = f32_to_bf16(o[qt][dt][1] * inv);
                pk.z = f32_to_bf16(o[qt][dt][2] * inv); pk.w = f32_to_bf16(o[qt][dt][3] * inv);
                // out[b][q][h*64 + dt*16 + g*4 .. +3]
                *(ushort4*)(out + ((size_t)b * S + q) * (H * 64) + h * 64 + dt * 16 + g * 4) = pk;
            }
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------------
// Second-generation encoder attention (the default): same S^T = K Q^T / O^T = V^T P^T register scheme as the kernel above, but
//   * 64 queries per wave (4 q-tiles), 256 per block: every K / V fragment read from LDS feeds 4 MFMAs instead of 2
//     (the 32-query kernel needs 128 B/clk of LDS reads per CU at full MFMA rate -- the LDS limit);
//   * V is staged row-major like K into four [64 keys][16 d] sub-tiles and read with ds_read_b64_tr_b16, the hardware
//     4x4 transpose read: lane (d = l15, g) of a 16-lane group receives V[key0 .. key0+3][d] from the 8-byte pieces its
//     group neighbours address -- no transposing 2-byte stores (16 per thread and tile in the first kernel);
//   * K, V and Q reach LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass; two K/V stages,
//     the DMA of tile t+1 issued before the MFMAs of tile t: ONE barrier per 64 keys, L2 latency under the tile's compute;
//   * Q lives in LDS (8 KB per wave), not in 32 registers: with O (64) and the scores (64) resident the kernel is
//     register-bound, and two waves per SIMD are needed to overlap one wave's barrier / LDS waits with the other's work;
//   * the tile body is software-pipelined over the q-tiles in scheduling regions (MFMAs of q-tile i+1 beside the
//     max / exp2 VALU work of q-tile i), with ONE wave-uniform rescale branch per tile instead of one per q-tile;
//   * exp2 with the log2(e) factor folded into one fma; v_max3 by inline asm (fmaxf on MFMA results makes the compiler
//     canonicalise every operand first); the key-validity mask only in the peeled last tile;
//   * O leaves through a wave-private LDS transpose as whole 128-byte rows.
// Measured at the bench shape (B=8, 20 heads, S=1500): 206 -> 133 us per layer (447 -> 690 TFLOP/s).  What is left
// (profiles/r02_attn_*): with head_dim 64 the softmax costs ~300 VALU instructions per 64 MFMAs and tile, and on this chip
// the two do not overlap inside a SIMD to any useful degree -- removing the exponentials alone gives 98 us, removing half the
// MFMAs 118 us: time ~ MFMA cycles + VALU cycles.  The next step is fewer VALU instructions per score (running max folded
// into the MFMA accumulator init), not more scheduling.
// LDS: 2 x (K 8 KB swizzled | 4 x 2080 B V sub-tiles) + 32 KB Q = 65 KB, two blocks per CU.
// ---------------------------------------------------------------------------------------------------
#define A2_KSZ 8192
#define A2_VSUB 2080   // 64 keys x 32 B (+ 32 B pad, kept from the register-staged version; harmless with the DMA)
#define A2_STAGE (A2_KSZ + 4 * A2_VSUB)
#define A2_LOG2E 1.44269504088896340736f

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ inline float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ inline void a3_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ inline s16x4_t lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

// One 64-key tile of attn_encoder_q64_kernel.  MASK: the tile holds keys beyond S.
// The body is two software-pipelined phases.  Each pipeline step is one scheduling region (closed by sched_barrier) that
// holds 16 MFMAs of one q-tile and the VALU work of its neighbour, so the compiler interleaves them (MFMA issue every
// ~4 VALU) without stretching live ranges across the whole tile -- unconstrained, the 1200-instruction block is scheduled
// into 70+ spilled registers, and a scratch reload's vmcnt wait also drains the in-flight DMA of the next stage.
template <bool MASK>
__device__ __forceinline__ void a3_qk(const unsigned char* qw, int qt, int k0, int S, int g, int koff, int kc0, int kc1,
                                      const bf16x8_t (&fk)[4][2], f32x4_t (&st)[4]) {
    // Q fragments from the wave-private LDS copy (same row / swizzle geometry as a K tile): registers are the scarce
    // resource of this kernel, LDS bandwidth is not
    const bf16x8_t fq0 = *(const bf16x8_t*)(qw + qt * 2048 + koff + kc0);
    const bf16x8_t fq1 = *(const bf16x8_t*)(qw + qt * 2048 + koff + kc1);
    const int lim = S - k0 - g * 4;   // MASK: this lane's key (kt, r) exists iff kt*16 + r < lim
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (MASK) {   // last tile: keys beyond S do not exist -- start their accumulators at -inf (K rows beyond S are zero-filled
                      // by the engine, so the products are finite and the score stays -inf; cheaper in registers than a select)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = (kt * 16 + r >= lim) ? -INFINITY : 0.f;
        }
        z = mfma16a(fk[kt][0], fq0, z);
        z = mfma16a(fk[kt][1], fq1, z);
        st[kt] = z;
    }
}

__device__ __forceinline__ float a3_rowmax(const f32x4_t (&st)[4]) {
    float ma = vmax3(st[0][0], st[0][1], st[0][2]);
    float mb = vmax3(st[0][3], st[1][0], st[1][1]);
    ma = vmax3(ma, st[1][2], st[1][3]);
    mb = vmax3(mb, st[2][0], st[2][1]);
    ma = vmax3(ma, st[2][2], st[2][3]);
    mb = vmax3(mb, st[3][0], st[3][1]);
    ma = vmax3(ma, st[3][2], st[3][3]);
    return xor32_max(xor16_max(vmax3(ma, mb, mb)));
}

// P = 2^(s*log2e + m2) for one q-tile, as the two B-operand fragments of O^T = V^T P^T: contraction index j<4 of k-step kp
// -> key (2kp)*16 + g*4 + j, j>=4 -> key (2kp+1)*16 + g*4 + (j-4); the V fragments use the same mapping.
__device__ __forceinline__ float a3_probs(const f32x4_t (&st)[4], float m2, bf16x8_t (&fp)[2]) {
    float psum = 0.f;
    bf16_t pb[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(st[kt][r], A2_LOG2E, m2));
            psum += pv;
            pb[kt * 4 + r] = f32_to_bf16(pv);
        }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
        bf16x8_t f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = (short)pb[(2 * kp) * 4 + j];
            f[4 + j] = (short)pb[(2 * kp + 1) * 4 + j];
        }
        fp[kp] = f;
    }
    return psum;
}

template <bool MASK, int QT>
__device__ __forceinline__ void a3_tile(const unsigned char* base, const unsigned char* qw, int k0, int S, int g, int koff,
                                        int kc0, int kc1, int voff, f32x4_t (&o)[QT][4], float (&mrow)[QT], float (&lrow)[QT]) {
    // ---- phase 1: S^T = K Q^T; st[qt][kt][r] = score(key = k0 + kt*16 + g*4 + r, query = qbase + qt*16 + l15) ----
    bf16x8_t fk[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        fk[kt][0] = *(const bf16x8_t*)(base + kt * 2048 + koff + kc0);
        fk[kt][1] = *(const bf16x8_t*)(base + kt * 2048 + koff + kc1);
    }
    f32x4_t st[QT][4];
    float mnew[QT];
    bool moved = false;
    a3_qk<MASK>(qw, 0, k0, S, g, koff, kc0, kc1, fk, st[0]);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        __builtin_amdgcn_sched_barrier(0);
        if (qt < QT - 1) a3_qk<MASK>(qw, qt + 1, k0, S, g, koff, kc0, kc1, fk, st[qt + 1]);   // MFMAs of q-tile qt+1 ...
        const float mx = a3_rowmax(st[qt]);                                               // ... over the max of q-tile qt
        // deferred rescale: the running max moves only when the row grew by more than RESCALE_THR (P <= e^THR otherwise)
        const bool mv = mx > mrow[qt] + RESCALE_THR;
        mnew[qt] = mv ? mx : mrow[qt];
        moved |= mv;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (__any(moved)) {   // rare after the first tiles: one wave-uniform branch for the four q-tiles
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mnew[qt]) * A2_LOG2E);   // 2^-inf = 0 on the first tile
            mrow[qt] = mnew[qt];
            lrow[qt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
        }
    }
    // ---- phase 2: O^T[d][q] += sum_key V^T[d][key] P^T[key][q]; V^T fragments by transpose read of the row-major sub-tiles ----
    bf16x8_t fv[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const s16x4_t lo = lds_tr16(base + voff + dt * A2_VSUB + (2 * kp) * 512);
            const s16x4_t hi = lds_tr16(base + voff + dt * A2_VSUB + (2 * kp + 1) * 512);
            bf16x8_t f;
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            fv[dt][kp] = f;
        }
    bf16x8_t fp[2][2];
    lrow[0] += a3_probs(st[0], -mrow[0] * A2_LOG2E, fp[0]);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        __builtin_amdgcn_sched_barrier(0);
        if (qt < QT - 1) lrow[qt + 1] += a3_probs(st[qt + 1], -mrow[qt + 1] * A2_LOG2E, fp[(qt + 1) & 1]);   // exps of q-tile qt+1 ...
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)                                                                   // ... under the MFMAs of qt
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = mfma16a(fv[dt][kp], fp[qt & 1][kp], o[qt][dt]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// grid: ceil(S / (64 QT)) * H * B blocks of 256 threads; wave w owns queries q0 + w*16*QT .. (QT q-tiles of 16).
template <int QT>
__global__ __launch_bounds__(256, 2) void attn_encoder_q64_kernel(const bf16_t* __restrict__ Q,
                                                                 const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V,
                                                                 bf16_t* __restrict__ out, int H, int S, int S_pad) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * A2_STAGE + 4 * QT * 2048];   // K/V stages | Q, 2 KB per q-tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (S + 64 * QT - 1) / (64 * QT);
    int lid;
    {   // XCD-aware logical id: the q-blocks of one (batch, head) share an XCD's L2 for their K/V re-reads
        const int nwg = gridDim.x, nx = 8, xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
        const int q = nwg / nx, r = nwg % nx;
        lid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bh = lid / nqb, qblk = lid - bh * nqb;
    const int b = bh / H, h = bh - b * H;
    const size_t head_off = ((size_t)b * H + h) * S_pad * 64;
    const bf16_t* Qh = Q + head_off;
    const bf16_t* Kh = K + head_off;
    const bf16_t* Vh = V + head_off;
    const int qbase = qblk * 64 * QT + wave * 16 * QT;

    f32x4_t o[QT][4];
    float mrow[QT], lrow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrow[qt] = -INFINITY; lrow[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    // staging by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, destination = wave base + lane*16, per-lane
    // source address -- no staging registers, no ds_write pass): wave w moves K rows w*16 .. w*16+15 (two instructions of
    // 8 rows, source chunk XOR-swizzled) and V sub-tile dt = w (two instructions of 32 keys x 32 B).
    const int krow = wave * 16 + (lane >> 3);                                    // + 8 for the second instruction
    const size_t ksrc = (size_t)krow * 64 + (((lane & 7) ^ (krow & 7)) << 3);   // (krow + 8) & 7 == krow & 7
    const size_t vsrc = (size_t)(lane >> 1) * 64 + wave * 16 + (lane & 1) * 8;  // + 32 keys for the second instruction
#define A3_STAGE_DMA(k0_, buf_)                                                             \
    do {                                                                                    \
        unsigned char* sb_ = sm + (buf_) * A2_STAGE;                                        \
        const bf16_t* kp_ = Kh + (size_t)(k0_) * 64;                                        \
        const bf16_t* vp_ = Vh + (size_t)(k0_) * 64;                                        \
        a3_glds16(kp_ + ksrc, sb_ + wave * 2048);                                           \
        a3_glds16(kp_ + ksrc + 8 * 64, sb_ + wave * 2048 + 1024);                           \
        a3_glds16(vp_ + vsrc, sb_ + A2_KSZ + wave * A2_VSUB);                               \
        a3_glds16(vp_ + vsrc + 32 * 64, sb_ + A2_KSZ + wave * A2_VSUB + 1024);              \
    } while (0)
    // per-lane fragment offsets inside a stage
    const int koff = l15 * 128;                                   // + kt*2048, chunk (kk*4+g) ^ (l15 & 7)   ((kt*16+l15)&7 == l15&7)
    const int kc0 = ((g) ^ (l15 & 7)) << 4, kc1 = ((4 + g) ^ (l15 & 7)) << 4;
    const int voff = A2_KSZ + (g * 4 + (l15 >> 2)) * 32 + (l15 & 3) * 8;   // + dt*A2_VSUB + (16-key block)*512

    const int ntiles = (S + 63) / 64;
    // Q: 64 query rows x 128 B per wave, by the same DMA (8 instructions of 8 rows, source chunk XOR-swizzled); rows
    // beyond S are clamped to the last row (computed, never stored)
    unsigned char* qw = sm + 2 * A2_STAGE + wave * QT * 2048;
#pragma unroll
    for (int i = 0; i < 2 * QT; ++i) {
        const int r = i * 8 + (lane >> 3);
        int q = qbase + r;
        if (q >= S) q = S - 1;
        a3_glds16(Qh + (size_t)q * 64 + (((lane & 7) ^ (r & 7)) << 3), qw + i * 1024);
    }
    A3_STAGE_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // full tiles in the loop (one code path: with the masked variant inside the loop the register allocator spills the
    // main path), the tile holding keys beyond S -- if any -- peeled behind it
    const int nfull = S / 64;
    for (int tile = 0; tile < nfull; ++tile) {
        const int k0 = tile * 64;
        const unsigned char* base = sm + (tile & 1) * A2_STAGE;
        if (tile + 1 < ntiles) A3_STAGE_DMA(k0 + 64, (tile + 1) & 1);   // stage (t+1)&1 was last read in tile t-1
        a3_tile<false, QT>(base, qw, k0, S, g, koff, kc0, kc1, voff, o, mrow, lrow);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next stage have landed ...
        __syncthreads();                                   // ... and everyone's; everyone done reading this stage
    }
    if (nfull < ntiles) {
        a3_tile<true, QT>(sm + (nfull & 1) * A2_STAGE, qw, nfull * 64, S, g, koff, kc0, kc1, voff, o, mrow, lrow);
        __syncthreads();   // the stages are reused for the output transpose below
    }

    // finalise: 1/l per query column, O^T -> wave-private LDS rows [64 q][64 d] (16-byte chunks XOR-swizzled by q & 7),
    // then whole 128-byte rows to out[b][q][h*64 ..]
    unsigned char* ow = sm + wave * QT * 2048;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrow[qt];
        l = xor32_sum(xor16_sum(l));
        const float inv = 1.0f / l;
        const int row = qt * 16 + l15;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ushort4 pk;
            pk.x = f32_to_bf16(o[qt][dt][0] * inv); pk.y = f32_to_bf16(o[qt][dt][1] * inv);
            pk.z = f32_to_bf16(o[qt][dt][2] * inv); pk.w = f32_to_bf16(o[qt][dt][3] * inv);
            const int chunk = (dt * 2 + (g >> 1)) ^ (row & 7);
            *(ushort4*)(ow + row * 128 + chunk * 16 + (g & 1) * 8) = pk;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (wave-private region)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 2 * QT; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int q = qbase + row;
        const uint4 v = *(const uint4*)(ow + row * 128 + ((ch ^ (row & 7)) << 4));
        if (q < S) *(uint4*)(out + ((size_t)b * S + q) * (H * 64) + h * 64 + ch * 8) = v;
    }
}

// f32 flavour: one wave per query, scores in LDS.  grid (ceil(S/4), H, B), 256 threads.
__global__ __launch_bounds__(256) void attn_encoder_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                               const float* __restrict__ V, float* __restrict__ out,
                                                               int H, int S, int S_pad) {
    extern __shared__ float sc[];  // [4][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 4 + wave;
    if (q >= S) return;
    const size_t head_off = ((size_t)b * H + h) * S_pad * 64;
    const float* qr = Q + head_off + (size_t)q * 64;
    float* s = sc + (size_t)wave * S;
    float mx = -INFINITY;
    for (int k = lane; k < S; k += 64) {
        const float* kr = K + head_off + (size_t)k * 64;
        float d = 0.f;
#pragma unroll 16
        for (int e = 0; e < 64; ++e) d = fmaf(qr[e], kr[e], d);
        s[k] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < S; k += 64) { float p = expf(s[k] - mx); s[k] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    // lane = output feature; keys sequentially (p read from LDS is a broadcast)
    float acc = 0.f;
    for (int k = 0; k < S; ++k) acc = fmaf(s[k] * inv, V[head_off + (size_t)k * 64 + lane], acc);
    out[((size_t)b * S + q) * (H * 64) + h * 64 + lane] = acc;
}

int cw_launch_attn_encoder(bool bf16, const void* Q, const void* K, const void* V, void* out, int B, int H, int S,
                           int S_pad, hipStream_t st) {
    if (bf16) {
        if (S_pad % KT != 0 || S_pad < S) return CW_ERR_INVALID;
#ifdef CW_EXPERIMENTS
        const bool v1 = cw_sw::cw_switches().attn_v1;   // the round-1 32-queries-per-wave kernel (A/B comparisons)
        if (v1)
            hipLaunchKernelGGL(attn_encoder_bf16_kernel, dim3(((S + 127) / 128) * H * B), dim3(256), 0, st, (const bf16_t*)Q,
                               (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)out, H, S, S_pad);
        else
#endif
            hipLaunchKernelGGL(attn_encoder_q64_kernel<4>, dim3(((S + 255) / 256) * H * B), dim3(256), 0, st, (const bf16_t*)Q,
                               (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)out, H, S, S_pad);
    } else {
        hipLaunchKernelGGL(attn_encoder_f32_kernel, dim3((S + 3) / 4, H, B), dim3(256), (size_t)4 * S * sizeof(float),
                           st, (const float*)Q, (const float*)K, (const float*)V, (float*)out, H, S, S_pad);
    }
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// Decode attention: grid (H, B), 512 threads.  8 lanes cooperate on one 64-wide key row.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct Row8;  // loads 8 consecutive elements as f32
template <> struct Row8<float> {
    __device__ static inline void ld(const float* p, float* o) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};
template <> struct Row8<bf16_t> {
    __device__ static inline void ld(const bf16_t* p, float* o) {
        uint4 a = *(const uint4*)p;
        h16_unpack8(a, o);
    }
};

// 8 consecutive elements held raw (no conversion) so that loads can be issued long before their use
template <typename T> struct Raw8;
template <> struct Raw8<float> {
    float4 a, b;
    __device__ inline void ld(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ inline void cvt(float* o) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};
template <> struct Raw8<bf16_t> {
    uint4 a;
    __device__ inline void ld(const bf16_t* p) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        const u4 t = CW_STREAM_LD((const u4*)p);
        a = make_uint4(t[0], t[1], t[2], t[3]);
    }
    __device__ inline void cvt(float* o) const { h16_unpack8(a, o); }
};

#define DEC_THREADS 512
#define DEC_GROUPS (DEC_THREADS / 8)
#define DEC_PRE 2          // keys per 8-lane group fetched up front (2 x 64 = 128 keys)

// Beam-search self-attention: key k of row b lives in the cache row anc[b][k] (the beam that was in that slot when
// position k was decoded), so hypotheses are re-ordered by rewriting a small index table instead of copying the caches
// of 32 layers every step (HF's cache.reorder_cache).  Same arithmetic as the plain kernel; loads are one table lookup
// deeper, which this (non-headline) path can afford.
template <typename T>
__device__ inline void attn_decode_anc(const DecAttnParams& p, int h, int b) {
    extern __shared__ float dsm[];
    float* sc = dsm;
    float* red = dsm + ((p.cap + 63) & ~63);
    float* scratch = red + DEC_GROUPS * 64;
    const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;
    const int* anc = p.anc + (size_t)b * p.cap;
    const int n_keys = p.pos[b] + 1;
    float qv[8];
    Row8<float>::ld(p.q + (size_t)b * p.H * 64 + h * 64 + sub * 8, qv);
    float mx = -INFINITY;
    for (int k = grp; k < n_keys; k += DEC_GROUPS) {
        const T* kr = (const T*)p.K + (((size_t)anc[k] * p.H + h) * p.cap + k) * 64 + sub * 8;
        float kv[8];
        Row8<T>::ld(kr, kv);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf(qv[e], kv[e], d);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (sub == 0) sc[k] = d;
        mx = fmaxf(mx, d);
    }
    mx = block_max(mx, scratch);
    float sum = 0.f;
    for (int k = tid; k < n_keys; k += DEC_THREADS) { float e = expf(sc[k] - mx); sc[k] = e; sum += e; }
    sum = block_sum(sum, scratch);
    const float inv = 1.0f / sum;
    float acc[8] = {};
    for (int k = grp; k < n_keys; k += DEC_GROUPS) {
        const T* vr = (const T*)p.V + (((size_t)anc[k] * p.H + h) * p.cap + k) * 64 + sub * 8;
        float vv[8];
        Row8<T>::ld(vr, vv);
        const float pk = sc[k] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp * 64 + sub * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        float r = 0.f;
        for (int gI = 0; gI < DEC_GROUPS; ++gI) r += red[gI * 64 + tid];
        if (p.out_frag) p.out_frag[frag_index(b, h * 64 + tid, p.H * 64)] = f32_to_bf16(r);
        else p.out[(size_t)b * p.H * 64 + h * 64 + tid] = r;
    }
}

template <typename T>
__global__ __launch_bounds__(DEC_THREADS) void attn_decode_anc_kernel(DecAttnParams p) { attn_decode_anc<T>(p, blockIdx.x, blockIdx.y); }

// ANC (beam search): the first 128 keys go through the same register-resident path as the plain kernel -- the ancestor table
// entries of a lane's two keys are fetched first (clamped to valid cache rows: entries beyond the current position are
// stale), then K AND V rows, nothing depends on the device-side position -- and only longer histories or alignment capture
// fall back to attn_decode_anc above (serial table lookups, scores through LDS: 9.6 us per layer at 8 items x 5 hypotheses).
// PRE (round 6): 8-lane-group keys requested up front.  2 = 128 keys, the whole history of a typical decode; 1 = 64 keys, chosen by the
// host while every row's history is at most 64 keys long (it knows the step index): at 17..64 rows the launch is bound by the bytes of
// the 128 unconditionally requested rows per (row, head), half of them stale in the first 64 steps of a generate call.  Keys beyond
// PRE x 64 take the on-demand path below either way, so a wrong hint costs time, not correctness; for <= 64 keys the PRE = 2 form
// masks its second key per group out, i.e. both forms do the same arithmetic in the same order (bit-identical).
template <typename T, bool ANC, int PRE = DEC_PRE>
__global__ __launch_bounds__(DEC_THREADS) void attn_decode_kernel(DecAttnParams p) {
    extern __shared__ float dsm[];          // scores [cap rounded] | red [DEC_GROUPS][64] | scratch [64]
    const int h = blockIdx.x, b = blockIdx.y;
    float* sc = dsm;
    float* red = dsm + ((p.cap + 63) & ~63);
    float* scratch = red + DEC_GROUPS * 64;
    const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;
    // cache row of this query row: itself; the audio item it belongs to when several rows (beams) share one encoder
    // window (kv_div = beams per item); or, per key, the ancestor row that wrote that position (beam search: `anc`)
    const int bk = p.kv_div > 1 ? b / p.kv_div : b;
    const T* Kh = (const T*)p.K + ((size_t)bk * p.H + h) * p.cap * 64;
    const T* Vh = (const T*)p.V + ((size_t)bk * p.H + h) * p.cap * 64;
    float qv[8];
    Row8<float>::ld(p.q + (size_t)b * p.H * 64 + h * 64 + sub * 8, qv);
    int arow[PRE];
    if (ANC) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) arow[u] = p.anc[(size_t)b * p.cap + min(grp + u * DEC_GROUPS, p.cap - 1)];
    }
    // The first PRE keys of every 8-lane group (128 keys in all: the whole self-attention history of a typical
    // decode) are fetched before anything else, K AND V, with the row clamped to the cache capacity instead of to
    // n_keys: the loads then depend neither on the device-side position nor on the softmax, which takes two memory
    // round trips (pos -> K rows, softmax -> V rows) out of this latency-bound kernel.  Rows >= n_keys hold stale but
    // addressable data and are masked out below.
    Raw8<T> kpre[PRE], vpre[PRE];
    if (ANC) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const size_t ro = (((size_t)min(max(arow[u], 0), p.B - 1) * p.H + h) * p.cap + min(grp + u * DEC_GROUPS, p.cap - 1)) * 64 + sub * 8;
            kpre[u].ld((const T*)p.K + ro);
            vpre[u].ld((const T*)p.V + ro);
        }
    } else {
#pragma unroll
        for (int u = 0; u < PRE; ++u) kpre[u].ld(Kh + (size_t)min(grp + u * DEC_GROUPS, p.cap - 1) * 64 + sub * 8);
#pragma unroll
        for (int u = 0; u < PRE; ++u) vpre[u].ld(Vh + (size_t)min(grp + u * DEC_GROUPS, p.cap - 1) * 64 + sub * 8);
    }
    const int n_keys = p.pos ? (p.n_keys > 0 ? p.n_keys : p.pos[b] + 1) : p.n_keys;
    if (ANC && (n_keys > PRE * DEC_GROUPS || p.align_out)) { attn_decode_anc<T>(p, h, b); return; }

    if (n_keys <= PRE * DEC_GROUPS && !p.align_out) {
        // Short history (<= 128 keys: every key row is already in registers): scores stay in registers, the only block-wide
        // exchanges are the 8 wave maxima and the final 8 x 64 partial outputs -- 2 barriers instead of 7 and no 64-deep
        // serial LDS reduction (same structure as attn_cross_split_kernel).
        float* s_max = scratch;               // [8]
        float* red8 = red;                    // [8][64] + [8] sums behind it
        const int lane = tid & 63, wave = tid >> 6;
        float d[PRE], mxl = -INFINITY;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            float kv[8];
            kpre[u].cvt(kv);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(qv[e], kv[e], t);
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
            d[u] = (grp + u * DEC_GROUPS < n_keys) ? t : -INFINITY;
            mxl = fmaxf(mxl, d[u]);
        }
        mxl = wave_max(mxl);
        if (lane == 0) s_max[wave] = mxl;
        __syncthreads();
        mxl = s_max[0];
#pragma unroll
        for (int w = 1; w < DEC_THREADS / 64; ++w) mxl = fmaxf(mxl, s_max[w]);
        float acc[8] = {};
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            if (grp + u * DEC_GROUPS < n_keys) {          // stale rows may hold non-finite bit patterns: skip, not scale
                const float pk = expf(d[u] - mxl);
                if (sub == 0) lsum += pk;
                float vv[8];
                vpre[u].cvt(vv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {                       // sum over the wave's 8 key groups (lane bits 3, 4, 5)
            float v = acc[e] + dpp_mov<0x128, 0xf>(0.f, acc[e]);
            acc[e] = xor32_sum(xor16_sum(v));
        }
        lsum = wave_sum(lsum);
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red8[wave * 64 + sub * 8 + e] = acc[e];
        }
        if (lane == 0) red8[8 * 64 + wave] = lsum;
        __syncthreads();
        if (tid < 64) {
            float r = 0.f, l = 0.f;
#pragma unroll
            for (int w = 0; w < DEC_THREADS / 64; ++w) { r += red8[w * 64 + tid]; l += red8[8 * 64 + w]; }
            r *= 1.0f / l;
            if (p.out_frag) p.out_frag[frag_index(b, h * 64 + tid, p.H * 64)] = f32_to_bf16(r);
            else p.out[(size_t)b * p.H * 64 + h * 64 + tid] = r;
        }
        return;
    }

    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int k = grp + u * DEC_GROUPS;
        float kv[8];
        kpre[u].cvt(kv);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf(qv[e], kv[e], d);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (k < n_keys) {
            if (sub == 0) sc[k] = d;
            mx = fmaxf(mx, d);
        }
    }
    // remaining keys: 4 key rows per thread in flight (unrolled by 4 x DEC_GROUPS keys)
    for (int k0 = grp + PRE * DEC_GROUPS; k0 < n_keys; k0 += 4 * DEC_GROUPS) {
        float kv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // unconditional loads (clamped row): no exec-masked blocks, all in flight
            const int k = min(k0 + u * DEC_GROUPS, n_keys - 1);
            Row8<T>::ld(Kh + (size_t)k * 64 + sub * 8, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * DEC_GROUPS;
            if (k < n_keys) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(qv[e], kv[u][e], d);
                d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);   // (a DPP quad_perm/half-mirror version measured slower here)
                if (sub == 0) sc[k] = d;
                mx = fmaxf(mx, d);
            }
        }
    }
    mx = block_max(mx, scratch);
    float sum = 0.f;
    for (int k = tid; k < n_keys; k += DEC_THREADS) { float e = expf(sc[k] - mx); sc[k] = e; sum += e; }
    sum = block_sum(sum, scratch);      // includes the barriers that publish sc[]
    const float inv = 1.0f / sum;

    const int slot = p.align_out ? p.align_slot[h] : -1;
    if (slot >= 0) {
        const int arow = p.pos[b];
        float* dst = p.align_out + (((size_t)b * p.n_align + slot) * p.align_rows + arow) * n_keys;
        for (int k = tid; k < n_keys; k += DEC_THREADS) dst[k] = sc[k] * inv;
    }

    float acc[8] = {};
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int k = grp + u * DEC_GROUPS;
        float vv[8];
        vpre[u].cvt(vv);
        const float pk = (k < n_keys) ? sc[k] * inv : 0.f;
        if (k < n_keys) {                                   // stale rows may hold non-finite bit patterns: skip, not scale
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
        }
    }
    for (int k0 = grp + PRE * DEC_GROUPS; k0 < n_keys; k0 += 4 * DEC_GROUPS) {
        float vv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + u * DEC_GROUPS, n_keys - 1);
            Row8<T>::ld(Vh + (size_t)k * 64 + sub * 8, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * DEC_GROUPS;
            if (k < n_keys) {
                const float pk = sc[k] * inv;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(pk, vv[u][e], acc[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp * 64 + sub * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        float r = 0.f;
        for (int gI = 0; gI < DEC_GROUPS; ++gI) r += red[gI * 64 + tid];
        if (p.out_frag) p.out_frag[frag_index(b, h * 64 + tid, p.H * 64)] = f32_to_bf16(r);
        else p.out[(size_t)b * p.H * 64 + h * 64 + tid] = r;
    }
}

int cw_launch_attn_decode(bool bf16, const DecAttnParams& p, hipStream_t st) {
    size_t lds = ((size_t)((p.cap + 63) & ~63) + DEC_GROUPS * 64 + 64) * sizeof(float);
    const bool anc_v1 = cw_sw::cw_switches().anc_attn_v1;   // A/B: beam self-attention through the serial kernel only
    if (p.anc && (anc_v1 || p.n_keys > 0 || !p.pos)) {
        if (bf16) hipLaunchKernelGGL((attn_decode_anc_kernel<bf16_t>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
        else hipLaunchKernelGGL((attn_decode_anc_kernel<float>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
    } else if (p.anc) {
        if (bf16 && p.short_hist && !p.align_out) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, true, 1>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
        else if (bf16) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, true>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
        else hipLaunchKernelGGL((attn_decode_kernel<float, true>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
    } else if (bf16 && p.short_hist && !p.align_out && p.pos && p.n_keys <= 0)
        hipLaunchKernelGGL((attn_decode_kernel<bf16_t, false, 1>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
    else if (bf16)
        hipLaunchKernelGGL((attn_decode_kernel<bf16_t, false>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
    else
        hipLaunchKernelGGL((attn_decode_kernel<float, false>), dim3(p.H, p.B), dim3(DEC_THREADS), lds, st, p);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// Cross-attention decode, split over keys.  grid (H, B, ATT_NS), 512 threads.
// ---------------------------------------------------------------------------------------------------
// Round-2 structure: every lane keeps the scores of its (up to 4) keys in registers -- the K pass, the exponentials and
// the V pass use the same key ownership, so no score ever goes through LDS (the alignment heads write theirs straight
// to the alignment buffer) -- and the only block-wide exchanges are the running maximum (8 floats) and the final
// 8 x 64 partial outputs: 2 barriers per block instead of 7, the 64-group serial LDS reduction became three cross-lane
// steps + 8 adds.  Needs nk <= 4 * 64 keys per block (ATT_NS >= 6 for 1500 frames).
__device__ inline float row_ror8_add(float v) { return v + dpp_mov<0x128, 0xf>(0.f, v); }   // + lane ^ 8 (row_ror:8)
// FUSED (NQ = 1, fused out-projection / query stage, decfuse.hip): the query is finished here,
//     q = rstd(x1) (qa + qb - mean(x1) qw) + qbias,
// see the FUSED block below (every wave-level 16 B-per-lane load costs 16 clocks of the CU's address unit whatever it fetches:
// all eight waves fetching the residual row and 8 columns of each constant measured +3 us per launch; wave 0 alone + a block
// barrier +1.3 us).
template <typename T, int NQ, bool FUSED>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_split_kernel(CrossSplitParams p) {
    __shared__ float s_max[8 * NQ];
    __shared__ float red[8 * NQ * 64];
    __shared__ float red_l[8 * NQ];
    __shared__ float s_q[FUSED ? 8 * 64 : 1];
    // blockIdx.y = group of NQ consecutive query rows that share one K/V (NQ = kv_div: the hypotheses of one audio item under
    // beam search; NQ = 1: one row per block, K/V of item row / kv_div).  The K/V slice is read ONCE into registers and all
    // NQ queries go through the same three block barriers together: one block per row re-streamed the 64 KB slice from L2
    // kv_div times (26 us per layer at 8 items x 5 beams against 12.5 us at 8 rows), and a serial loop over the queries
    // inside one block was slower still (32 us: 3 barriers + reductions per query).
    const int h = blockIdx.x, b0 = blockIdx.y * NQ, sp = blockIdx.z;
    const int bk = p.kv_div > 1 ? b0 / p.kv_div : b0;
    const int per = (p.n_keys + ATT_NS - 1) / ATT_NS;
    const int k_lo = sp * per, k_hi = min(p.n_keys, k_lo + per), nk = k_hi - k_lo;
    const int tid = threadIdx.x, lane = tid & 63, sub = tid & 7, grp = tid >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T* Kh = (const T*)p.K + (((size_t)bk * p.H + h) * p.n_keys + k_lo) * 64 + sub * 8;
    const T* Vh = (const T*)p.V + (((size_t)bk * p.H + h) * p.n_keys + k_lo) * 64 + sub * 8;
    constexpr int G = CROSS_THREADS / 8;                       // 64 key groups

    // FUSED: every wave finishes the query itself -- lane c takes column c of the head: two 8-byte loads of the per-block
    // LayerNorm partial sums the producing GEMV left behind (decfuse.hip, StackSeg::pstats) and four 4-byte loads (256 B per
    // wave instruction: a quarter of the address-unit time of a 16 B-per-lane load), a wave-local reduction, and a trip
    // through a wave-private LDS row to hand each lane its 8 columns.  No block barrier, no wave waits for another one;
    // the kernel stays at 64 VGPRs (one more costs a resident block per CU).
    float2 pt0 = make_float2(0.f, 0.f), pt1 = pt0;
    float qa1 = 0.f, qb1 = 0.f, qw1 = 0.f, qc1 = 0.f;
    if (FUSED) {                                                // loads only: using a value in here makes hipcc wait before the K/V loads go out
        const int Dm = p.H * 64;
        const float* ps = p.pstats + (size_t)(b0 >> 4) * p.n_pstats * 32;   // one [tiles][16][2] plane per group of 16 rows (gemv_stack_kernel)
        pt0 = *(const float2*)(ps + ((size_t)min(lane, p.n_pstats - 1) * 16 + (b0 & 15)) * 2);
        pt1 = *(const float2*)(ps + ((size_t)min(lane + 64, p.n_pstats - 1) * 16 + (b0 & 15)) * 2);
        const size_t col = (size_t)h * 64 + lane;
        qa1 = p.qa[(size_t)b0 * Dm + col]; qb1 = p.qb[(size_t)b0 * Dm + col];
        qw1 = p.qw[col]; qc1 = p.qbias[col];
    }
    Raw8<T> kr[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kr[u].ld(Kh + (size_t)min(grp + u * G, nk - 1) * 64);   // unconditional, clamped
#pragma unroll
    for (int u = 0; u < 4; ++u) vr[u].ld(Vh + (size_t)min(grp + u * G, nk - 1) * 64);
    if (FUSED) __builtin_amdgcn_sched_barrier(0);              // every load is out before the first wait (hipcc otherwise holds two V loads back)
    // block-uniform scalars of the alignment capture, requested HERE -- behind the K/V requests, under their flight: left at their uses
    // hipcc emits `s_load_dword` + `s_waitcnt lgkmcnt(0)` behind the first barrier and inside the V pass (scalar-cache round trips on
    // the block's critical path); in front of the K/V requests they would delay the stream by two dependent round trips
    const int slot = p.align_out ? p.align_slot[h] : -1;
    int apos[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) apos[q] = (slot >= 0) ? p.pos[b0 + q] : 0;
    float qv[NQ][8];
    if (FUSED) {
        const float inv_d = 1.0f / (float)(p.H * 64);
        const float ps1 = (lane < p.n_pstats ? pt0.x : 0.f) + (lane + 64 < p.n_pstats ? pt1.x : 0.f);
        const float ps2 = (lane < p.n_pstats ? pt0.y : 0.f) + (lane + 64 < p.n_pstats ? pt1.y : 0.f);
        const float mean = wave_sum(ps1) * inv_d;
        const float var = fmaxf(wave_sum(ps2) * inv_d - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float* sq = s_q + wave * 64;                             // wave-private: ordered by the wave's own lgkmcnt
        sq[lane] = ((qa1 + qb1) - mean * qw1) * rstd + qc1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[0][e] = sq[sub * 8 + e];
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) Row8<float>::ld(p.q + (size_t)(b0 + q) * p.H * 64 + h * 64 + sub * 8, qv[q]);
    }
    float d[NQ][4], mx[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) mx[q] = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float kv[8];
        kr[u].cvt(kv);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(qv[q][e], kv[e], t);
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
            d[q][u] = (grp + u * G < nk) ? t : -INFINITY;
            mx[q] = fmaxf(mx[q], d[q][u]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        mx[q] = wave_max(mx[q]);
        if (lane == 0) s_max[wave * NQ + q] = mx[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float m = s_max[q];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, s_max[w * NQ + q]);
        mx[q] = m;
    }

    float acc[NQ][8];
    float lsum[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        lsum[q] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = grp + u * G;
        float vv[8];
        vr[u].cvt(vv);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float pk = (k < nk) ? expf(d[q][u] - mx[q]) : 0.f;
            if (sub == 0 && k < nk) {
                lsum[q] += pk;
                if (slot >= 0) {                               // un-normalised; align_normalize_kernel finishes the row
                    const size_t rowi = ((size_t)(b0 + q) * p.n_align + slot) * p.align_rows + apos[q];
                    p.align_out[rowi * p.n_keys + k_lo + k] = pk;
                }
            }
            if (k < nk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[q][e] = fmaf(pk, vv[e], acc[q][e]);
            }
        }
    }
    // sum over the wave's 8 key groups (lane bits 3, 4, 5)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = xor32_sum(xor16_sum(row_ror8_add(acc[q][e])));
        lsum[q] = wave_sum(lsum[q]);
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(wave * NQ + q) * 64 + sub * 8 + e] = acc[q][e];
        }
        if (lane == 0) red_l[wave * NQ + q] = lsum[q];
    }
    __syncthreads();
    for (int i = tid; i < NQ * 64; i += CROSS_THREADS) {
        const int q = i >> 6, c = i & 63;
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) r += red[(w * NQ + q) * 64 + c];
        p.part_o[((size_t)sp * p.B + b0 + q) * p.H * 64 + h * 64 + c] = r;
    }
    if (tid >= 64 * NQ && tid < 64 * NQ + NQ) {
        const int q = tid - 64 * NQ;
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) l += red_l[w * NQ + q];
        float* ml = p.part_ml + (((size_t)(b0 + q) * p.H + h) * ATT_NS + sp) * 2;
        ml[0] = mx[q]; ml[1] = l;
        if (slot >= 0) {
            const size_t rowi = ((size_t)(b0 + q) * p.n_align + slot) * p.align_rows + apos[q];
            p.align_ml[(rowi * ATT_NS + sp) * 2] = mx[q]; p.align_ml[(rowi * ATT_NS + sp) * 2 + 1] = l;
        }
    }
}

#ifdef CW_EXPERIMENTS   // measured slower than the key-split kernel at every batch size (17.1 vs 11.7 us at B = 8, 103 vs 76 us at B = 64)
// ---------------------------------------------------------------------------------------------------
// Cross-attention decode, ONE block per (row, head) over all keys (six-launch layer, decfuse.hip).  grid (H, B), 512 threads:
// an 8-lane group owns keys grp, grp+64, ... (24 rows of 128 B at 1500 frames) and every K and V row of the block is
// requested before anything is waited for (48 x 16 B per lane in flight, 384 KB per block), so there are no key splits, no
// partial planes and nothing to combine: the block writes the finished attention output.  The query is finished here as well,
//     q = rstd(x1) (qa + qb - mean(x1) qw) + qbias
// by wave 0 alone (LayerNorm statistics of the residual row, wave-local) and handed to the other waves through LDS: every
// wave-level load costs 16 clocks of the CU's address unit whatever it fetches, and eight waves fetching the same row cost
// 3 us per launch.  Alignment heads write exp(s - M) with the block maximum M and (M, L) in split slot 0 of align_ml (the
// other slots (M, 0)), which is exactly what align_normalize_kernel expects from a one-split launch.
// ---------------------------------------------------------------------------------------------------
#define CROSSF_U 24
template <typename T>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_full_kernel(CrossSplitParams p) {
    __shared__ float s_q[64];
    __shared__ float s_max[8];
    __shared__ float red[8 * 64];
    __shared__ float red_l[8];
    const int h = blockIdx.x, b = blockIdx.y;
    const int nk = p.n_keys;
    const int tid = threadIdx.x, lane = tid & 63, sub = tid & 7, grp = tid >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dm = p.H * 64, nvec = Dm >> 2;
    const T* Kh = (const T*)p.K + ((size_t)b * p.H + h) * nk * 64 + sub * 8;
    const T* Vh = (const T*)p.V + ((size_t)b * p.H + h) * nk * 64 + sub * 8;
    constexpr int G = CROSS_THREADS / 8;                       // 64 key groups
    float4 xrow[5];
    float qa1 = 0.f, qb1 = 0.f, qw1 = 0.f, qc1 = 0.f;
    if (wave == 0 && p.xstat) {                                 // wave-uniform
        const float* xr = p.xstat + (size_t)b * Dm;
#pragma unroll
        for (int c = 0; c < 5; ++c) xrow[c] = *(const float4*)(xr + (size_t)min(lane + 64 * c, nvec - 1) * 4);
        const size_t col = (size_t)h * 64 + lane;
        qa1 = p.qa[(size_t)b * Dm + col];
        for (int sI = 1; sI < p.q_planes; ++sI) qa1 += p.qa[(size_t)sI * p.q_plane_stride + (size_t)b * Dm + col];   // K-split planes (skinny.hip), slice order
        qb1 = p.qb ? p.qb[(size_t)b * Dm + col] : 0.f;
        qw1 = p.qw[col]; qc1 = p.qbias[col];
    } else if (wave == 0) {
        qa1 = p.q[(size_t)b * Dm + h * 64 + lane];              // finished query
    }
    Raw8<T> kr[CROSSF_U], vr[CROSSF_U];
#pragma unroll
    for (int u = 0; u < CROSSF_U; ++u) kr[u].ld(Kh + (size_t)min(grp + u * G, nk - 1) * 64);   // unconditional, clamped
#pragma unroll
    for (int u = 0; u < CROSSF_U; ++u) vr[u].ld(Vh + (size_t)min(grp + u * G, nk - 1) * 64);
    if (wave == 0 && !p.xstat) s_q[lane] = qa1;
    if (wave == 0 && p.xstat) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            sx += ok * ((xrow[c].x + xrow[c].y) + (xrow[c].z + xrow[c].w));
        }
        const float mean = wave_sum(sx) / (float)Dm;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float ok = (lane + 64 * c < nvec) ? 1.f : 0.f;
            const float a = xrow[c].x - mean, bb = xrow[c].y - mean, cc = xrow[c].z - mean, d2 = xrow[c].w - mean;
            sq += ok * ((a * a + bb * bb) + (cc * cc + d2 * d2));
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)Dm + 1e-5f);
        s_q[lane] = ((qa1 + qb1) - mean * qw1) * rstd + qc1;
    }
    __syncthreads();
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = s_q[sub * 8 + e];
    float d[CROSSF_U], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < CROSSF_U; ++u) {
        float kv[8];
        kr[u].cvt(kv);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(qv[e], kv[e], t);
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        d[u] = (grp + u * G < nk) ? t : -INFINITY;
        mx = fmaxf(mx, d[u]);
    }
    mx = wave_max(mx);
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    mx = s_max[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, s_max[w]);

    const int slot = p.align_out ? p.align_slot[h] : -1;
    const size_t rowi = slot >= 0 ? ((size_t)b * p.n_align + slot) * p.align_rows + p.pos[b] : 0;
    float acc[8] = {};
    float lsum = 0.f;
#pragma unroll
    for (int u = 0; u < CROSSF_U; ++u) {
        const int k = grp + u * G;
        float vv[8];
        vr[u].cvt(vv);
        const float pk = (k < nk) ? expf(d[u] - mx) : 0.f;
        if (sub == 0 && k < nk) {
            lsum += pk;
            if (slot >= 0) p.align_out[rowi * nk + k] = pk;     // un-normalised; align_normalize_kernel finishes the row
        }
        if (k < nk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = xor32_sum(xor16_sum(row_ror8_add(acc[e])));
    lsum = wave_sum(lsum);
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * 64 + sub * 8 + e] = acc[e];
    }
    if (lane == 0) red_l[wave] = lsum;
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) l += red_l[w];
    if (tid < 64) {
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) r += red[w * 64 + tid];
        if (p.a_frag) ((bf16_t*)p.a_frag)[frag_index(b, h * 64 + tid, Dm)] = f32_to_bf16(r * (1.0f / l));   // the out-projection's MFMA rows
        else p.a_out[(size_t)b * Dm + h * 64 + tid] = r * (1.0f / l);
    } else if (slot >= 0 && tid < 64 + ATT_NS) {
        const int sI = tid - 64;
        p.align_ml[(rowi * ATT_NS + sI) * 2] = mx;
        p.align_ml[(rowi * ATT_NS + sI) * 2 + 1] = sI == 0 ? l : 0.f;
    }
}
#endif

// ---------------------------------------------------------------------------------------------------
// Cross-attention decode for nq = 2..16 rows that share one K/V (the hypotheses of one audio item under beam search), on the
// matrix cores.  grid (H, B / nq, ATT_NS), 512 threads, 16-bit caches.  The VALU kernel above does nq x 250 dot products, the
// exponentials and nq x 250 x 64 multiply-adds per block in 8-lane groups: at 5 hypotheses it is instruction-bound (2600
// instructions per wave, 25.8 us per layer against 13 us at one row for the same 61 MB of K/V).  Here a wave owns 32 consecutive
// keys of the block's slice and
//   * S^T = K Q^T is six 16x16x32 MFMAs per 16-key tile (A = the K rows as loaded from HBM, B = the queries; the queries are
//     split q = hi + mid + lo into three 16-bit numbers so the product is that of the f32 query the VALU kernel used;
//     the contraction index is permuted, dim = g*16 + s*8 + j, so that a lane's two K loads are 32 contiguous bytes),
//   * the C fragment of S^T (lane: query l&15, keys g*4 .. g*4+3 of each tile) IS the B fragment of the second product
//     O^T = V^T P^T under the key order (g*4+j of tile 0, then of tile 1): no cross-lane movement for P (again three halves),
//   * V^T comes out of a wave-private LDS image of the wave's 32 V rows (row-major as loaded, 144-byte rows) through
//     `ds_read_b64_tr_b16`, the gfx950 transposing read: each 16-lane group hands the hardware four rows x 16 columns and gets
//     column l&15 of them.
// The running maximum (one exchange through LDS), the partial planes, (m, l) pairs and alignment rows are exactly those of the
// VALU kernel, so the out-projection's combine and align_normalize_kernel do not change.
// ---------------------------------------------------------------------------------------------------
#define XM_VS 72                                                  // V image row stride (elements): 16-byte aligned rows
// x = hi + mid + lo in the 16-bit type: 3 x 8 mantissa bits (bf16) cover the 24 of an f32, so the MFMAs below multiply by the f32
// value the 8-lane-group kernel multiplies by (two halves leave 2^-17: enough for the tolerance, but the synthetic bench model's
// beam search then parts from the other kernel's at near-ties; three halves agree to f32 summation order)
__device__ inline void split3(const float* x, bf16x8_t& hi, bf16x8_t& mid, bf16x8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = f32_to_bf16(x[e]);
        const float r1 = x[e] - bf16_to_f32(h);
        const bf16_t m = f32_to_bf16(r1);
        hi[e] = (short)h;
        mid[e] = (short)m;
        lo[e] = (short)f32_to_bf16(r1 - bf16_to_f32(m));
    }
}
typedef short xm_v4s __attribute__((ext_vector_type(4)));
typedef unsigned int xm_u4 __attribute__((ext_vector_type(4)));
// FUSED (round 6: the fused out-projection / cross-query stage of decfuse.hip under beam search): the rows arrive as the two halves
// qa, qb of the un-normalised query and per-tile LayerNorm partial sums of x1 (gemv_stack_kernel), and the block finishes
//     q = rstd(x1) (qa + qb - mean(x1) qw) + qbias
// for its nq rows itself: every wave two of the up to 16 rows into a 4 KB LDS image, one block barrier, then every lane picks its
// fragment (row r, columns g*16 .. +15) up.  See the load block in the kernel for why the operands are requested first and by all waves.
template <bool TR, bool FUSED = false>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_mfma_kernel(CrossSplitParams p, int nq) {
    __shared__ __attribute__((aligned(16))) bf16_t s_v[8 * 32 * XM_VS];   // per wave: 32 V rows, later its 16 x 64 partial outputs
    __shared__ float s_max[8 * 16];
    __shared__ float red_l[8 * 16];
    __shared__ __attribute__((aligned(16))) float s_qf[FUSED ? 16 * 64 : 4];
    const int h = blockIdx.x, b0 = blockIdx.y * nq, sp = blockIdx.z;
    const int bk = b0 / p.kv_div;
    const int per = (p.n_keys + ATT_NS - 1) / ATT_NS;
    const int k_lo = sp * per, k_hi = min(p.n_keys, k_lo + per), nk = k_hi - k_lo;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave * 32;
    const int D = p.H * 64;
    const bf16_t* Kh = (const bf16_t*)p.K + (((size_t)bk * p.H + h) * p.n_keys + k_lo) * 64;
    const bf16_t* Vh = (const bf16_t*)p.V + (((size_t)bk * p.H + h) * p.n_keys + k_lo) * 64;
    // FUSED: the query operands are requested BEFORE the K/V rows -- issued behind them they would queue behind the CU's own 256 KB
    // of K/V requests and the block would sit at the barrier below until the stream has landed (27.3 us per launch against 15.6
    // for a finished query) -- and by ALL waves, two rows each (wave w: rows 2 w, 2 w + 1; lane: row 2 w + lane / 32, columns
    // 2 (lane % 32), + 1 and statistics tiles lane % 32, + 32, + 64): 7 eight-byte loads and 14 registers per lane.  (Wave 0 alone
    // carrying the 16 x 64 values: 160 VGPRs, one block per CU instead of three.)
    float2 fs[3], fa, fb, fw, fc;
    const int qrow_l = 2 * wave + (lane >> 5), qcol = (lane & 31) * 2;
    if (FUSED) {
        const int row = b0 + min(qrow_l, nq - 1);
        const float* ps = p.pstats + ((size_t)(row >> 4) * p.n_pstats * 16 + (row & 15)) * 2;   // [group of 16 rows][tile][16][2]
#pragma unroll
        for (int i = 0; i < 3; ++i) fs[i] = *(const float2*)(ps + (size_t)min((lane & 31) + 32 * i, p.n_pstats - 1) * 32);
        const size_t col = (size_t)h * 64 + qcol;
        fa = *(const float2*)(p.qa + (size_t)row * D + col); fb = *(const float2*)(p.qb + (size_t)row * D + col);
        fw = *(const float2*)(p.qw + col); fc = *(const float2*)(p.qbias + col);
    }
    // every load of the block first: 4 x 16 B of K (A fragments), 4 x 16 B of V (row-major), the lane's 16 query values
    // (ext-vector registers, not HIP's struct uint4: a struct copy is a memcpy, and across the query barrier of the FUSED form hipcc kept
    // the four V rows in SCRATCH memory -- stored after the load, reloaded for the LDS write: 25.9 us per launch against 15.2)
    xm_u4 kf[2][2], vr[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const bf16_t* kp = Kh + (size_t)min(kb + t * 16 + r, nk - 1) * 64 + g * 16;
        kf[t][0] = *(const xm_u4*)kp; kf[t][1] = *(const xm_u4*)(kp + 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        vr[u] = *(const xm_u4*)(Vh + (size_t)min(kb + (lane >> 3) + 8 * u, nk - 1) * 64 + (lane & 7) * 8);
    float qf[2][8];
    if (FUSED) {
        __builtin_amdgcn_sched_barrier(0);                        // every request is out before the first wait
        // (hipcc otherwise consumes the statistics BETWEEN the K and the V requests -- three waits in the middle of the issue sequence, to
        // reuse their registers for the V rows: 25.9 us per launch against 15.2; the empty asm pins the operands behind the barrier)
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(fs[i].x), "+v"(fs[i].y));
        asm volatile("" : "+v"(fa.x), "+v"(fa.y), "+v"(fb.x), "+v"(fb.y), "+v"(fw.x), "+v"(fw.y), "+v"(fc.x), "+v"(fc.y));
        {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float ok = ((lane & 31) + 32 * i < p.n_pstats) ? 1.f : 0.f;
                s1 += ok * fs[i].x; s2 += ok * fs[i].y;
            }
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }   // over the 32 lanes of the row
            const float inv_d = 1.0f / (float)D;
            const float mean = s1 * inv_d;
            const float rstd = 1.0f / sqrtf(fmaxf(s2 * inv_d - mean * mean, 0.f) + 1e-5f);
            float2 o;
            o.x = ((fa.x + fb.x) - mean * fw.x) * rstd + fc.x;
            o.y = ((fa.y + fb.y) - mean * fw.y) * rstd + fc.y;
            *(float2*)(s_qf + qrow_l * 64 + qcol) = o;
        }
        // barrier WITHOUT the vmcnt drain of __syncthreads(): the K/V rows stay in flight across it -- the query was requested first and
        // is long there; LDS traffic is ordered by lgkmcnt alone
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* qp = s_qf + r * 64 + g * 16;
        const float4 q0 = *(const float4*)qp, q1 = *(const float4*)(qp + 4), q2 = *(const float4*)(qp + 8), q3 = *(const float4*)(qp + 12);
        const float z = r < nq ? 1.f : 0.f;
        qf[0][0] = q0.x * z; qf[0][1] = q0.y * z; qf[0][2] = q0.z * z; qf[0][3] = q0.w * z;
        qf[0][4] = q1.x * z; qf[0][5] = q1.y * z; qf[0][6] = q1.z * z; qf[0][7] = q1.w * z;
        qf[1][0] = q2.x * z; qf[1][1] = q2.y * z; qf[1][2] = q2.z * z; qf[1][3] = q2.w * z;
        qf[1][4] = q3.x * z; qf[1][5] = q3.y * z; qf[1][6] = q3.z * z; qf[1][7] = q3.w * z;
    } else {
        const float* qp = p.q + (size_t)(b0 + min(r, nq - 1)) * D + h * 64 + g * 16;
        const float4 q0 = *(const float4*)qp, q1 = *(const float4*)(qp + 4), q2 = *(const float4*)(qp + 8), q3 = *(const float4*)(qp + 12);
        const float z = r < nq ? 1.f : 0.f;                       // fragment columns past the last row: zero queries
        qf[0][0] = q0.x * z; qf[0][1] = q0.y * z; qf[0][2] = q0.z * z; qf[0][3] = q0.w * z;
        qf[0][4] = q1.x * z; qf[0][5] = q1.y * z; qf[0][6] = q1.z * z; qf[0][7] = q1.w * z;
        qf[1][0] = q2.x * z; qf[1][1] = q2.y * z; qf[1][2] = q2.z * z; qf[1][3] = q2.w * z;
        qf[1][4] = q3.x * z; qf[1][5] = q3.y * z; qf[1][6] = q3.z * z; qf[1][7] = q3.w * z;
    }
    bf16x8_t qh[2], qm[2], ql[2];
    split3(qf[0], qh[0], qm[0], ql[0]);
    split3(qf[1], qh[1], qm[1], ql[1]);
    // S^T: lane holds query r, keys kb + t*16 + g*4 + i
    f32x4_t st[2];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, kf[t][0]), a1 = __builtin_bit_cast(bf16x8_t, kf[t][1]);
        c = cw_mfma_16x16x32(a0, ql[0], c);                     // small terms first
        c = cw_mfma_16x16x32(a1, ql[1], c);
        c = cw_mfma_16x16x32(a0, qm[0], c);
        c = cw_mfma_16x16x32(a1, qm[1], c);
        c = cw_mfma_16x16x32(a0, qh[0], c);
        c = cw_mfma_16x16x32(a1, qh[1], c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = (kb + t * 16 + g * 4 + i < nk) ? c[i] : -INFINITY;
            mx = fmaxf(mx, c[i]);
        }
        st[t] = c;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (g == 0) s_max[wave * 16 + r] = mx;
    // the wave's V rows into its LDS image (row-major, as loaded)
    bf16_t* sv = s_v + wave * (32 * XM_VS);
#pragma unroll
    for (int u = 0; u < 4; ++u) *(xm_u4*)(sv + ((lane >> 3) + 8 * u) * XM_VS + (lane & 7) * 8) = vr[u];
    __syncthreads();
    {
        float m = s_max[r];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, s_max[w * 16 + r]);
        mx = m;
    }
    const int slot = p.align_out ? p.align_slot[h] : -1;
    float pr[8], lsum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kb + t * 16 + g * 4 + i;
            const float pk = (k < nk) ? expf(st[t][i] - mx) : 0.f;
            pr[t * 4 + i] = pk;
            lsum += pk;
        }
    if (slot >= 0 && r < nq) {                                    // un-normalised; align_normalize_kernel finishes the row
        const size_t rowi = ((size_t)(b0 + r) * p.n_align + slot) * p.align_rows + p.pos[b0 + r];
        float* ao = p.align_out + rowi * p.n_keys + k_lo;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb + t * 16 + g * 4 + i;
                if (k < nk) ao[k] = pr[t * 4 + i];
            }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (g == 0) red_l[wave * 16 + r] = lsum;
    // O^T = V^T P^T: B fragment = the probabilities this lane already holds (k index g*8 + j <-> key g*4 + j of tile j / 4)
    bf16x8_t ph, pm, pl;
    split3(pr, ph, pm, pl);
    f32x4_t oc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        bf16x8_t a;
        if (TR) {
            // a 16-lane group passes 4 rows x 16 columns (lane i: row i / 4, columns 4 (i % 4) .. +3) and receives column i of them
            const bf16_t* a0 = sv + (g * 4 + (r >> 2)) * XM_VS + dt * 16 + (r & 3) * 4;
            const xm_v4s v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xm_v4s __attribute__((address_space(3)))*)a0);
            const xm_v4s v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xm_v4s __attribute__((address_space(3)))*)(a0 + 16 * XM_VS));
            a = (bf16x8_t){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (short)sv[(g * 4 + j) * XM_VS + dt * 16 + r];
                a[4 + j] = (short)sv[(16 + g * 4 + j) * XM_VS + dt * 16 + r];
            }
        }
        f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        c = cw_mfma_16x16x32(a, pl, c);
        c = cw_mfma_16x16x32(a, pm, c);
        c = cw_mfma_16x16x32(a, ph, c);
        oc[dt] = c;                                               // query r, dims dt*16 + g*4 + i
    }
    // partial outputs of the wave over its own V image (wave-private; the LDS pipe keeps a wave's accesses in order)
    float* red = (float*)sv;
    if (r < nq) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(f32x4_t*)(red + r * 64 + dt * 16 + g * 4) = oc[dt];
    }
    __syncthreads();
    for (int i = tid; i < nq * 64; i += CROSS_THREADS) {
        const int q = i >> 6, c = i & 63;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += ((const float*)(s_v + w * (32 * XM_VS)))[q * 64 + c];
        p.part_o[((size_t)sp * p.B + b0 + q) * D + h * 64 + c] = acc;
    }
    if (tid < nq) {                                               // lane tid of wave 0: query r = tid, mx is that query's maximum
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) l += red_l[w * 16 + tid];
        float* ml = p.part_ml + (((size_t)(b0 + tid) * p.H + h) * ATT_NS + sp) * 2;
        ml[0] = mx; ml[1] = l;
        if (slot >= 0) {
            const size_t rowi = ((size_t)(b0 + tid) * p.n_align + slot) * p.align_rows + p.pos[b0 + tid];
            p.align_ml[(rowi * ATT_NS + sp) * 2] = mx; p.align_ml[(rowi * ATT_NS + sp) * 2 + 1] = l;
        }
    }
}

template <int NQ>
static void launch_cross_split(bool bf16, const CrossSplitParams& p, hipStream_t st) {
    dim3 grid(p.H, p.B / NQ, ATT_NS);
    if (bf16) hipLaunchKernelGGL((attn_cross_split_kernel<bf16_t, NQ, false>), grid, dim3(CROSS_THREADS), 0, st, p);
    else hipLaunchKernelGGL((attn_cross_split_kernel<float, NQ, false>), grid, dim3(CROSS_THREADS), 0, st, p);
}

static int g_cross_valu = 0;
void cw_cross_set_valu(int on) { g_cross_valu = on; }
static int g_cross_per_row = 0;   // test option "cross_per_row" (same meaning as CW_CROSS_PER_ROW=1, switchable inside a process)
void cw_cross_set_per_row(int on) { g_cross_per_row = on; }
int cw_launch_attn_cross_split(bool bf16, const CrossSplitParams& p, hipStream_t st) {
    if ((p.n_keys + ATT_NS - 1) / ATT_NS > 4 * (CROSS_THREADS / 8) || CROSS_THREADS != 512) return CW_ERR_INVALID;
    // every key split must own at least one key (the kernels clamp their loads to the split's last key)
    if (p.n_keys < 1 || (ATT_NS - 1) * ((p.n_keys + ATT_NS - 1) / ATT_NS) >= p.n_keys) return CW_ERR_INVALID;
#ifndef CW_EXPERIMENTS
    if (p.a_frag || p.a_out) return CW_ERR_INVALID;   // the full-key kernel is an A/B build (-DCW_EXPERIMENTS)
#else
    if (p.a_frag) {   // 17..64 greedy rows: one block per (row, head) writes the out-projection's 16-bit rows -- no partials, no combine
        if (!bf16 || p.kv_div > 1 || p.H > 20 || p.n_keys > CROSSF_U * (CROSS_THREADS / 8)) return CW_ERR_INVALID;
        if (p.xstat ? (!p.qa || !p.qw || !p.qbias || p.q_planes < 1) : !p.q) return CW_ERR_INVALID;
        hipLaunchKernelGGL((attn_cross_full_kernel<bf16_t>), dim3(p.H, p.B), dim3(CROSS_THREADS), 0, st, p);
        return CW_OK;
    }
#endif
    if (p.pstats && p.kv_div > 1) {   // fused stage under beam search: the hypotheses of an item in one block, query finished in the kernel
        if (!bf16 || p.kv_div > 16 || p.B % p.kv_div || !p.qa || !p.qb || !p.qw || !p.qbias || p.xstat || p.q_planes > 1) return CW_ERR_INVALID;
        if (p.n_pstats < 1 || p.n_pstats > 96 || p.B > 64 || (p.n_keys + ATT_NS - 1) / ATT_NS > 256) return CW_ERR_INVALID;
        hipLaunchKernelGGL((attn_cross_mfma_kernel<true, true>), dim3(p.H, p.B / p.kv_div, ATT_NS), dim3(CROSS_THREADS), 0, st, p, p.kv_div);
        return CW_OK;
    }
    if (p.xstat || p.pstats) {   // fused out-projection / query stage: the query is finished in the kernel; 16-bit caches
        if (!bf16 || p.kv_div > 1 || p.H > 20 || !p.qa || !p.qb || !p.qw || !p.qbias || p.q_planes > 1) return CW_ERR_INVALID;
#ifdef CW_EXPERIMENTS
        if (p.a_out) {
            if (!p.xstat) return CW_ERR_INVALID;   // one block per (row, head), finished output (A/B: 17 us per launch against 12 with six key splits)
            if (p.n_keys > CROSSF_U * (CROSS_THREADS / 8)) return CW_ERR_INVALID;
            hipLaunchKernelGGL((attn_cross_full_kernel<bf16_t>), dim3(p.H, p.B), dim3(CROSS_THREADS), 0, st, p);
            return CW_OK;
        }
#endif
        if (!p.pstats || p.n_pstats < 1 || p.n_pstats > 128 || p.B > 64) return CW_ERR_INVALID;
        // A/B (CW_CROSS_LDS_PAD=bytes): unused dynamic LDS that limits how many blocks share a CU (the 960 blocks of a B = 8 launch
        // are all resident at once: their tails -- softmax, V pass, reductions -- then run together after the last byte landed)
        const int lds_pad = cw_sw::cw_switches().cross_lds_pad;
        if (lds_pad > 0) {
            static std::once_flag attr;
            std::call_once(attr, [] { (void)hipFuncSetAttribute((const void*)attn_cross_split_kernel<bf16_t, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
        }
        hipLaunchKernelGGL((attn_cross_split_kernel<bf16_t, 1, true>), dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), (size_t)lds_pad, st, p);
        return CW_OK;
    }
    const bool per_row = cw_sw::cw_switches().cross_per_row || g_cross_per_row;   // A/B: one block per row even under beam search
    const int valu = cw_sw::cw_switches().cross_valu; // A/B: instruction-bound 8-lane-group kernel for 2..6 rows per K/V
    const bool no_tr = cw_sw::cw_switches().cross_no_tr; // A/B: 2-byte LDS reads instead of the transposing read
    if (bf16 && p.kv_div > 1 && p.kv_div <= 16 && p.B % p.kv_div == 0 && !per_row && !(valu || g_cross_valu) &&
        (p.n_keys + ATT_NS - 1) / ATT_NS <= 256) {
        dim3 grid(p.H, p.B / p.kv_div, ATT_NS);
        if (no_tr) hipLaunchKernelGGL((attn_cross_mfma_kernel<false>), grid, dim3(CROSS_THREADS), 0, st, p, p.kv_div);
        else hipLaunchKernelGGL((attn_cross_mfma_kernel<true>), grid, dim3(CROSS_THREADS), 0, st, p, p.kv_div);
        return CW_OK;
    }
    // A/B (CW_CROSS_MFMA1=1): rows that share nothing (greedy, 17..64 rows) through the matrix-core kernel as well, one row per block
    if (bf16 && p.kv_div <= 1 && cw_sw::cw_switches().cross_mfma1 && (p.n_keys + ATT_NS - 1) / ATT_NS <= 256) {
        CrossSplitParams p1 = p; p1.kv_div = 1;
        hipLaunchKernelGGL((attn_cross_mfma_kernel<true>), dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), 0, st, p1, 1);
        return CW_OK;
    }
    const int nq = (p.kv_div > 1 && p.kv_div <= 6 && p.B % p.kv_div == 0 && !per_row) ? p.kv_div : 1;
    switch (nq) {
        case 2: launch_cross_split<2>(bf16, p, st); break;
        case 3: launch_cross_split<3>(bf16, p, st); break;
        case 4: launch_cross_split<4>(bf16, p, st); break;
        case 5: launch_cross_split<5>(bf16, p, st); break;
        case 6: launch_cross_split<6>(bf16, p, st); break;
        default: launch_cross_split<1>(bf16, p, st); break;
    }
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// Opt-in fp8 (OCP e4m3) cross-attention cache.  The cross K/V stream is the largest byte mover of the decode step
// (245.76 MB per sequence per step in bf16); storing it in e4m3 with one f32 scale per (batch, head, K|V) halves it.
// Quantisation runs once per encoded window (kv_quant_fp8_kernel, after the bf16 projection); the split kernel keeps
// the structure of attn_cross_split_kernel with 4 lanes x 16 elements per key row.
// ---------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define X8_SPLIT_BYTES 16384   // fragment-major V of one (batch, head, key split): 8 waves x 64 lanes x 4 dim tiles x 8 keys

// vfrag: V is written in the order attn_cross_mfma8_kernel's B fragments read it -- per (batch, head, key split) X8_SPLIT_BYTES
// bytes [wave 8][half 2][lane 64][2 dim tiles][8 keys]: lane (j, g) of wave w holds, for dim 16 dt + j (dt = 2 half + 0 | 1),
// keys 32 w + 8 g .. + 7 of the split (zeros past the split's last key), so a wave's V operand is two fully used contiguous
// 1 KB reads (16 B per lane each) and needs no transposition.
__global__ __launch_bounds__(512) void kv_quant_fp8_kernel(const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                           unsigned char* __restrict__ K8, unsigned char* __restrict__ V8,
                                                           float* __restrict__ kv_scale, int S, int vfrag) {
    __shared__ float scratch[64];
    const int h = blockIdx.x, b = blockIdx.y, which = blockIdx.z, H = gridDim.x;
    const size_t base = ((size_t)b * H + h) * S * 64;
    const bf16_t* src = (which ? V : K) + base;
    unsigned char* dst = (which ? V8 : K8) + base;
    const int nvec = S * 8;                                    // uint4 = 8 bf16
    float amax = 0.f;
    for (int i = threadIdx.x; i < nvec; i += 512) {
        float v[8];
        Row8<bf16_t>::ld(src + (size_t)i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    }
    amax = block_max(amax, scratch);
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;     // e4m3 max finite = 448
    const float inv = 1.0f / scale;
    if (threadIdx.x == 0) kv_scale[((size_t)b * H + h) * 2 + which] = scale;
    if (which == 1 && vfrag) {
        // per key split: the rows are quantised from coalesced 16-byte reads and their bytes scattered into a transposed LDS
        // image [dim 64][key 256 (+8)], from which every 8-byte unit of the fragment order is one aligned LDS read
        __shared__ __attribute__((aligned(8))) unsigned char tr[64 * 264];
        unsigned char* dstf = V8 + ((size_t)b * H + h) * (ATT_NS * X8_SPLIT_BYTES);
        const int per = (S + ATT_NS - 1) / ATT_NS;
        for (int sp = 0; sp < ATT_NS; ++sp) {
            const int k0 = sp * per, nk = min(S, k0 + per) - k0;
            __syncthreads();                                      // the previous split's image has been read
            for (int i = threadIdx.x; i < 256 * 8; i += 512) {
                const int k = i >> 3, o = i & 7;
                unsigned lo = 0u, hi = 0u;                        // zeros behind the split's last key
                if (k < nk) {
                    float v[8];
                    Row8<bf16_t>::ld(src + (size_t)(k0 + k) * 64 + o * 8, v);
                    int t = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
                    lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, t, true);
                    t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
                    hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, t, true);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    tr[(8 * o + e) * 264 + k] = (unsigned char)(lo >> (8 * e));
                    tr[(8 * o + 4 + e) * 264 + k] = (unsigned char)(hi >> (8 * e));
                }
            }
            __syncthreads();
            for (int u = threadIdx.x; u < X8_SPLIT_BYTES / 8; u += 512) {
                const int dt = ((u >> 7) & 1) * 2 + (u & 1), ln = (u >> 1) & 63, w = (u >> 8) & 7;
                const int j = ln & 15, g = ln >> 4;
                *(uint2*)(dstf + ((size_t)sp * (X8_SPLIT_BYTES / 8) + u) * 8) = *(const uint2*)&tr[(16 * dt + j) * 264 + 32 * w + 8 * g];
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < nvec; i += 512) {
        float v[8];
        Row8<bf16_t>::ld(src + (size_t)i * 8, v);
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        *(uint2*)(dst + (size_t)i * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// the e4m3 cross-attention runs on the fp8 matrix cores and reads V in fragment-major order unless CW_CROSS8_VALU=1 (round-2/4
// VALU kernel over a row-major V, A/B); the quantiser writes the layout the attention kernel of this process reads
static bool cross8_mfma(int n_keys) { return !cw_sw::cw_switches().cross8_valu && (n_keys + ATT_NS - 1) / ATT_NS <= 256; }
bool cw_cross8_is_mfma(int n_keys) { return cross8_mfma(n_keys); }   // the engine composes the fused query stage only with the matrix-core kernel
size_t cw_kv8_v_bytes(int H, int S) {   // bytes of the e4m3 V cache per batch item (either layout fits)
    const size_t row_major = (size_t)H * S * 64, frag = (size_t)H * ATT_NS * X8_SPLIT_BYTES;
    return row_major > frag ? row_major : frag;
}
int cw_launch_kv_quant_fp8(const void* K, const void* V, void* K8, void* V8, float* kv_scale, int B, int H, int S,
                           hipStream_t st) {
    if (S < 1 || (ATT_NS - 1) * ((S + ATT_NS - 1) / ATT_NS) >= S) return CW_ERR_INVALID;
    hipLaunchKernelGGL(kv_quant_fp8_kernel, dim3(H, B, 2), dim3(512), 0, st, (const bf16_t*)K, (const bf16_t*)V,
                       (unsigned char*)K8, (unsigned char*)V8, kv_scale, S, cross8_mfma(S) ? 1 : 0);
    return CW_OK;
}

__device__ inline void fp8x16_to_f32(const uint4& r, float* o) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
        const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
        o[4 * i] = a[0]; o[4 * i + 1] = a[1]; o[4 * i + 2] = c[0]; o[4 * i + 3] = c[1];
    }
}

#define CROSS8_GROUPS (CROSS_THREADS / 4)
#define C8U 2             // key rows per lane: 2 x 128 groups = 256 slots for the 250 keys of a 6-way split
__device__ inline float row_ror4_add(float v) { return v + dpp_mov<0x124, 0xf>(0.f, v); }   // + lane + 4 within the row of 16 (row_ror:4)
// Round 4: the round-2 structure of attn_cross_split_kernel on e4m3 rows.  A block moves only 32 KB, so its time is its latency
// chain, not its bytes: every K and V row is requested up front (one memory round trip), a lane keeps the scores of its two
// keys in registers through the K pass, the exponentials and the V pass (no score goes through LDS; alignment heads write
// theirs straight to the alignment buffer), and the only block-wide exchanges are the running maximum (8 floats) and the
// 8 x 64 partial outputs: 2 barriers per block instead of 7, the 128-group serial LDS sum became four cross-lane steps + 8
// adds.  Needs nk <= 2 * 128 keys per block.  profiles/r04_cross_fp8_latency.txt
// NSB = key splits per block.  With many rows (B x H x ATT_NS blocks >> what is resident) the launch is bound by the bytes a CU
// has in flight -- resident blocks x bytes per block / block lifetime -- and a 32 KB block spends most of its lifetime in its
// fixed chain: NSB = 2 lets one block carry two adjacent splits (64 KB requested up front, the same two barriers), each split
// computed exactly as a block of its own would (bit-identical partials).  grid (H, B, ATT_NS / NSB).
template <int NSB>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_split_fp8_kernel(CrossSplitParams p) {
    __shared__ float s_max[NSB * 8];
    __shared__ __attribute__((aligned(16))) float red[NSB * 8 * 64];
    __shared__ float red_l[NSB * 8];
    const int h = blockIdx.x, b = blockIdx.y, sp0 = blockIdx.z * NSB;
    const int per = (p.n_keys + ATT_NS - 1) / ATT_NS;
    const int tid = threadIdx.x, lane = tid & 63, sub = tid & 3, grp = tid >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int G = CROSS8_GROUPS;
    const int bk = p.kv_div > 1 ? b / p.kv_div : b;
    const unsigned char* Kb = (const unsigned char*)p.K + ((size_t)bk * p.H + h) * p.n_keys * 64 + sub * 16;
    const unsigned char* Vb = (const unsigned char*)p.V + ((size_t)bk * p.H + h) * p.n_keys * 64 + sub * 16;
    int k_lo[NSB], nk[NSB];
    uint4 kr[NSB][C8U], vr[NSB][C8U];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        k_lo[s] = (sp0 + s) * per;
        nk[s] = min(p.n_keys, k_lo[s] + per) - k_lo[s];
#pragma unroll
        for (int u = 0; u < C8U; ++u) kr[s][u] = *(const uint4*)(Kb + (size_t)(k_lo[s] + min(grp + u * G, nk[s] - 1)) * 64);   // unconditional, clamped
    }
#pragma unroll
    for (int s = 0; s < NSB; ++s)
#pragma unroll
        for (int u = 0; u < C8U; ++u) vr[s][u] = *(const uint4*)(Vb + (size_t)(k_lo[s] + min(grp + u * G, nk[s] - 1)) * 64);
    const float ks = p.kv_scale[((size_t)bk * p.H + h) * 2], vs = p.kv_scale[((size_t)bk * p.H + h) * 2 + 1];
    float qv[16];
    Row8<float>::ld(p.q + (size_t)b * p.H * 64 + h * 64 + sub * 16, qv);
    Row8<float>::ld(p.q + (size_t)b * p.H * 64 + h * 64 + sub * 16 + 8, qv + 8);

    float d[NSB][C8U], mx[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        mx[s] = -INFINITY;
#pragma unroll
        for (int u = 0; u < C8U; ++u) {
            float kv[16];
            fp8x16_to_f32(kr[s][u], kv);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) t = fmaf(qv[e], kv[e], t);
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
            d[s][u] = (grp + u * G < nk[s]) ? t * ks : -INFINITY;
            mx[s] = fmaxf(mx[s], d[s][u]);
        }
        mx[s] = wave_max(mx[s]);
        if (lane == 0) s_max[s * 8 + wave] = mx[s];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        float m = s_max[s * 8];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, s_max[s * 8 + w]);
        mx[s] = m;
    }

    const int slot = p.align_out ? p.align_slot[h] : -1;
    const size_t rowi = slot >= 0 ? ((size_t)b * p.n_align + slot) * p.align_rows + p.pos[b] : 0;
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        float acc[16] = {};
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < C8U; ++u) {
            const int k = grp + u * G;
            float vv[16];
            fp8x16_to_f32(vr[s][u], vv);
            const float pk = (k < nk[s]) ? expf(d[s][u] - mx[s]) : 0.f;
            if (sub == 0 && k < nk[s]) {
                lsum += pk;
                if (slot >= 0) p.align_out[rowi * p.n_keys + k_lo[s] + k] = pk;   // un-normalised; align_normalize_kernel finishes the row
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = fmaf(pk, vv[e], acc[e]);
        }
        // sum over the wave's 16 key groups (lane bits 2..5)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = xor32_sum(xor16_sum(row_ror8_add(row_ror4_add(acc[e]))));
        lsum = wave_sum(lsum);
        if (lane < 4) {
#pragma unroll
            for (int e = 0; e < 16; e += 4) *(float4*)&red[(s * 8 + wave) * 64 + sub * 16 + e] = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
        }
        if (lane == 0) red_l[s * 8 + wave] = lsum;
    }
    __syncthreads();
    if (tid < 64 * NSB) {
        const int s = tid >> 6, c = tid & 63;
        float r = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) r += red[(s * 8 + w) * 64 + c];
        p.part_o[((size_t)(sp0 + s) * p.B + b) * p.H * 64 + h * 64 + c] = r * vs;
    } else if (tid < 64 * NSB + NSB) {
        const int s = tid - 64 * NSB;
        float l = 0.f, m = s_max[s * 8];
#pragma unroll
        for (int w = 0; w < 8; ++w) { l += red_l[s * 8 + w]; m = fmaxf(m, s_max[s * 8 + w]); }
        float* ml = p.part_ml + (((size_t)b * p.H + h) * ATT_NS + sp0 + s) * 2;
        ml[0] = m; ml[1] = l;
        if (slot >= 0) { p.align_ml[(rowi * ATT_NS + sp0 + s) * 2] = m; p.align_ml[(rowi * ATT_NS + sp0 + s) * 2 + 1] = l; }
    }
}

// ---------------------------------------------------------------------------------------------------
// e4m3 cross-attention on the fp8 matrix cores (v_mfma_f32_16x16x32_fp8_fp8; operand layout probed in
// tools/probe/mfma_fp8_layout.hip).  grid (H, B, ATT_NS), 512 threads; a wave owns 32 consecutive keys of the split.
// The VALU kernel above spends 16 conversions + 32 multiply-adds per key and lane plus 64 cross-lane adds per block; here the
// cache bytes go to the matrix pipe as they come from HBM and nothing is converted:
//   * S = Q K^T: A = the query (one row of work, so rows 0..2 of the 16 carry the query as THREE e4m3 terms
//     q c1 = t0 + t1 / 16 + t2 / 256, residuals exact in f32: 12 significant bits, the rows are weighted after the product),
//     B = the K rows as loaded (lane (key j, g): dims 16 g .. + 15, two MFMAs over the permuted contraction index);
//     lane (j, 0) then holds the three row sums of key j -- no cross-lane add;
//   * O = P V: A = the probabilities (x 256) as three e4m3 terms again (through a wave-private 32-float LDS row: the score
//     fragment has key j in lane j, the A fragment wants keys 8 g .. + 7 per lane), B = V in the fragment-major order the
//     quantiser wrote (one contiguous 32-byte read per lane): four MFMAs, lane (j, 0) holds dim 16 dt + j;
//   * the running maximum (8 floats), the partial planes, (m, l) pairs and alignment rows are those of the VALU kernel.
// ---------------------------------------------------------------------------------------------------
__device__ inline unsigned pk4_e4m3(float a, float b, float c, float d) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
__device__ inline void unpk4_e4m3(unsigned w, float* o) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), c = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    o[0] = a[0]; o[1] = a[1]; o[2] = c[0]; o[3] = c[1];
}
// x[0..7] (|x| <= 448) = t0 + t1 / 16 + t2 / 256 with t_i on the e4m3 grid; returns term `term` (0..2) as 8 bytes, 0 otherwise.
// Contraction is off in here: x is usually a product (query x scale), and hipcc fuses `x - t0` with that product in one inlining
// context and not in the next (attn_cross_mfma8_rows_kernel got fma(q, c, -t0), the one-row kernel v_mul + v_sub: residuals one
// rounding apart, scores a few units in the last place apart -- found by the bit-identity test of the two kernels).
__device__ inline long split3_e4m3(const float* x, int term) {
#pragma clang fp contract(off)
    float r[8], f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = x[e];
    uint2 t[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[s].x = pk4_e4m3(r[0], r[1], r[2], r[3]);
        t[s].y = pk4_e4m3(r[4], r[5], r[6], r[7]);
        if (s < 2) {
            unpk4_e4m3(t[s].x, f); unpk4_e4m3(t[s].y, f + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (r[e] - f[e]) * 16.f;
        }
    }
    const uint2 sel = term == 0 ? t[0] : term == 1 ? t[1] : term == 2 ? t[2] : make_uint2(0u, 0u);
    return (long)(((unsigned long)sel.y << 32) | sel.x);
}
// NSB = key splits per block (grid (H, B, ATT_NS / NSB)): with thousands of blocks the launch is bound by the bytes a CU has in
// flight; two adjacent splits per block double them (a wave owns 32 keys of each) under the same three barriers, and every
// split is computed exactly as a block of its own would (bit-identical partials).
// FUSED (fused out-projection / cross-query stage in front, decfuse.hip): wave 0 finishes the query exactly as
// attn_cross_split_kernel<T, 1, true> does -- lane c = column c of the head, q = rstd(x1) (qa + qb - mean(x1) qw) + qbias from the
// per-tile LayerNorm partial sums the producing GEMV left behind -- and hands the 16 columns of each lane over through LDS.
// FUSED = 2 (17..64 rows behind the fused stage: thousands of blocks, a throughput problem): wave 0 alone fetches the six small
// operands and finishes the query, the other waves pick the fragments up behind the block barrier like the plain kernel's -- same
// arithmetic as FUSED = 1 (bit-identical), an eighth of its small loads.
template <int NSB, int FUSED>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_mfma8_kernel(CrossSplitParams p) {
    __shared__ float s_max[NSB * 8];
    __shared__ __attribute__((aligned(16))) float s_qf[FUSED ? 8 * 64 : 1];   // FUSED: a wave-private row each
    __shared__ __attribute__((aligned(16))) float red[NSB * 8 * 64];
    __shared__ float red_l[NSB * 8];
    __shared__ __attribute__((aligned(16))) float s_p[8 * 32];
    __shared__ long s_aq[2 * 64];
    __shared__ float s_c1;
    const int h = blockIdx.x, b = blockIdx.y, sp0 = blockIdx.z * NSB;
    const int per = (p.n_keys + ATT_NS - 1) / ATT_NS;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave * 32;
    const int D = p.H * 64;
    const int bk = p.kv_div > 1 ? b / p.kv_div : b;
    const size_t bh = (size_t)bk * p.H + h;
    // FUSED: every wave finishes the query itself (no block barrier in front of the first MFMA); its six small loads (256 B per
    // wave instruction) go out before the K / V rows so that they come back first
    float2 pt0 = make_float2(0.f, 0.f), pt1 = pt0;
    float qa1 = 0.f, qb1 = 0.f, qw1 = 0.f, qc1 = 0.f;
    if (FUSED == 1 || (FUSED == 2 && wave == 0)) {
        const float* ps = p.pstats + (size_t)(b >> 4) * p.n_pstats * 32;    // one [tiles][16][2] plane per group of 16 rows (gemv_stack_kernel)
        pt0 = *(const float2*)(ps + ((size_t)min(lane, p.n_pstats - 1) * 16 + (b & 15)) * 2);
        pt1 = *(const float2*)(ps + ((size_t)min(lane + 64, p.n_pstats - 1) * 16 + (b & 15)) * 2);
        const size_t col = (size_t)h * 64 + lane;
        qa1 = p.qa[(size_t)b * D + col]; qb1 = p.qb[(size_t)b * D + col]; qw1 = p.qw[col]; qc1 = p.qbias[col];
    }
    // every load of the block first
    int k_lo[NSB], nk[NSB];
    uint4 kf[NSB][2], vf[NSB][2];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        k_lo[s] = (sp0 + s) * per;
        nk[s] = min(p.n_keys, k_lo[s] + per) - k_lo[s];
        const unsigned char* Kh = (const unsigned char*)p.K + (bh * p.n_keys + k_lo[s]) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) kf[s][t] = *(const uint4*)(Kh + (size_t)min(kb + t * 16 + r, nk[s] - 1) * 64 + g * 16);
    }
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        const unsigned char* Vf = (const unsigned char*)p.V + (bh * ATT_NS + sp0 + s) * X8_SPLIT_BYTES + (size_t)wave * 2048 + lane * 16;
        vf[s][0] = *(const uint4*)Vf; vf[s][1] = *(const uint4*)(Vf + 1024);
    }
    const float ks = p.kv_scale[bh * 2], vs = p.kv_scale[bh * 2 + 1];
    // block-uniform values of the alignment capture, requested HERE with everything else (round 6): at their use hipcc fetches each by a
    // uniform-address vector load + `s_waitcnt vmcnt(0)` + v_readfirstlane right behind the first barrier -- in EVERY block (align_out
    // is set whenever the engine decodes): one L2 round trip on the block's critical path, and the wait drains the V rows as well
    const int slot = p.align_out ? p.align_slot[h] : -1;
    const int apos = (p.align_out && p.pos) ? p.pos[b] : 0;
    // the query fragments are the same for all eight waves: wave 0 loads the row (16 B-per-lane loads cost the CU's address
    // unit 16 clocks each whatever they fetch), scales it to the e4m3 range and splits it; the others pick the 16 bytes up from
    // LDS behind a barrier that their own K / V loads are in flight across
    long aq0, aq1;
    float c1;
    if (FUSED == 1) {
        __builtin_amdgcn_sched_barrier(0);                        // the K / V requests are out before the first wait
        const float inv_d = 1.0f / (float)D;
        const float ps1 = (lane < p.n_pstats ? pt0.x : 0.f) + (lane + 64 < p.n_pstats ? pt1.x : 0.f);
        const float ps2 = (lane < p.n_pstats ? pt0.y : 0.f) + (lane + 64 < p.n_pstats ? pt1.y : 0.f);
        const float mean = wave_sum(ps1) * inv_d;
        const float var = fmaxf(wave_sum(ps2) * inv_d - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        float* sq = s_qf + wave * 64;                              // wave-private: ordered by the wave's own lgkmcnt
        sq[lane] = ((qa1 + qb1) - mean * qw1) * rstd + qc1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float qf[16];
        const float* qp = sq + g * 16;
        const float4 q0 = *(const float4*)qp, q1 = *(const float4*)(qp + 4), q2 = *(const float4*)(qp + 8), q3 = *(const float4*)(qp + 12);
        qf[0] = q0.x; qf[1] = q0.y; qf[2] = q0.z; qf[3] = q0.w; qf[4] = q1.x; qf[5] = q1.y; qf[6] = q1.z; qf[7] = q1.w;
        qf[8] = q2.x; qf[9] = q2.y; qf[10] = q2.z; qf[11] = q2.w; qf[12] = q3.x; qf[13] = q3.y; qf[14] = q3.z; qf[15] = q3.w;
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(qf[e]));
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        c1 = am > 0.f ? 448.0f / am : 1.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) qf[e] *= c1;
        aq0 = split3_e4m3(qf, r);
        aq1 = split3_e4m3(qf + 8, r);
    } else {
    if (wave == 0) {
        float qf[16];
        if (FUSED == 2) {                                          // the query row from the fused stage's summands, through wave 0's LDS row
            const float inv_d = 1.0f / (float)D;
            const float ps1 = (lane < p.n_pstats ? pt0.x : 0.f) + (lane + 64 < p.n_pstats ? pt1.x : 0.f);
            const float ps2 = (lane < p.n_pstats ? pt0.y : 0.f) + (lane + 64 < p.n_pstats ? pt1.y : 0.f);
            const float mean = wave_sum(ps1) * inv_d;
            const float var = fmaxf(wave_sum(ps2) * inv_d - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            s_qf[lane] = ((qa1 + qb1) - mean * qw1) * rstd + qc1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const float* qp = FUSED == 2 ? s_qf + g * 16 : p.q + (size_t)b * D + h * 64 + g * 16;
        const float4 q0 = *(const float4*)qp, q1 = *(const float4*)(qp + 4), q2 = *(const float4*)(qp + 8), q3 = *(const float4*)(qp + 12);
        qf[0] = q0.x; qf[1] = q0.y; qf[2] = q0.z; qf[3] = q0.w; qf[4] = q1.x; qf[5] = q1.y; qf[6] = q1.z; qf[7] = q1.w;
        qf[8] = q2.x; qf[9] = q2.y; qf[10] = q2.z; qf[11] = q2.w; qf[12] = q3.x; qf[13] = q3.y; qf[14] = q3.z; qf[15] = q3.w;
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(qf[e]));
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float cq = am > 0.f ? 448.0f / am : 1.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) qf[e] *= cq;
        s_aq[lane] = split3_e4m3(qf, r);
        s_aq[64 + lane] = split3_e4m3(qf + 8, r);
        if (lane == 0) s_c1 = cq;
    }
    __syncthreads();
    aq0 = s_aq[lane]; aq1 = s_aq[64 + lane];
    c1 = s_c1;
    }
    const float s_unscale = ks / c1;
    float sc[NSB][2], mx[NSB];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        mx[s] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq0, (long)(((unsigned long)kf[s][t].y << 32) | kf[s][t].x), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq1, (long)(((unsigned long)kf[s][t].w << 32) | kf[s][t].z), c, 0, 0, 0);
            const float v = ((c[2] * 0.0625f + c[1]) * 0.0625f + c[0]) * s_unscale;      // small terms first
            sc[s][t] = (g == 0 && kb + t * 16 + r < nk[s]) ? v : -INFINITY;
            mx[s] = fmaxf(mx[s], sc[s][t]);
        }
        mx[s] = wave_max(mx[s]);
        if (lane == 0) s_max[s * 8 + wave] = mx[s];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        float m = s_max[s * 8];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, s_max[s * 8 + w]);
        mx[s] = m;
    }

    const size_t rowi = slot >= 0 ? ((size_t)b * p.n_align + slot) * p.align_rows + apos : 0;
    float* spw = s_p + wave * 32;
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
        float pk[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = kb + t * 16 + r;
            pk[t] = (g == 0 && k < nk[s]) ? expf(sc[s][t] - mx[s]) : 0.f;
            if (slot >= 0 && g == 0 && k < nk[s]) p.align_out[rowi * p.n_keys + k_lo[s] + k] = pk[t];   // un-normalised; align_normalize_kernel finishes the row
        }
        const float lsum = wave_sum(pk[0] + pk[1]);
        // score fragment (key j in lane j) -> A fragment (lane (term, g): keys 8 g .. + 7): a wave-private LDS row
        if (s > 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // the previous split's reads are done
        if (g == 0) { spw[r] = pk[0] * 256.f; spw[16 + r] = pk[1] * 256.f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float pf[8];
        {
            const float4 p0 = *(const float4*)(spw + g * 8), p1 = *(const float4*)(spw + g * 8 + 4);
            pf[0] = p0.x; pf[1] = p0.y; pf[2] = p0.z; pf[3] = p0.w; pf[4] = p1.x; pf[5] = p1.y; pf[6] = p1.z; pf[7] = p1.w;
        }
        const long ap = split3_e4m3(pf, r);
        const unsigned vw[8] = {vf[s][0].x, vf[s][0].y, vf[s][0].z, vf[s][0].w, vf[s][1].x, vf[s][1].y, vf[s][1].z, vf[s][1].w};
        float o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ap, (long)(((unsigned long)vw[2 * dt + 1] << 32) | vw[2 * dt]), c, 0, 0, 0);
            o[dt] = (c[2] * 0.0625f + c[1]) * 0.0625f + c[0];
        }
        if (g == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) red[(s * 8 + wave) * 64 + dt * 16 + r] = o[dt];
        }
        if (lane == 0) red_l[s * 8 + wave] = lsum;
    }
    __syncthreads();
    if (tid < 64 * NSB) {
        const int s = tid >> 6, c = tid & 63;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += red[(s * 8 + w) * 64 + c];
        p.part_o[((size_t)(sp0 + s) * p.B + b) * D + h * 64 + c] = a * (vs * (1.0f / 256.0f));
    } else if (tid < 64 * NSB + NSB) {
        const int s = tid - 64 * NSB;
        float l = 0.f, m = s_max[s * 8];
#pragma unroll
        for (int w = 0; w < 8; ++w) { l += red_l[s * 8 + w]; m = fmaxf(m, s_max[s * 8 + w]); }
        float* ml = p.part_ml + (((size_t)b * p.H + h) * ATT_NS + sp0 + s) * 2;
        ml[0] = m; ml[1] = l;
        if (slot >= 0) { p.align_ml[(rowi * ATT_NS + sp0 + s) * 2] = m; p.align_ml[(rowi * ATT_NS + sp0 + s) * 2 + 1] = l; }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same kernel for nq = 2..8 rows that share one e4m3 K/V (the hypotheses of one audio item under beam search).
// grid (H, B / nq, ATT_NS): the block's 32 KB of cache bytes are read once for all of its rows instead of once per row out of L2.
// A one-row block uses rows 0..2 of the 16-row A operand (the three e4m3 terms of its query) and lanes (j, 0) of the result;
// here row 4 qq + term of A tile T carries term `term` of query 4 T + qq (row 4 qq + 3 is zero), so that lane (j, g) of the C
// fragment -- rows 4 g .. 4 g + 3 -- holds the three row sums of query 4 T + g against key j: every query is a 16-lane row of
// the wave, its maximum and its sum of exponentials are the row-local halves of wave_max / wave_sum (same DPP steps, same
// order), and nothing moves across lanes.  NT = tiles of four queries (5 hypotheses: two).  Per row the arithmetic is the
// one-row kernel's, operation for operation -- the MFMA rows are independent, scales, exponentials, the three-term recombination
// and the w = 0..7 partial sums are the same expressions in the same order -- so a row's partial plane, (m, l) pair and
// alignment row are bit-identical to a one-row launch (test_e4m3_cache_beam_rows_equal_one_row_blocks).
// ---------------------------------------------------------------------------------------------------
__device__ inline float row16_max(float v) {                  // lane 15 of every 16-lane row: the row's maximum (first four steps of wave_max)
    v = fmaxf(v, dpp_mov<0x111, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_mov<0x112, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_mov<0x114, 0xf>(-INFINITY, v));
    v = fmaxf(v, dpp_mov<0x118, 0xf>(-INFINITY, v));
    return v;
}
__device__ inline float row16_sum(float v) {                  // lane 15 of every 16-lane row: the row's sum (first four steps of wave_sum)
    v += dpp_mov<0x111, 0xf>(0.f, v);
    v += dpp_mov<0x112, 0xf>(0.f, v);
    v += dpp_mov<0x114, 0xf>(0.f, v);
    v += dpp_mov<0x118, 0xf>(0.f, v);
    return v;
}
template <int NT>
__global__ __launch_bounds__(CROSS_THREADS) void attn_cross_mfma8_rows_kernel(CrossSplitParams p, int nq) {
    constexpr int NQM = 4 * NT;                                   // query slots of the block
    __shared__ float s_max[8 * NQM];
    __shared__ __attribute__((aligned(16))) float red[8 * NQM * 64];
    __shared__ float red_l[8 * NQM];
    __shared__ __attribute__((aligned(16))) float s_p[8 * NQM * 32];
    __shared__ long s_aq[NT * 2 * 64];
    __shared__ float s_c1[NQM];
    const int h = blockIdx.x, b0 = blockIdx.y * nq, sp = blockIdx.z;
    const int per = (p.n_keys + ATT_NS - 1) / ATT_NS;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave * 32;
    const int D = p.H * 64;
    const size_t bh = (size_t)(b0 / p.kv_div) * p.H + h;
    const int k_lo = sp * per, nk = min(p.n_keys, k_lo + per) - k_lo;
    // every load of the block first
    uint4 kf[2], vf[2];
    {
        const unsigned char* Kh = (const unsigned char*)p.K + (bh * p.n_keys + k_lo) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) kf[t] = *(const uint4*)(Kh + (size_t)min(kb + t * 16 + r, nk - 1) * 64 + g * 16);
        const unsigned char* Vf = (const unsigned char*)p.V + (bh * ATT_NS + sp) * X8_SPLIT_BYTES + (size_t)wave * 2048 + lane * 16;
        vf[0] = *(const uint4*)Vf; vf[1] = *(const uint4*)(Vf + 1024);
    }
    const float ks = p.kv_scale[bh * 2], vs = p.kv_scale[bh * 2 + 1];
    // wave T scales and splits the four queries of tile T (lane (r, g): query 4 T + r / 4, term r % 4, dims 16 g .. + 15); the
    // others pick the fragments up from LDS behind a barrier that their own K / V loads are in flight across
    if (wave < NT) {
        const int qi = 4 * wave + (r >> 2), term = r & 3;
        const float z = qi < nq ? 1.f : 0.f;                      // slots past the last row: zero queries, never stored
        float qf[16];
        const float* qp = p.q + (size_t)(b0 + min(qi, nq - 1)) * D + h * 64 + g * 16;
        const float4 q0 = *(const float4*)qp, q1 = *(const float4*)(qp + 4), q2 = *(const float4*)(qp + 8), q3 = *(const float4*)(qp + 12);
        qf[0] = q0.x; qf[1] = q0.y; qf[2] = q0.z; qf[3] = q0.w; qf[4] = q1.x; qf[5] = q1.y; qf[6] = q1.z; qf[7] = q1.w;
        qf[8] = q2.x; qf[9] = q2.y; qf[10] = q2.z; qf[11] = q2.w; qf[12] = q3.x; qf[13] = q3.y; qf[14] = q3.z; qf[15] = q3.w;
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { qf[e] *= z; am = fmaxf(am, fabsf(qf[e])); }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float cq = am > 0.f ? 448.0f / am : 1.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) qf[e] *= cq;
        s_aq[(wave * 2) * 64 + lane] = split3_e4m3(qf, term);     // term 3: zeros
        s_aq[(wave * 2 + 1) * 64 + lane] = split3_e4m3(qf + 8, term);
        if (g == 0 && term == 0) s_c1[qi] = cq;
    }
    __syncthreads();
    const bool kval[2] = {kb + r < nk, kb + 16 + r < nk};
    float sc[NT][2], mx[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const long aq0 = s_aq[(T * 2) * 64 + lane], aq1 = s_aq[(T * 2 + 1) * 64 + lane];
        const float s_unscale = ks / s_c1[4 * T + g];
        mx[T] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq0, (long)(((unsigned long)kf[t].y << 32) | kf[t].x), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aq1, (long)(((unsigned long)kf[t].w << 32) | kf[t].z), c, 0, 0, 0);
            const float v = ((c[2] * 0.0625f + c[1]) * 0.0625f + c[0]) * s_unscale;      // small terms first
            sc[T][t] = kval[t] ? v : -INFINITY;
            mx[T] = fmaxf(mx[T], sc[T][t]);
        }
        mx[T] = row16_max(mx[T]);
        if (r == 15) s_max[wave * NQM + 4 * T + g] = mx[T];
    }
    __syncthreads();
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        float m = s_max[4 * T + g];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, s_max[w * NQM + 4 * T + g]);
        mx[T] = m;
    }

    const int slot = p.align_out ? p.align_slot[h] : -1;
    float* spw = s_p + wave * (NQM * 32);
    long ap[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const int qi = 4 * T + g;
        float pk[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) pk[t] = kval[t] ? expf(sc[T][t] - mx[T]) : 0.f;
        if (slot >= 0 && qi < nq) {                               // un-normalised; align_normalize_kernel finishes the row
            const size_t rowi = ((size_t)(b0 + qi) * p.n_align + slot) * p.align_rows + p.pos[b0 + qi];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (kval[t]) p.align_out[rowi * p.n_keys + k_lo + kb + t * 16 + r] = pk[t];
        }
        const float lsum = row16_sum(pk[0] + pk[1]);
        if (r == 15) red_l[wave * NQM + qi] = lsum;
        // score fragment (query g, key r) -> A fragment (lane (r, g): query r / 4, term r % 4, keys 8 g .. + 7): a wave-private LDS row per query
        spw[qi * 32 + r] = pk[0] * 256.f; spw[qi * 32 + 16 + r] = pk[1] * 256.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const float* pp = spw + (4 * T + (r >> 2)) * 32 + g * 8;
        const float4 p0 = *(const float4*)pp, p1 = *(const float4*)(pp + 4);
        const float pf[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        ap[T] = split3_e4m3(pf, r & 3);
    }
    const unsigned vw[8] = {vf[0].x, vf[0].y, vf[0].z, vf[0].w, vf[1].x, vf[1].y, vf[1].z, vf[1].w};
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4_t c = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ap[T], (long)(((unsigned long)vw[2 * dt + 1] << 32) | vw[2 * dt]), c, 0, 0, 0);
            red[(wave * NQM + 4 * T + g) * 64 + dt * 16 + r] = (c[2] * 0.0625f + c[1]) * 0.0625f + c[0];
        }
    __syncthreads();
    for (int i = tid; i < nq * 64; i += CROSS_THREADS) {
        const int q = i >> 6, c = i & 63;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) a += red[(w * NQM + q) * 64 + c];
        p.part_o[((size_t)sp * p.B + b0 + q) * D + h * 64 + c] = a * (vs * (1.0f / 256.0f));
    }
    if (tid < nq) {
        const int q = tid;
        float l = 0.f, m = s_max[q];
#pragma unroll
        for (int w = 0; w < 8; ++w) { l += red_l[w * NQM + q]; m = fmaxf(m, s_max[w * NQM + q]); }
        float* ml = p.part_ml + (((size_t)(b0 + q) * p.H + h) * ATT_NS + sp) * 2;
        ml[0] = m; ml[1] = l;
        if (slot >= 0) {
            const size_t rowi = ((size_t)(b0 + q) * p.n_align + slot) * p.align_rows + p.pos[b0 + q];
            p.align_ml[(rowi * ATT_NS + sp) * 2] = m; p.align_ml[(rowi * ATT_NS + sp) * 2 + 1] = l;
        }
    }
}

int cw_launch_attn_cross_split_fp8(const CrossSplitParams& p, hipStream_t st) {
    if ((p.n_keys + ATT_NS - 1) / ATT_NS > C8U * CROSS8_GROUPS || !p.kv_scale) return CW_ERR_INVALID;
    if (p.n_keys < 1 || (ATT_NS - 1) * ((p.n_keys + ATT_NS - 1) / ATT_NS) >= p.n_keys) return CW_ERR_INVALID;   // a split without a key
    if (cross8_mfma(p.n_keys)) {
        // rows that share one cache (beam search): one block per (item, head, split) takes all of them (A/B: CW_CROSS_PER_ROW=1)
        if (!p.qa && p.kv_div > 1 && p.kv_div <= 8 && p.B % p.kv_div == 0 && !cw_sw::cw_switches().cross_per_row && !g_cross_per_row) {
            const dim3 grid(p.H, p.B / p.kv_div, ATT_NS);
            if (p.kv_div <= 4) hipLaunchKernelGGL((attn_cross_mfma8_rows_kernel<1>), grid, dim3(CROSS_THREADS), 0, st, p, p.kv_div);
            else hipLaunchKernelGGL((attn_cross_mfma8_rows_kernel<2>), grid, dim3(CROSS_THREADS), 0, st, p, p.kv_div);
            return CW_OK;
        }
        // CW_CROSS8_NSB=2 (A/B): two splits per block -- twice the bytes in flight per CU, measured equal (the memory system is the bound)
        const int f = cw_sw::cw_switches().cross8_nsb;
        if (p.qa) {   // fused stage in front: the kernel finishes the query
            if (!p.qb || !p.qw || !p.qbias || !p.pstats || p.n_pstats < 1 || p.n_pstats > 128 || p.kv_div > 1) return CW_ERR_INVALID;
            if (ATT_NS % 2 == 0 && f == 2)
                hipLaunchKernelGGL((attn_cross_mfma8_kernel<2, 1>), dim3(p.H, p.B, ATT_NS / 2), dim3(CROSS_THREADS), 0, st, p);
            else if (p.B > 16) hipLaunchKernelGGL((attn_cross_mfma8_kernel<1, 2>), dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), 0, st, p);
            else hipLaunchKernelGGL((attn_cross_mfma8_kernel<1, 1>), dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), 0, st, p);
            return CW_OK;
        }
        // (17..64 rows as persistent blocks with the next item's bytes in flight: bit-identical, measured slower -- 50.9 against 38.5 us
        // per launch at 64 rows; four independent blocks per CU are the better prefetch.  profiles/r05_e4m3_stream_kernel_rejected.txt,
        // the kernel is in commit 6a5babd)
        if (ATT_NS % 2 == 0 && f == 2)
            hipLaunchKernelGGL((attn_cross_mfma8_kernel<2, 0>), dim3(p.H, p.B, ATT_NS / 2), dim3(CROSS_THREADS), 0, st, p);
        else hipLaunchKernelGGL((attn_cross_mfma8_kernel<1, 0>), dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), 0, st, p);
        return CW_OK;
    }
    if (p.qa) return CW_ERR_INVALID;                            // the VALU kernel takes a finished query
    // two splits per block once the grid is several times what is resident (4 blocks of 512 threads per CU); A/B: CW_CROSS8_NSB=1|2
    const int force = cw_sw::cw_switches().cross8_nsb;
    const bool pair = ATT_NS % 2 == 0 && (force ? force == 2 : (size_t)p.H * p.B * ATT_NS >= 4096);
    if (pair) hipLaunchKernelGGL(attn_cross_split_fp8_kernel<2>, dim3(p.H, p.B, ATT_NS / 2), dim3(CROSS_THREADS), 0, st, p);
    else hipLaunchKernelGGL(attn_cross_split_fp8_kernel<1>, dim3(p.H, p.B, ATT_NS), dim3(CROSS_THREADS), 0, st, p);
    return CW_OK;
}

// p[k] = e[k] * exp(m_s - M) / sum_s l_s exp(m_s - M) for the key range of split s.  grid (L, n_align, B).
__global__ void align_normalize_kernel(float* __restrict__ align, const float* __restrict__ align_ml, int n_align,
                                       int align_rows, int n_keys) {
    const int row = blockIdx.x, a = blockIdx.y, b = blockIdx.z;
    const size_t rowi = ((size_t)b * n_align + a) * align_rows + row;
    const float* ml = align_ml + rowi * ATT_NS * 2;
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < ATT_NS; ++s) M = fmaxf(M, ml[2 * s]);
    float L = 0.f, w[ATT_NS];
#pragma unroll
    for (int s = 0; s < ATT_NS; ++s) { w[s] = expf(ml[2 * s] - M); L += ml[2 * s + 1] * w[s]; }
    const float inv = 1.0f / L;
    const int per = (n_keys + ATT_NS - 1) / ATT_NS;
    float* dst = align + rowi * n_keys;
    for (int k = threadIdx.x; k < n_keys; k += blockDim.x) dst[k] = dst[k] * (w[k / per] * inv);
}

int cw_launch_align_normalize(float* align, const float* align_ml, int B, int n_align, int align_rows, int L,
                              int n_keys, hipStream_t st) {
    if (L <= 0) return CW_OK;
    hipLaunchKernelGGL(align_normalize_kernel, dim3(L, n_align, B), dim3(256), 0, st, align, align_ml, n_align,
                       align_rows, n_keys);
    return CW_OK;
}

}  // namespace CW_NS
