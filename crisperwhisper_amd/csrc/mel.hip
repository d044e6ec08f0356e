// Log-mel front end on device (TF/models/whisper/feature_extraction_whisper.py:135-168).
//
// Pass 1 (mel_kernel): reflect-padded, Hann-windowed 400-point real DFT per frame (hop 160), power,
// slaney mel filterbank, log10(clamp 1e-10), plus a per-clip running max (ordered-uint atomicMax).
// The DFT is evaluated directly in float64 with a 400-entry twiddle table in LDS and both real-DFT
// symmetries folded in (see mel_kernel); 8 frames share each twiddle fetch.  f64 keeps the result closer to
// the exact spectrum than torch.stft's own f32 FFT, so the residual vs the reference is the reference's
// own rounding (<= 6e-5 measured); the stage is ~1 GFLOP/clip and far from the critical path.
// Pass 2 (mel_finish_kernel): max(x, clipmax - 8), (x + 4) / 4, written time-major [B][3000][n_mels]
// in the encoder's activation type (conv1 consumes it as an implicit GEMM) and optionally in HF's
// [B][n_mels][3000] f32 layout for host inspection.
#include "common.h"
#include "kernels.h"
#include <mutex>

namespace CW_NS {

#define MEL_FR 16       // frames per block
#define N_FFT 400
#define HOP 160
#define N_BINS 201
#define N_FRAMES 3000
#define N_SAMPLES 480000

__device__ inline unsigned int float_to_ordered(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ordered_to_float(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void mel_kernel(MelTables tb, const float* __restrict__ pcm, int n_mels,
                                                  float* __restrict__ logspec_tm, unsigned int* __restrict__ gmax) {
    // Real 400-point DFT with both symmetries folded in (4x fewer FMAs than the direct form):
    //   n-fold:  a[n] = x[n] + x[400-n], b[n] = x[n] - x[400-n]  (n = 1..199), a[0] = x[0], a[200] = x[200]
    //            Re X[k] = sum_{n=0..200} a[n] cos(2 pi k n / 400),  Im X[k] = -sum_{n=1..199} b[n] sin(2 pi k n / 400)
    //   k-fold:  cos(2 pi (200-k) n / 400) = (-1)^n cos(2 pi k n / 400), sin(...) = -(-1)^n sin(...)
    //            so one thread accumulates the even-n and odd-n partial sums of bin k and gets bin 200-k for free.
    __shared__ double s_cos[N_FFT];
    __shared__ double s_sin[N_FFT];
    __shared__ double s_a[MEL_FR][204];      // folded frames (n = 0..200), later reused as f32 power [MEL_FR][204]
    __shared__ double s_b[MEL_FR][204];
    __shared__ float s_red[8];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * MEL_FR;
    const float* x = pcm + (size_t)b * N_SAMPLES;

    for (int i = tid; i < N_FFT; i += 256) { s_cos[i] = tb.cos_t[i]; s_sin[i] = tb.sin_t[i]; }
    auto sample = [&](int frame, int n) -> double {     // windowed, reflect-padded (center=True) sample n of a frame
        int k = frame * HOP + n - N_FFT / 2;
        if (k < 0) k = -k;
        if (k >= N_SAMPLES) k = 2 * (N_SAMPLES - 1) - k;
        return (double)x[k] * tb.window[n];
    };
    for (int i = tid; i < MEL_FR * 201; i += 256) {
        const int f = i / 201, n = i - f * 201;
        const int frame = f0 + f;
        double av = 0.0, bv = 0.0;
        if (frame < N_FRAMES) {
            const double xn = sample(frame, n);
            if (n == 0 || n == 200) { av = xn; }
            else { const double xm = sample(frame, N_FFT - n); av = xn + xm; bv = xn - xm; }
        }
        s_a[f][n] = av; s_b[f][n] = bv;
    }
    __syncthreads();

    // thread -> (bin k = tid & 127 in 0..100, frame half fh = tid >> 7): 8 frames x {Ce, Co, Se, So}
    const int k = tid & 127, fh = tid >> 7;
    double ce[8], co[8], se[8], so[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) { ce[f] = 0.0; co[f] = 0.0; se[f] = 0.0; so[f] = 0.0; }
    if (k <= 100) {
        int idx = 0;                                               // (k * n) mod 400
        for (int n = 0; n <= 200; n += 2) {
            {   // even n
                const double c = s_cos[idx], sn = s_sin[idx];
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    ce[f] = fma(s_a[fh * 8 + f][n], c, ce[f]);
                    se[f] = fma(s_b[fh * 8 + f][n], sn, se[f]);
                }
                idx += k; if (idx >= N_FFT) idx -= N_FFT;
            }
            if (n + 1 <= 199) {   // odd n
                const double c = s_cos[idx], sn = s_sin[idx];
#pragma unroll
                for (int f = 0; f < 8; ++f) {
                    co[f] = fma(s_a[fh * 8 + f][n + 1], c, co[f]);
                    so[f] = fma(s_b[fh * 8 + f][n + 1], sn, so[f]);
                }
                idx += k; if (idx >= N_FFT) idx -= N_FFT;
            } else { idx += k; if (idx >= N_FFT) idx -= N_FFT; }
        }
    }
    __syncthreads();
    float* s_pw = (float*)&s_a[0][0];                              // [MEL_FR][204] floats fit in the first 13 KB
    if (k <= 100) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const double re1 = ce[f] + co[f], im1 = se[f] + so[f];
            const double re2 = ce[f] - co[f], im2 = so[f] - se[f];
            s_pw[(fh * 8 + f) * 204 + k] = (float)(re1 * re1 + im1 * im1);
            s_pw[(fh * 8 + f) * 204 + 200 - k] = (float)(re2 * re2 + im2 * im2);   // k = 100 writes the same bin twice
        }
    }
    __syncthreads();

    float lmax = -INFINITY;
    if (tid < n_mels) {
        // one filter-bank load per bin, reused by the 16 frames of the block (the triangles are sparse: zero
        // weights are skipped, which keeps the summation order of the non-zero terms)
        double acc[MEL_FR];
#pragma unroll
        for (int f = 0; f < MEL_FR; ++f) acc[f] = 0.0;
        for (int kk = 0; kk < N_BINS; ++kk) {
            const float w = tb.filters[kk * n_mels + tid];
            if (w != 0.f) {
                const double wd = (double)w;
#pragma unroll
                for (int f = 0; f < MEL_FR; ++f) acc[f] = fma(wd, (double)s_pw[f * 204 + kk], acc[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < MEL_FR; ++f) {
            const int frame = f0 + f;
            if (frame < N_FRAMES) {
                const float lv = log10f(fmaxf((float)acc[f], 1e-10f));
                logspec_tm[((size_t)b * N_FRAMES + frame) * n_mels + tid] = lv;
                lmax = fmaxf(lmax, lv);
            }
        }
    }
    lmax = block_max(lmax, s_red);
    if (tid == 0) atomicMax(gmax + b, float_to_ordered(lmax));
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the same folded DFT as a matrix product on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
// With the n-fold (a[n] = x[n] + x[400-n], b[n] = x[n] - x[400-n]) and the k-fold (bin 200-k from the even-n / odd-n partial
// sums of bin k) a block of 32 frames is four products  [32 frames x ~101 samples] x [~101 x 101 bins]:
//     Ce = A_even C_even,  Co = A_odd C_odd,  Se = B_even S_even,  So = B_odd S_odd        (one per wave)
//     Re X[k] = Ce + Co,  Im X[k] = -(Se + So),  Re X[200-k] = Ce - Co,  Im X[200-k] = So - Se
// 1.95 GFLOP per 8 clips in f64 -- the arithmetic of the VALU kernel above (which reaches 6 % of the f64 rate: every FMA of
// its inner loop waits for two LDS operands).  A wave owns two 16-bin tiles of ALL FOUR products, so Ce, Co, Se, So of a
// (frame, bin) end up in one lane; the basis fragments are gathered from the 400-entry cos / sin tables in LDS by phase
// (k n mod 400) -- a fragment-major basis in memory (373 KB per block from L2, 280 MB per 8 clips) was the first version's
// limiter.  f64 keeps the spectrum exact to 1e-13; an f32 product would sit 7-9e-5
// from the reference on pure tones (bins 80 dB below the peak see the f32 rounding floor), too close to the 1e-4 bar.
// LDS: frames a/b even/odd [4][32][105] f64 | raw samples 5360 f32 (later the power spectrum [32][204] f32) | Hann window,
// cos, sin tables [400] f64 each.
// ---------------------------------------------------------------------------------------------------
#define MM_FR 32                       // frames per block
#define MM_KP 105                      // row stride of the frame arrays (doubles)
#define MM_KS 26                       // k-steps of 4 (even n: 101 -> 104; odd n: 100)
typedef __attribute__((ext_vector_type(4))) double f64x4_t;

__global__ __launch_bounds__(256) void mel_mfma_kernel(MelTables tb, const float* __restrict__ pcm, int n_mels,
                                                       float* __restrict__ logspec_tm, unsigned int* __restrict__ gmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mm_smem[];
    double* s_ab = (double*)mm_smem;                                    // [4][32][105]
    float* s_raw = (float*)(mm_smem + (size_t)4 * MM_FR * MM_KP * 8);   // 5360 floats -> power [32][204]
    double* s_win = (double*)(mm_smem + (size_t)4 * MM_FR * MM_KP * 8 + (size_t)MM_FR * 204 * 4);
    double* s_cos = s_win + N_FFT;
    double* s_sin = s_cos + N_FFT;
    __shared__ float s_red[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CW_SK_DEBUG
    const int dbg = tb.dbg;
#else
    const int dbg = 0;
#endif
    const int b = blockIdx.y, f0 = blockIdx.x * MM_FR;
    const float* x = pcm + (size_t)b * N_SAMPLES;

    // ---- raw samples of the block's frames (reflect-padded at the clip edges), window
    const int s0 = f0 * HOP - N_FFT / 2, n_raw = (MM_FR - 1) * HOP + N_FFT;
    for (int i = tid; i < n_raw; i += 256) {
        int k = s0 + i;
        if (k < 0) k = -k;
        if (k >= N_SAMPLES) k = 2 * (N_SAMPLES - 1) - k;
        s_raw[i] = (k >= 0 && k < N_SAMPLES) ? x[k] : 0.f;
    }
    for (int i = tid; i < N_FFT; i += 256) { s_win[i] = tb.window[i]; s_cos[i] = tb.cos_t[i]; s_sin[i] = tb.sin_t[i]; }
    __syncthreads();
    // ---- folded frames: q = 0 a even n, 1 a odd n, 2 b even n, 3 b odd n; column j <-> n = 2 j (+ 1)
    for (int j = tid & 7, f = tid >> 3; j < ((dbg & 2) ? 8 : MM_KP); j += 8) {            // 8 threads per frame
        const bool live = f0 + f < N_FRAMES;
        double ae = 0.0, ao = 0.0, be = 0.0, bo = 0.0;
        if (live && j <= 100) {
            const int n = 2 * j;
            const double xn = (double)s_raw[f * HOP + n] * s_win[n];
            if (n == 0 || n == 200) ae = xn;
            else { const double xm = (double)s_raw[f * HOP + N_FFT - n] * s_win[N_FFT - n]; ae = xn + xm; be = xn - xm; }
        }
        if (live && j <= 99) {
            const int n = 2 * j + 1;
            const double xn = (double)s_raw[f * HOP + n] * s_win[n], xm = (double)s_raw[f * HOP + N_FFT - n] * s_win[N_FFT - n];
            ao = xn + xm; bo = xn - xm;
        }
        s_ab[(0 * MM_FR + f) * MM_KP + j] = ae; s_ab[(1 * MM_FR + f) * MM_KP + j] = ao;
        s_ab[(2 * MM_FR + f) * MM_KP + j] = be; s_ab[(3 * MM_FR + f) * MM_KP + j] = bo;
    }
    __syncthreads();

    // ---- wave w: bin tiles 2 w, 2 w + 1 (wave 3: tile 6 only) of all four products, both frame tiles.  A fragment: lane ->
    // frame lane % 16, sample column 4 ks + lane / 16.  The basis never leaves the CU: cos / sin of the 400 phases sit in LDS and
    // a lane gathers entry (bin * n) mod 400 of its (sample, bin), advancing the phase by 8 bin per k-step (n grows by 8).
    const int ntl = wave < 3 ? 2 : 1;                                    // wave-uniform
    f64x4_t acc[4][2][2];                                                // [product][frame tile][bin tile]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[q][mt][t] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
    int ph_e[2], ph_o[2], dph[2];                                        // phases (k n mod 400) of the even / odd sample, step 8 k mod 400
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int bin = (wave * 2 + t) * 16 + (lane & 15), j = lane >> 4;
        ph_e[t] = (bin * (2 * j)) % N_FFT; ph_o[t] = (bin * (2 * j + 1)) % N_FFT; dph[t] = (bin * 8) % N_FFT;
    }
    const double* arow = s_ab + (size_t)(lane & 15) * MM_KP + (lane >> 4);
    // operands of k-step ks + 1 are read from LDS under the MFMAs of k-step ks
    double an[4][2], bn[2][4];
    auto fetch = [&](int ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) an[q][mt] = arow[((size_t)q * MM_FR + mt * 16) * MM_KP + ks * 4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bn[t][0] = s_cos[ph_e[t]]; bn[t][1] = s_cos[ph_o[t]]; bn[t][2] = s_sin[ph_e[t]]; bn[t][3] = s_sin[ph_o[t]];
            ph_e[t] += dph[t]; ph_e[t] -= ph_e[t] >= N_FFT ? N_FFT : 0;
            ph_o[t] += dph[t]; ph_o[t] -= ph_o[t] >= N_FFT ? N_FFT : 0;
        }
    };
    fetch(0);
    const int n_ks = (dbg & 1) ? 1 : MM_KS;
    for (int ks = 0; ks < n_ks; ++ks) {
        double a[4][2], bc[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q][0] = an[q][0]; a[q][1] = an[q][1]; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[t][q] = bn[t][q];
        if (ks + 1 < n_ks) fetch(ks + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < ntl) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[q][mt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][mt], bc[t][q], acc[q][mt][t], 0, 0, 0);
            }
        }
    }
    // D[frame = mt*16 + lane / 16 + 4 i][bin = nt*16 + lane % 16]   (f64 layout; tools/probe/mfma_f64_layout.hip): the four
    // products of a (frame, bin) sit in the same lane, so the power spectrum needs no exchange
    float* s_pw = s_raw;                                                 // [32][204]; the raw samples were consumed by the fold
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = (wave * 2 + t) * 16 + (lane & 15);
        if (t < ntl && k <= 100) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = mt * 16 + (lane >> 4) + 4 * i;
                    const double ce = acc[0][mt][t][i], co = acc[1][mt][t][i], se = acc[2][mt][t][i], so = acc[3][mt][t][i];
                    const double re1 = ce + co, im1 = se + so, re2 = ce - co, im2 = so - se;
                    if (k < 100) s_pw[f * 204 + 200 - k] = (float)(re2 * re2 + im2 * im2);
                    s_pw[f * 204 + k] = k == 100 ? (float)(re2 * re2 + im2 * im2) : (float)(re1 * re1 + im1 * im1);   // bin 100: the value the VALU kernel's second write leaves
                }
        }
    }
    __syncthreads();

    // ---- slaney mel bank (sparse triangles: zero weights are skipped, the non-zero terms keep their order), log10, clip maximum
    float lmax = -INFINITY;
    const int m = tid & 127, fh = tid >> 7;                              // filter, frame half
    if (m < n_mels) {
        double macc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) macc[f] = 0.0;
        const int lo = tb.fb_lo[m], hi = tb.fb_hi[m];                    // bins with a non-zero weight: lo .. hi (<= 14 wide up to 80 mels)
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = tb.filters[min(lo + u, N_BINS - 1) * n_mels + m];   // all requests out at once
        for (int u0 = 0; lo + u0 <= hi && !(dbg & 4); u0 += 16) {
            if (u0 > 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = tb.filters[min(lo + u0 + u, N_BINS - 1) * n_mels + m];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int kk = lo + u0 + u;
                const float w = wv[u];
                if (kk <= hi && w != 0.f) {
                    const double wd = (double)w;
#pragma unroll
                    for (int f = 0; f < 16; ++f) macc[f] = fma(wd, (double)s_pw[(fh * 16 + f) * 204 + kk], macc[f]);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int frame = f0 + fh * 16 + f;
            if (frame < N_FRAMES) {
                const float lv = log10f(fmaxf((float)macc[f], 1e-10f));
                logspec_tm[((size_t)b * N_FRAMES + frame) * n_mels + m] = lv;
                lmax = fmaxf(lmax, lv);
            }
        }
    }
    lmax = block_max(lmax, s_red);
    if (tid == 0) atomicMax(gmax + b, float_to_ordered(lmax));
}

template <typename T>
__global__ void mel_finish_kernel(const float* __restrict__ logspec_tm, const unsigned int* __restrict__ gmax,
                                  int n_mels, T* __restrict__ feats_tm, float* __restrict__ feats_hf) {
    const int b = blockIdx.y;
    const float floor_v = ordered_to_float(gmax[b]) - 8.0f;
    const size_t per = (size_t)N_FRAMES * n_mels;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        float v = logspec_tm[(size_t)b * per + i];
        v = (fmaxf(v, floor_v) + 4.0f) / 4.0f;
        if (feats_tm) Act<T>::st(feats_tm + (size_t)b * per + i, v);
        if (feats_hf) {
            int frame = (int)(i / n_mels), m = (int)(i - (size_t)frame * n_mels);
            feats_hf[(size_t)b * per + (size_t)m * N_FRAMES + frame] = v;
        }
    }
}

int cw_launch_mel(const MelTables& t, const float* pcm, int B, int n_mels, float* logspec_tm, unsigned int* gmax,
                  hipStream_t st) {
    if (n_mels > 256 || B <= 0) return CW_ERR_INVALID;
    (void)hipMemsetAsync(gmax, 0, sizeof(unsigned int) * B, st);   // ordered encoding: 0 is below every float
    const bool valu = cw_sw::cw_switches().mel_valu;   // A/B: round-1 VALU kernel
#ifdef CW_SK_DEBUG
    MelTables t2 = t; t2.dbg = cw_sw::cw_switches().mel_dbg;
#else
    const MelTables& t2 = t;
#endif
    if (t.fb_lo && n_mels <= 128 && !valu) {
        const size_t lds = (size_t)4 * MM_FR * MM_KP * 8 + (size_t)MM_FR * 204 * 4 + (size_t)3 * N_FFT * 8;
        static std::once_flag attr;
        std::call_once(attr, [lds] { (void)hipFuncSetAttribute((const void*)mel_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
        hipLaunchKernelGGL(mel_mfma_kernel, dim3((N_FRAMES + MM_FR - 1) / MM_FR, B), dim3(256), lds, st, t2, pcm, n_mels, logspec_tm, gmax);
        return CW_OK;
    }
    hipLaunchKernelGGL(mel_kernel, dim3((N_FRAMES + MEL_FR - 1) / MEL_FR, B), dim3(256), 0, st, t, pcm, n_mels,
                       logspec_tm, gmax);
    return CW_OK;
}

int cw_launch_mel_finish(const float* logspec_tm, const unsigned int* gmax, int B, int n_mels, void* feats_tm,
                         int feats_bf16, float* feats_hf, hipStream_t st) {
    dim3 grid(256, B);
    if (feats_bf16)
        hipLaunchKernelGGL((mel_finish_kernel<bf16_t>), grid, dim3(256), 0, st, logspec_tm, gmax, n_mels,
                           (bf16_t*)feats_tm, feats_hf);
    else
        hipLaunchKernelGGL((mel_finish_kernel<float>), grid, dim3(256), 0, st, logspec_tm, gmax, n_mels,
                           (float*)feats_tm, feats_hf);
    return CW_OK;
}

}  // namespace CW_NS
