// Log-mel front end on device (TF/models/whisper/feature_extraction_whisper.py:135-168).
//
// Pass 1 (mel_kernel): reflect-padded, Hann-windowed 400-point real DFT per frame (hop 160), power,
// slaney mel filterbank, log10(clamp 1e-10), plus a per-clip running max (ordered-uint atomicMax).
// The DFT is evaluated directly in float64 with a 400-entry twiddle table in LDS: 16 frames share each
// twiddle fetch, so the inner loop is 32 f64 FMAs per 2 table reads.  f64 keeps the result closer to
// the exact spectrum than torch.stft's own f32 FFT, so the residual vs the reference is the reference's
// own rounding (<= 6e-5 measured); the stage is ~1 GFLOP/clip and far from the critical path.
// Pass 2 (mel_finish_kernel): max(x, clipmax - 8), (x + 4) / 4, written time-major [B][3000][n_mels]
// in the encoder's activation type (conv1 consumes it as an implicit GEMM) and optionally in HF's
// [B][n_mels][3000] f32 layout for host inspection.
#include "common.h"
#include "kernels.h"

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
    __shared__ double s_cos[N_FFT];
    __shared__ double s_sin[N_FFT];
    __shared__ double s_xw[MEL_FR][N_FFT];   // 51.2 KB windowed frames; reused as f32 power [MEL_FR][N_BINS]
    __shared__ float s_red[8];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * MEL_FR;
    const float* x = pcm + (size_t)b * N_SAMPLES;

    for (int i = tid; i < N_FFT; i += 256) { s_cos[i] = tb.cos_t[i]; s_sin[i] = tb.sin_t[i]; }
    for (int i = tid; i < MEL_FR * N_FFT; i += 256) {
        int f = i / N_FFT, n = i - f * N_FFT;
        int frame = f0 + f;
        double v = 0.0;
        if (frame < N_FRAMES) {
            int k = frame * HOP + n - N_FFT / 2;                 // index into the unpadded signal
            if (k < 0) k = -k;                                     // reflect (center=True)
            if (k >= N_SAMPLES) k = 2 * (N_SAMPLES - 1) - k;
            v = (double)x[k] * tb.window[n];
        }
        s_xw[f][n] = v;
    }
    __syncthreads();

    double re[MEL_FR], im[MEL_FR];
#pragma unroll
    for (int f = 0; f < MEL_FR; ++f) { re[f] = 0.0; im[f] = 0.0; }
    if (tid < N_BINS) {
        int idx = 0;                                               // (tid * n) mod 400
        for (int n = 0; n < N_FFT; ++n) {
            const double c = s_cos[idx], s = s_sin[idx];
#pragma unroll
            for (int f = 0; f < MEL_FR; ++f) {
                const double v = s_xw[f][n];
                re[f] = fma(v, c, re[f]);
                im[f] = fma(v, s, im[f]);
            }
            idx += tid;
            if (idx >= N_FFT) idx -= N_FFT;
        }
    }
    __syncthreads();
    float* s_pw = (float*)&s_xw[0][0];                             // [MEL_FR][N_BINS + 3]
    if (tid < N_BINS) {
#pragma unroll
        for (int f = 0; f < MEL_FR; ++f) s_pw[f * 204 + tid] = (float)(re[f] * re[f] + im[f] * im[f]);
    }
    __syncthreads();

    float lmax = -INFINITY;
    if (tid < n_mels) {
        for (int f = 0; f < MEL_FR; ++f) {
            int frame = f0 + f;
            if (frame >= N_FRAMES) break;
            double acc = 0.0;
            for (int k = 0; k < N_BINS; ++k) acc = fma((double)tb.filters[k * n_mels + tid], (double)s_pw[f * 204 + k], acc);
            float mel = (float)acc;
            float lv = log10f(fmaxf(mel, 1e-10f));
            logspec_tm[((size_t)b * N_FRAMES + frame) * n_mels + tid] = lv;
            lmax = fmaxf(lmax, lv);
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
    hipMemsetAsync(gmax, 0, sizeof(unsigned int) * B, st);   // ordered encoding: 0 is below every float
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
