// Audio ingest on device: the step in front of the feature extractor.
//   pcm_to_mono_kernel   interleaved RIFF/WAVE sample frames (u8 / s16 / s24 / s32 / f32 / f64, little endian) ->
//                        mono f32 (integer formats scaled to [-1, 1), channels averaged) -- what the ffmpeg call
//                        `-ac 1 -f f32le` of TF/pipelines/audio_utils.py:9-45 hands the pipeline;
//   moments_kernel       sum / sum of squares in f64 (REF/app.py:85-93 normalises (y - mean) / std / 8 before resampling);
//   resample_kernel      torchaudio.functional.resample with its defaults (sinc_interp_hann, lowpass_filter_width 6,
//                        rolloff 0.99), the resampler behind TF/pipelines/automatic_speech_recognition.py:398-412
//                        and REF/app.py:94-95: zero-padded polyphase FIR, out[i*new + p] = sum_j K[p][j] x[i*orig + j - width].
//                        The tap table is built on the host in f64 (cw_resample_taps) and stored transposed [j][p] so
//                        that consecutive output samples read consecutive taps.
// All three are single-pass streaming kernels (HBM-bound; the FIR re-reads its <= 2*width + orig input window from L1/L2).
#include "common.h"
#include "kernels.h"
#include "../../include/crisperwhisper.h"

__device__ inline float pcm_sample(const unsigned char* p, int fmt) {
    switch (fmt) {
        case CW_PCM_U8: return ((float)p[0] - 128.0f) / 128.0f;
        case CW_PCM_S16: return (float)(short)(p[0] | (p[1] << 8)) / 32768.0f;
        case CW_PCM_S24: {
            int v = p[0] | (p[1] << 8) | (p[2] << 16);
            v = (v << 8) >> 8;                                  // sign extend
            return (float)v / 8388608.0f;
        }
        case CW_PCM_S32: {
            int v = (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
            return (float)v / 2147483648.0f;
        }
        case CW_PCM_F32: {
            unsigned u = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
            return __uint_as_float(u);
        }
        default: {   // CW_PCM_F64
            unsigned long long u = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) u |= (unsigned long long)p[i] << (8 * i);
            return (float)__longlong_as_double((long long)u);
        }
    }
}

__global__ void pcm_to_mono_kernel(const unsigned char* __restrict__ raw, int fmt, int bytes, int channels,
                                   long long n_frames, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames;
         i += (long long)gridDim.x * blockDim.x) {
        const unsigned char* p = raw + (size_t)i * channels * bytes;
        if (channels == 1) {
            out[i] = pcm_sample(p, fmt);
        } else {                                                // mean over channels, accumulated like numpy (f32, in order)
            float s = 0.f;
            for (int ch = 0; ch < channels; ++ch) s += pcm_sample(p + ch * bytes, fmt);
            out[i] = s / (float)channels;
        }
    }
}

__global__ void moments_kernel(const float* __restrict__ x, long long n, double* __restrict__ acc /* [2] */) {
    __shared__ double s_a[256], s_b[256];
    double a = 0.0, b = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        a += v; b += v * v;
    }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s_a[threadIdx.x] += s_a[threadIdx.x + s]; s_b[threadIdx.x] += s_b[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(acc, s_a[0]); atomicAdd(acc + 1, s_b[0]); }
}

// y = (x - mean) / std / 8 with the population std (np.std), from the two moments
__global__ void normalise_kernel(float* __restrict__ x, long long n, const double* __restrict__ acc) {
    const double mean = acc[0] / (double)n;
    double var = acc[1] / (double)n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float m = (float)mean, inv = (float)(1.0 / sqrt(var));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] = (x[i] - m) * inv / 8.0f;
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long n_in,
                                                       const float* __restrict__ taps_t /* [n_taps][new] */,
                                                       int orig, int nw, int width, int n_taps, long long n_out,
                                                       float* __restrict__ out) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_out) return;
    const long long i = n / nw;
    const int p = (int)(n - i * nw);
    const long long base = i * orig - width;                    // first input sample under the window
    float acc = 0.f;
    int j0 = 0, j1 = n_taps;
    if (base < 0) j0 = (int)(-base);
    if (base + n_taps > n_in) j1 = (int)(n_in - base);
    for (int j = j0; j < j1; ++j) acc = fmaf(taps_t[(size_t)j * nw + p], x[base + j], acc);
    out[n] = acc;
}

int cw_launch_pcm_to_mono(const void* raw, int fmt, int channels, long long n_frames, float* out, hipStream_t st) {
    static const int bytes_of[] = {1, 2, 3, 4, 4, 8};
    if (fmt < 0 || fmt > CW_PCM_F64 || channels < 1 || n_frames < 1) return CW_ERR_INVALID;
    long long blocks = (n_frames + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pcm_to_mono_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned char*)raw, fmt,
                       bytes_of[fmt], channels, n_frames, out);
    return CW_OK;
}

int cw_launch_normalise(float* x, long long n, double* acc2, hipStream_t st) {
    if (n < 1) return CW_ERR_INVALID;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipMemsetAsync(acc2, 0, 16, st);
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, acc2);
    hipLaunchKernelGGL(normalise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, acc2);
    return CW_OK;
}

int cw_launch_resample(const float* x, long long n_in, const float* taps_t, int orig, int nw, int width,
                       long long n_out, float* out, hipStream_t st) {
    if (n_in < 1 || n_out < 1) return CW_ERR_INVALID;
    const long long blocks = (n_out + 255) / 256;
    if (blocks > 0x7fffffffLL) return CW_ERR_INVALID;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n_in, taps_t, orig, nw, width,
                       2 * width + orig, n_out, out);
    return CW_OK;
}
