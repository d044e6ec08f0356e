// Token-timestamp stage on device (TF/models/whisper/generation_whisper.py:241-381):
//   align_stats_kernel   population mean/std over the token axis per (item, head, frame)   (:343-345)
//   align_filter_kernel  z-score -> width-w running median along frames (reflect edges, exact selection
//                        by sorting network, :43-61) -> mean over heads (:349), fused, one LDS-staged row
//                        segment per head
//   dtw_kernel           anti-diagonal wavefront DTW over the (N+1) x (M+1) cost lattice with the
//                        reference's asymmetric tie rule (:80-85) and f32 cost accumulation (:70,87),
//                        three rolling diagonals in LDS, byte trace in global (L2), on-device backtrace
//                        (:89-115) that also emits, per token row, the first frame of its run (= jump
//                        time / 0.02 s, :368-369)
//   pauses_kernel        REF/utils.py:8-26: per-boundary pause redistribution (boundaries independent)
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

// w: [B][Ha][rows_cap][S]; token rows row0 .. row0+N-1; frames j < n_cols[b].
__global__ void align_stats_kernel(const float* __restrict__ w, int Ha, int rows_cap, int S, int row0, int N,
                                   const int* __restrict__ n_cols, float* __restrict__ mean, float* __restrict__ stdv) {
    const int a = blockIdx.y, b = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cols[b]) return;
    const float* base = w + (((size_t)b * Ha + a) * rows_cap + row0) * S + j;
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += (double)base[(size_t)i * S];
    const double mu = s / N;
    double q = 0.0;
    for (int i = 0; i < N; ++i) { double d = (double)base[(size_t)i * S] - mu; q += d * d; }
    const size_t o = ((size_t)b * Ha + a) * S + j;
    mean[o] = (float)mu;
    stdv[o] = (float)sqrt(q / N);
}

__device__ inline void cswap(float& a, float& b) { float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }

// exact median of up to 9 values (odd width) by insertion-free sorting network on registers
template <int W> __device__ inline float median_w(float* v) {
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j + 1 < W - i; ++j) cswap(v[j], v[j + 1]);
    return v[W / 2];
}

#define FT 256  // frames per block
// grid: (ceil(maxcols/FT), N, B); out mat[b][i][j] (row stride S)
template <int W>
__global__ __launch_bounds__(FT) void align_filter_kernel(const float* __restrict__ w, int Ha, int rows_cap, int S,
                                                          int row0, int N, const int* __restrict__ n_cols,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, float* __restrict__ mat) {
    __shared__ float z[FT + 2 * (W / 2)];
    const int b = blockIdx.z, i = blockIdx.y;
    const int M = n_cols[b];
    const int j0 = blockIdx.x * FT;
    if (j0 >= M) return;
    const int pad = W / 2;
    const int j = j0 + threadIdx.x;
    float acc = 0.f;
    for (int a = 0; a < Ha; ++a) {
        const float* row = w + (((size_t)b * Ha + a) * rows_cap + row0 + i) * S;
        const float* mu = mean + ((size_t)b * Ha + a) * S;
        const float* sd = stdv + ((size_t)b * Ha + a) * S;
        __syncthreads();
        for (int t = threadIdx.x; t < FT + 2 * pad; t += FT) {
            int jj = j0 - pad + t;
            if (M > pad) {                     // reflect padding (:57)
                if (jj < 0) jj = -jj;
                if (jj >= M) jj = 2 * (M - 1) - jj;
            }
            float zz = 0.f;
            if (jj >= 0 && jj < M) zz = (row[jj] - mu[jj]) / sd[jj];
            z[t] = zz;
        }
        __syncthreads();
        if (j < M) {
            float v;
            if (M <= pad) {
                v = z[threadIdx.x + pad];      // early return of _median_filter (:53-54)
            } else {
                float win[W];
#pragma unroll
                for (int t = 0; t < W; ++t) win[t] = z[threadIdx.x + t];
                v = median_w<W>(win);
            }
            acc += v;
        }
    }
    if (j < M) mat[((size_t)b * N + i) * S + j] = acc / (float)Ha;
}

// One block per item.  Diagonal d = i + j (1 <= i <= N, 1 <= j <= M).  cost index by i.
#define DTW_THREADS 512
__global__ __launch_bounds__(DTW_THREADS) void dtw_kernel(const float* __restrict__ mat, int N, int S,
                                                          const int* __restrict__ n_cols,
                                                          unsigned char* __restrict__ trace,
                                                          int* __restrict__ first_col, int* __restrict__ path_text,
                                                          int* __restrict__ path_time, int* __restrict__ path_len) {
    __shared__ float diag[3][DTW_THREADS + 1];
    const int b = blockIdx.x;
    const int M = n_cols[b];
    const float* x = mat + (size_t)b * N * S;              // x[i-1][j-1]; the DP runs on -x (:367)
    unsigned char* tr = trace + (size_t)b * N * S;           // tr[(i-1)*S + (j-1)]
    const int i = threadIdx.x + 1;                           // this thread's row, 1..N
    const bool active = i <= N;

    // d = 0: only cost[0][0] = 0; d = 1: cost[0][1] = cost[1][0] = inf
    for (int k = threadIdx.x; k <= DTW_THREADS; k += DTW_THREADS) {
        diag[0][k] = INFINITY; diag[1][k] = INFINITY; diag[2][k] = INFINITY;
    }
    if (threadIdx.x == 0) { diag[0][DTW_THREADS] = INFINITY; diag[1][DTW_THREADS] = INFINITY; diag[2][DTW_THREADS] = INFINITY; }
    __syncthreads();
    if (threadIdx.x == 0) diag[0][0] = 0.f;                  // diag for d-2 at step d=2 is "d = 0"
    __syncthreads();

    int p2 = 0, p1 = 1, cur = 2;                             // buffers for d-2, d-1, d
    float xnext = 0.f;
    {   // prefetch for d = 2
        int j = 2 - i;
        if (active && j >= 1 && j <= M) xnext = -x[(size_t)(i - 1) * S + (j - 1)];
    }
    for (int d = 2; d <= N + M; ++d) {
        const int j = d - i;
        const bool on = active && j >= 1 && j <= M;
        const float xv = xnext;
        {   // prefetch next diagonal's matrix value (independent of the DP chain)
            int jn = d + 1 - i;
            xnext = (active && jn >= 1 && jn <= M) ? -x[(size_t)(i - 1) * S + (jn - 1)] : 0.f;
        }
        float c = INFINITY;
        if (on) {
            const float c0 = diag[p2][i - 1];               // cost[i-1][j-1]
            const float c1 = diag[p1][i - 1];               // cost[i-1][j]
            const float c2 = diag[p1][i];                   // cost[i][j-1]
            float cm; unsigned char t;
            if (c0 < c1 && c0 < c2) { cm = c0; t = 0; }
            else if (c1 < c0 && c1 < c2) { cm = c1; t = 1; }
            else { cm = c2; t = 2; }
            c = xv + cm;
            tr[(size_t)(i - 1) * S + (j - 1)] = t;
        }
        if (active) diag[cur][i] = c;                        // cells off the lattice stay +inf
        if (threadIdx.x == 0) diag[cur][0] = INFINITY;       // cost[0][d] = inf for d >= 1
        __syncthreads();
        int tmp = p2; p2 = p1; p1 = cur; cur = tmp;
    }

    // backtrace (:89-115), single lane; trace[0][:] = 2 and trace[:][0] = 1 are implicit
    if (threadIdx.x == 0) {
        __threadfence_block();
        int ii = N, jj = M, n = 0;
        int* pt = path_text + (size_t)b * (N + S + 2);
        int* pj = path_time + (size_t)b * (N + S + 2);
        while (ii > 0 || jj > 0) {
            pt[n] = ii - 1; pj[n] = jj - 1; ++n;
            if (ii > 0 && jj > 0) first_col[(size_t)b * N + (ii - 1)] = jj - 1;
            unsigned char t = (ii == 0) ? 2 : (jj == 0) ? 1 : tr[(size_t)(ii - 1) * S + (jj - 1)];
            if (t == 0) { --ii; --jj; } else if (t == 1) { --ii; } else { --jj; }
        }
        path_len[b] = n;   // stored in reverse order (end -> start); the host flips it
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-local DTW (round 2).  The block kernel above pays one __syncthreads per anti-diagonal (0.43-0.59 us each, 1627 of
// them at N = 128) and walks the byte trace back through global memory one dependent load per step.  Here one WAVE owns a
// sequence: lane l holds rows l*R+1 .. l*R+R, the three cells a cell depends on are either in the lane's own registers
// or the previous lane's last row (one wave_shr:1 DPP move per diagonal), so a diagonal is ~R x 12 dependent-free
// instructions and no barrier.  The cost matrix is first skewed into diagonal-major order (dtw_skew_kernel, parallel and
// bandwidth-bound), so the wave's operands of diagonal d are one contiguous, coalesced vector that is requested P
// diagonals (several hundred cycles) before it is needed; the 2-bit trace codes are packed 16 per dword per row (one LDS
// store per row per 16 diagonals) and the backtrace re-uses a loaded word while it walks along a row.
// Same arithmetic, tie rule and boundary conventions as dtw_kernel / generation_whisper.py:64-115 (bit-exact paths).
// ---------------------------------------------------------------------------------------------------
// xd[b][d][c] = -x[b][c][d - c - 2]  for rows c = i - 1 in 0..NP-1 and diagonals d = i + j in 2..N+M (0 off the lattice)
__global__ void dtw_skew_kernel(const float* __restrict__ mat, int N, int S, const int* __restrict__ n_cols, int NP,
                                int D_cap, float* __restrict__ xd) {
    const int b = blockIdx.z, d = blockIdx.x + 2, M = n_cols[b];
    if (d > N + M) return;
    const float* x = mat + (size_t)b * N * S;
    float* dst = xd + ((size_t)b * D_cap + d) * NP;
    for (int c = threadIdx.x; c < NP; c += blockDim.x) {
        const int j = d - (c + 1);
        dst[c] = (c < N && j >= 1 && j <= M) ? -x[(size_t)c * S + (j - 1)] : 0.f;
    }
}

template <int R> struct DtwVec;
template <> struct DtwVec<2> { float v[2]; __device__ inline void ld(const float* p) { const float2 t = *(const float2*)p; v[0] = t.x; v[1] = t.y; } };
template <> struct DtwVec<4> { float v[4]; __device__ inline void ld(const float* p) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; } };
template <> struct DtwVec<8> { float v[8]; __device__ inline void ld(const float* p) { const float4 t = *(const float4*)p, u = *(const float4*)(p + 4); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w; } };

template <int R, int P>
__global__ __launch_bounds__(64) void dtw_wave_kernel(const float* __restrict__ xd, int N, int S, int D_cap,
                                                      const int* __restrict__ n_cols, unsigned int* __restrict__ trace_g,
                                                      size_t trace_stride_words, int use_lds, int* __restrict__ first_col,
                                                      int* __restrict__ path_text, int* __restrict__ path_time,
                                                      int* __restrict__ path_len) {
    extern __shared__ unsigned int tr_lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int M = n_cols[b];
    constexpr int NP = R * 64;
    const float* xb = xd + (size_t)b * D_cap * NP + lane * R;
    const int SW = (S + 15) >> 4;                              // trace words per row
    unsigned int* tr = use_lds ? tr_lds : trace_g + (size_t)b * trace_stride_words;
    float v1[R], v2[R];                                        // this lane's rows on diagonals d-1 and d-2
    unsigned int acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { v1[r] = INFINITY; v2[r] = INFINITY; acc[r] = 0u; }
    float up1 = INFINITY, up2 = INFINITY;                      // previous lane's last row on diagonals d-1 / d-2
    const int d_end = N + M;                                   // last diagonal
    DtwVec<R> ring[P], nxt[P];
#pragma unroll
    for (int u = 0; u < P; ++u) ring[u].ld(xb + (size_t)min(2 + u, D_cap - 1) * NP);
    for (int d0 = 2; d0 <= d_end; d0 += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) nxt[u].ld(xb + (size_t)min(d0 + P + u, D_cap - 1) * NP);     // P diagonals ahead
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int d = d0 + u;
            if (d <= d_end) {
                float nv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = lane * R + r + 1, j = d - i;
                    const bool on = i <= N && j >= 1 && j <= M;
                    float c0 = (r == 0) ? up2 : v2[r > 0 ? r - 1 : 0];
                    float c1 = (r == 0) ? up1 : v1[r > 0 ? r - 1 : 0];
                    if (i == 1) { c1 = INFINITY; c0 = (j == 1) ? 0.f : INFINITY; }     // row 0 of the DP: cost[0][0] = 0, else inf
                    const float c2 = v1[r];
                    float cm; unsigned int t;
                    if (c0 < c1 && c0 < c2) { cm = c0; t = 0u; }
                    else if (c1 < c0 && c1 < c2) { cm = c1; t = 1u; }
                    else { cm = c2; t = 2u; }
                    nv[r] = on ? ring[u].v[r] + cm : INFINITY;
                    if (on) {
                        const int q = j - 1;
                        acc[r] |= t << (2 * (q & 15));
                        if ((q & 15) == 15 || j == M) { tr[(size_t)(i - 1) * SW + (q >> 4)] = acc[r]; acc[r] = 0u; }
                    }
                }
                const float last_new = nv[R - 1];
#pragma unroll
                for (int r = 0; r < R; ++r) { v2[r] = v1[r]; v1[r] = nv[r]; }
                up2 = up1;
                up1 = dpp_mov<0x138, 0xf>(INFINITY, last_new);       // wave_shr:1 -- lane l <- lane l-1, lane 0 keeps +inf
            }
        }
#pragma unroll
        for (int u = 0; u < P; ++u) ring[u] = nxt[u];
    }
    __syncthreads();                                           // single wave: orders the trace stores before the walk
    if (lane == 0) {
        int ii = N, jj = M, n = 0;
        int* pt = path_text ? path_text + (size_t)b * (N + S + 2) : nullptr;    // full path only for the stand-alone cw_dtw
        int* pj = path_time ? path_time + (size_t)b * (N + S + 2) : nullptr;
        int wrow = -1, widx = -1; unsigned int word = 0u;
        while (ii > 0 || jj > 0) {
            if (pt) { pt[n] = ii - 1; pj[n] = jj - 1; }
            ++n;
            if (ii > 0 && jj > 0) first_col[(size_t)b * N + (ii - 1)] = jj - 1;
            unsigned int t;
            if (ii == 0) t = 2u;
            else if (jj == 0) t = 1u;
            else {
                const int q = jj - 1;
                if (wrow != ii || widx != (q >> 4)) { wrow = ii; widx = q >> 4; word = tr[(size_t)(ii - 1) * SW + widx]; }
                t = (word >> (2 * (q & 15))) & 3u;
            }
            if (t == 0u) { --ii; --jj; } else if (t == 1u) { --ii; } else { --jj; }
        }
        path_len[b] = n;   // stored in reverse order (end -> start); the host flips it
    }
}

__global__ void pauses_kernel(const double* __restrict__ start_in, const double* __restrict__ end_in,
                              double* __restrict__ start_out, double* __restrict__ end_out, int W, double thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    // word i's end is moved by boundary (i, i+1); its start by boundary (i-1, i) -- both computed from
    // the *original* neighbour fields, exactly what the sequential loop of REF/utils.py:8-26 reads.
    double s = start_in[i], e = end_in[i];
    if (i + 1 < W) {
        double pause = start_in[i + 1] - e;
        if (pause > 0) e = e + (pause > thr ? thr / 2 : pause / 2);
    }
    if (i > 0) {
        double pause = s - end_in[i - 1];
        if (pause > 0) s = s - (pause > thr ? thr / 2 : pause / 2);
    }
    start_out[i] = s; end_out[i] = e;
}

int cw_launch_align_stats(const float* w, int B, int Ha, int rows_cap, int S, int row0, int N, const int* n_cols,
                          float* mean, float* stdv, hipStream_t st) {
    hipLaunchKernelGGL(align_stats_kernel, dim3((S + 255) / 256, Ha, B), dim3(256), 0, st, w, Ha, rows_cap, S, row0, N,
                       n_cols, mean, stdv);
    return CW_OK;
}

int cw_launch_align_filter(const float* w, int B, int Ha, int rows_cap, int S, int row0, int N, const int* n_cols,
                           const float* mean, const float* stdv, int width, float* mat, hipStream_t st) {
    dim3 grid((S + FT - 1) / FT, N, B);
#define LAUNCH_W(WW) hipLaunchKernelGGL((align_filter_kernel<WW>), grid, dim3(FT), 0, st, w, Ha, rows_cap, S, row0, N, \
                                        n_cols, mean, stdv, mat)
    switch (width) {
        case 1: LAUNCH_W(1); break;
        case 3: LAUNCH_W(3); break;
        case 5: LAUNCH_W(5); break;
        case 7: LAUNCH_W(7); break;
        case 9: LAUNCH_W(9); break;
        default: return CW_ERR_INVALID;
    }
#undef LAUNCH_W
    return CW_OK;
}

size_t cw_dtw_skew_floats(int B, int N, int S) {           // workspace of the wave-local DTW: [B][N + S + 1][rows padded to 64 R]
    const int R = N <= 128 ? 2 : (N <= 256 ? 4 : 8);
    return (size_t)B * (size_t)(N + S + 1) * (R * 64);
}

int cw_launch_dtw(const float* mat, int B, int N, int S, const int* n_cols, unsigned char* trace, int* first_col,
                  int* path_text, int* path_time, int* path_len, hipStream_t st, float* skew) {
    if (N > DTW_THREADS || N <= 0) return CW_ERR_INVALID;
    if (!skew) {   // no skew workspace: the round-1 block kernel (contexts created under CW_DTW_BLOCK=1, A/B runs)
        hipLaunchKernelGGL(dtw_kernel, dim3(B), dim3(DTW_THREADS), 0, st, mat, N, S, n_cols, trace, first_col, path_text,
                           path_time, path_len);
        return CW_OK;
    }
    const int R = N <= 128 ? 2 : (N <= 256 ? 4 : 8), NP = R * 64, D_cap = N + S + 1;
    hipLaunchKernelGGL(dtw_skew_kernel, dim3(N + S - 1, 1, B), dim3(NP < 256 ? NP : 256), 0, st, mat, N, S, n_cols, NP, D_cap, skew);
    // trace words: N rows x ceil(S/16); in LDS when they fit (N = 128: 48 KB), else in the byte-trace buffer (N*S bytes >= that)
    const size_t words = (size_t)N * ((S + 15) >> 4);
    const int use_lds = words * 4 <= 150 * 1024 ? 1 : 0;
    const size_t lds = use_lds ? words * 4 : 0;
    const size_t stride_words = ((size_t)N * S) / 4;
#define CW_DTW_LAUNCH(RR, PP)                                                                                          \
    do {                                                                                                               \
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)dtw_wave_kernel<RR, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((dtw_wave_kernel<RR, PP>), dim3(B), dim3(64), lds, st, skew, N, S, D_cap, n_cols,            \
                           (unsigned int*)trace, stride_words, use_lds, first_col, path_text, path_time, path_len);   \
    } while (0)
    if (R == 2) CW_DTW_LAUNCH(2, 16);
    else if (R == 4) CW_DTW_LAUNCH(4, 16);
    else CW_DTW_LAUNCH(8, 8);
#undef CW_DTW_LAUNCH
    return CW_OK;
}

int cw_launch_pauses(double* start, double* end, int W, double thr, hipStream_t st) {
    // in: start/end ; out: start + W / end + W  (caller provides 2*W doubles each)
    hipLaunchKernelGGL(pauses_kernel, dim3((W + 255) / 256), dim3(256), 0, st, start, end, start + W, end + W, W, thr);
    return CW_OK;
}
