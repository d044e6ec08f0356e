/* DTW + backtrace, C restatement of TF/models/whisper/generation_whisper.py:64-115
 * (TEST ORACLE -- see oracle/__init__.py; never linked into the product).
 *
 * matrix: [N][M] float64 (the caller passes -weights.double()), cost/trace arrays float32
 * exactly like the reference (np.float32 arrays, float64 addend, result rounded to f32 on
 * store :87).  Tie rule :80-85: c0<c1&&c0<c2 -> diag(0); c1<c0&&c1<c2 -> up(1); else left(2).
 * Returns path length; text_idx/time_idx are written in forward order.
 */
#include <stdlib.h>
#include <math.h>

int cw_oracle_dtw(const double* m, int N, int M, int* text_idx, int* time_idx) {
    int W = M + 1;
    float* cost = (float*)malloc(sizeof(float) * (size_t)(N + 1) * W);
    signed char* trace = (signed char*)malloc((size_t)(N + 1) * W);
    if (!cost || !trace) { free(cost); free(trace); return -1; }
    for (long i = 0; i < (long)(N + 1) * W; ++i) { cost[i] = INFINITY; trace[i] = -1; }
    cost[0] = 0.0f;
    for (int j = 1; j <= M; ++j) {
        for (int i = 1; i <= N; ++i) {
            float c0 = cost[(i - 1) * W + (j - 1)];
            float c1 = cost[(i - 1) * W + j];
            float c2 = cost[i * W + (j - 1)];
            float c; signed char t;
            if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
            else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
            else { c = c2; t = 2; }
            cost[i * W + j] = (float)(m[(long)(i - 1) * M + (j - 1)] + (double)c);
            trace[i * W + j] = t;
        }
    }
    for (int j = 0; j <= M; ++j) trace[j] = 2;          /* :93 */
    for (int i = 0; i <= N; ++i) trace[i * W] = 1;      /* :94 */
    int i = N, j = M, n = 0;
    int cap = N + M + 2;
    int* ti = (int*)malloc(sizeof(int) * cap);
    int* tj = (int*)malloc(sizeof(int) * cap);
    while (i > 0 || j > 0) {
        ti[n] = i - 1; tj[n] = j - 1; ++n;
        signed char t = trace[i * W + j];
        if (t == 0) { --i; --j; } else if (t == 1) { --i; } else if (t == 2) { --j; }
        else { free(cost); free(trace); free(ti); free(tj); return -2; }  /* :108-111 */
    }
    for (int k = 0; k < n; ++k) { text_idx[k] = ti[n - 1 - k]; time_idx[k] = tj[n - 1 - k]; }
    free(cost); free(trace); free(ti); free(tj);
    return n;
}
