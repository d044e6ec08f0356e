// Row-wise kernels: LayerNorm, token embedding, and the fused logits-processing + greedy sampling
// kernel (TF/generation/logits_process.py:203-260, 1816-2047 + argmax TF/generation/utils.py:2925).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace CW_NS {

// One wave per row (rows of d_model f32 from the residual stream), vectorised float4 loads,
// two-pass statistics (mean, then variance) like nn.LayerNorm, eps = 1e-5.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, T* __restrict__ out, int rows,
                                                        int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    T* o = out + (size_t)row * d;
    if (d <= 2048) {   // row cached in registers: one HBM read (the 3-pass form measured 2.2x fetch)
        float4 v[8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (lane + 64 * c) * 4;
            v[c] = (k < d) ? *(const float4*)(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if ((lane + 64 * c) * 4 < d) {
                float a = v[c].x - mean, bb = v[c].y - mean, cc = v[c].z - mean, e = v[c].w - mean;
                q += (a * a + bb * bb) + (cc * cc + e * e);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k = (lane + 64 * c) * 4;
            if (k < d) {
                const float4 gg = *(const float4*)(g + k), bb = *(const float4*)(b + k);
                Pack4<T>::st(o + k, (v[c].x - mean) * rstd * gg.x + bb.x, (v[c].y - mean) * rstd * gg.y + bb.y,
                             (v[c].z - mean) * rstd * gg.z + bb.z, (v[c].w - mean) * rstd * gg.w + bb.w);
            }
        }
        return;
    }
    float s = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        float4 v = *(const float4*)(xr + k);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        float4 v = *(const float4*)(xr + k);
        float a = v.x - mean, bb = v.y - mean, c = v.z - mean, e = v.w - mean;
        q += (a * a + bb * bb) + (c * c + e * e);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    for (int k = lane * 4; k < d; k += 256) {
        float4 v = *(const float4*)(xr + k);
        float4 gg = *(const float4*)(g + k), bb = *(const float4*)(b + k);
        Pack4<T>::st(o + k, (v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                     (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
    }
}

int cw_launch_layernorm(bool bf16_out, const float* x, const float* g, const float* b, void* out, int rows, int d,
                        hipStream_t st) {
    if (d % 4 != 0 || rows <= 0) return CW_ERR_INVALID;
    dim3 grid((rows + 3) / 4);
    if (bf16_out)
        hipLaunchKernelGGL((layernorm_kernel<bf16_t>), grid, dim3(256), 0, st, x, g, b, (bf16_t*)out, rows, d);
    else
        hipLaunchKernelGGL((layernorm_kernel<float>), grid, dim3(256), 0, st, x, g, b, (float*)out, rows, d);
    return CW_OK;
}

// LayerNorm whose output feeds an e4m3 GEMM (opt-in fp8 encoder mode): the normalised row never leaves the registers in 16 bits --
// its maximum gives the row scale s = max|y| / 448, and y / s goes out as one byte per element (4 per lane and pass).
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, unsigned char* __restrict__ out,
                                                            float* __restrict__ scale, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = (lane + 64 * c) * 4;
        v[c] = (k < d) ? *(const float4*)(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if ((lane + 64 * c) * 4 < d) {
            float a = v[c].x - mean, bb = v[c].y - mean, cc = v[c].z - mean, e = v[c].w - mean;
            q += (a * a + bb * bb) + (cc * cc + e * e);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = (lane + 64 * c) * 4;
        if (k < d) {
            const float4 gg = *(const float4*)(g + k), bb = *(const float4*)(b + k);
            v[c].x = (v[c].x - mean) * rstd * gg.x + bb.x; v[c].y = (v[c].y - mean) * rstd * gg.y + bb.y;
            v[c].z = (v[c].z - mean) * rstd * gg.z + bb.z; v[c].w = (v[c].w - mean) * rstd * gg.w + bb.w;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[c].x), fabsf(v[c].y))), fmaxf(fabsf(v[c].z), fabsf(v[c].w)));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int k = (lane + 64 * c) * 4;
        if (k < d) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[c].x * inv, v[c].y * inv, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[c].z * inv, v[c].w * inv, w, true);
            *(int*)(out + (size_t)row * d + k) = w;
        }
    }
}
int cw_launch_layernorm_fp8(const float* x, const float* g, const float* b, void* out8, float* scale, int rows, int d, hipStream_t st) {
    if (d % 4 != 0 || d > 2048 || rows <= 0) return CW_ERR_INVALID;
    hipLaunchKernelGGL(layernorm_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, g, b, (unsigned char*)out8, scale, rows, d);
    return CW_OK;
}

int cw_launch_layernorm_f32(const float* x, const float* g, const float* b, float* out, int rows, int d,
                            hipStream_t st) {
    return cw_launch_layernorm(false, x, g, b, out, rows, d, st);
}

// x_out[b][:] = embed[ids[b][t]][:] + pos_embed[t][:]   (TF modeling_whisper.py:737, 754-762)
template <typename T>
__global__ void embed_kernel(const int* __restrict__ ids, int ids_stride, int t, const T* __restrict__ embed,
                             const float* __restrict__ pos_embed, float* __restrict__ x_out, int d) {
    const int b = blockIdx.x;
    const int tok = ids[(size_t)b * ids_stride + t];
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float v = Act<T>::ld(embed + (size_t)tok * d + k) + pos_embed[(size_t)t * d + k];
        x_out[(size_t)b * d + k] = sizeof(T) == 2 ? resid_grid(v) : v;     // bf16 engine: residual stream on the 2^-12 grid
    }
}

__global__ void set_pos_kernel(int* pos, int value, int B, unsigned int* epoch) {
    if ((int)threadIdx.x < B) pos[threadIdx.x] = value;
    if (epoch && threadIdx.x == 0) *epoch += 1u;   // one decoder forward follows (declayer.hip: granule tags)
}
int cw_launch_set_pos(int* pos, int value, int B, hipStream_t st, unsigned int* epoch) {
    hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(64), 0, st, pos, value, B, epoch);
    return CW_OK;
}

int cw_launch_embed(const int* ids, int ids_stride, int t, const void* embed, int embed_bf16, const float* pos_embed,
                    float* x_out, int B, int d, hipStream_t st) {
    if (embed_bf16)
        hipLaunchKernelGGL((embed_kernel<bf16_t>), dim3(B), dim3(256), 0, st, ids, ids_stride, t,
                           (const bf16_t*)embed, pos_embed, x_out, d);
    else
        hipLaunchKernelGGL((embed_kernel<float>), dim3(B), dim3(256), 0, st, ids, ids_stride, t, (const float*)embed,
                           pos_embed, x_out, d);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused logits processors + greedy choice + next-step embedding.  One block (1024 threads) per row.
//
// Effective score s(v) after, in HF's order: MinNewTokensLength -> SuppressTokensAtBegin ->
// SuppressTokens -> WhisperTimeStamp.  All of them only write -inf, so they commute and collapse
// into one predicate.  The timestamp "logsumexp rule" (logits_process.py:2040-2045):
//     logsumexp(logp[tb:]) > max(logp[:tb])   <=>   log(sum_{v>=tb} exp(s_v - M)) > max_text - M
// with M the row max, so the partition function cancels and never has to be computed.
// ---------------------------------------------------------------------------------------------------
struct ArgPair { float v; int i; };
__device__ inline ArgPair arg_better(ArgPair a, ArgPair b) {  // larger value wins, ties -> lower index
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
template <int CTRL, int ROW_MASK>
__device__ inline ArgPair dpp_pair(ArgPair a) {   // identity (-inf, INT_MAX) where the DPP source lane does not exist
    ArgPair b;
    b.v = dpp_mov<CTRL, ROW_MASK>(-INFINITY, a.v);
    b.i = __builtin_amdgcn_update_dpp(0x7fffffff, a.i, CTRL, ROW_MASK, 0xf, false);
    return b;
}
__device__ inline ArgPair wave_argmax(ArgPair a) {   // same DPP scan as wave_sum; every lane gets lane 63's total
    a = arg_better(a, dpp_pair<0x111, 0xf>(a));
    a = arg_better(a, dpp_pair<0x112, 0xf>(a));
    a = arg_better(a, dpp_pair<0x114, 0xf>(a));
    a = arg_better(a, dpp_pair<0x118, 0xf>(a));
    a = arg_better(a, dpp_pair<0x142, 0xa>(a));
    a = arg_better(a, dpp_pair<0x143, 0xc>(a));
    ArgPair r;
    r.v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.v), 63));
    r.i = __builtin_amdgcn_readlane(a.i, 63);
    return r;
}

// Stage 1 of the sampler: the vocabulary is cut into SAMPLE_NS slices, one block each (a single 1024-thread block per
// row took 21 us of every decode step for 200 KB of logits: 8 blocks on a 256-CU chip).  Per slice: best allowed text
// token, best allowed timestamp token, and sum over allowed timestamps of exp(s - slice max); stage 2 (sample_kernel)
// merges the SAMPLE_NS records of its row.
#define SAMPLE_NS 16
struct SamplePart { float bt_v; int bt_i; float bs_v; int bs_i; float ts_sum; float pad[3]; };
__global__ __launch_bounds__(256) void sample_partial_kernel(SampleParams p, SamplePart* __restrict__ part) {
    __shared__ float s_f[64];
    __shared__ int s_i[64];
    const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = p.logits + (size_t)b * p.ldv;
    const int* ids = p.ids + (size_t)b * p.ids_stride;
    const int n_prompt = p.cfg[0], min_new_tokens = p.cfg[1];
    const int t = p.pos[b] + 1, tb = p.timestamp_begin;
    const int n_gen = t - n_prompt;
    const bool last_ts = n_gen >= 1 && ids[t - 1] >= tb;
    const bool penult_ts = n_gen < 2 || ids[t - 2] >= tb;
    const int last_tok = p.last_ts_tok[b];
    const int ts_floor = (last_tok >= 0) ? ((last_ts && !penult_ts) ? last_tok : last_tok + 1) : tb;
    const bool at_begin = (n_gen == 0);
    const int ts_cap = (at_begin && p.max_initial_timestamp_index >= 0) ? tb + p.max_initial_timestamp_index : 0x7fffffff;
    const int per4 = ((p.ldv >> 2) + SAMPLE_NS - 1) / SAMPLE_NS;           // float4 groups per slice
    const int lo4 = sl * per4, hi4 = min(p.ldv >> 2, lo4 + per4);
    const float4* lg4 = (const float4*)lg;
    const uchar4* mk4 = (const uchar4*)p.mask;
    ArgPair bt = {-INFINITY, 0x7fffffff}, bs = {-INFINITY, 0x7fffffff};
    float sv[4][4]; int nmine = 0;                                         // this thread's allowed timestamp scores
    float tv[4][4];                                                        // ... and allowed text scores (p.lp_sum only)
    for (int i4 = lo4 + tid, it = 0; i4 < hi4; i4 += 256, ++it) {
        const float4 x = lg4[i4]; const uchar4 mk = mk4[i4];
        const float xs[4] = {x.x, x.y, x.z, x.w};
        const unsigned char ms[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = i4 * 4 + j;
            float val = -INFINITY;
            if (v < p.V) {
                bool dead = (ms[j] & 1) || (at_begin && (ms[j] & 2));
                dead |= (v == p.eos && n_gen < min_new_tokens);
                if (last_ts) dead |= penult_ts ? (v >= tb) : (v < p.eos);
                dead |= (v >= tb && v < ts_floor);
                if (at_begin) dead |= (v < tb) || (v > ts_cap);
                if (!dead) val = xs[j];
                ArgPair c = {val, v};
                if (v < tb) bt = arg_better(bt, c); else bs = arg_better(bs, c);
            }
            if (it < 4) { sv[it][j] = (v >= tb && v < p.V) ? val : -INFINITY; tv[it][j] = (v < tb) ? val : -INFINITY; }
        }
        nmine = it + 1;
    }
    bt = wave_argmax(bt);
    bs = wave_argmax(bs);
    if (lane == 0) { s_f[wave] = bt.v; s_i[wave] = bt.i; s_f[32 + wave] = bs.v; s_i[32 + wave] = bs.i; }
    __syncthreads();
    bt = {-INFINITY, 0x7fffffff}; bs = {-INFINITY, 0x7fffffff};
    for (int w = 0; w < 4; ++w) {
        bt = arg_better(bt, ArgPair{s_f[w], s_i[w]});
        bs = arg_better(bs, ArgPair{s_f[32 + w], s_i[32 + w]});
    }
    float acc = 0.f;
    if (bs.v > -INFINITY) {
        for (int it = 0; it < nmine && it < 4; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (sv[it][j] > -INFINITY) acc += expf(sv[it][j] - bs.v);
    }
    __syncthreads();
    acc = block_sum(acc, s_f);
    float acc_t = 0.f;                                          // sum over allowed text tokens of exp(s - slice max), on request
    if (p.lp_sum) {
        if (bt.v > -INFINITY) {
            for (int it = 0; it < nmine && it < 4; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) if (tv[it][j] > -INFINITY) acc_t += expf(tv[it][j] - bt.v);
        }
        __syncthreads();
        acc_t = block_sum(acc_t, s_f);
    }
    if (tid == 0 && b == 0 && sl == 0) {
        *p.n_unfinished = 0;                                     // stage 2 (a later launch) counts the running rows into it
        if (p.epoch) *p.epoch += 1u;                             // the next decoder forward tags its granules with a fresh epoch (declayer.hip)
    }
    if (tid == 0) {
        SamplePart o; o.bt_v = bt.v; o.bt_i = bt.i; o.bs_v = bs.v; o.bs_i = bs.i; o.ts_sum = acc; o.pad[0] = acc_t; o.pad[1] = o.pad[2] = 0.f;
        part[(size_t)b * SAMPLE_NS + sl] = o;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void sample_kernel(SampleParams p) {
    __shared__ float s_f[64];
    __shared__ int s_i[64];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* lg = p.logits + (size_t)b * p.ldv;
    int* ids = p.ids + (size_t)b * p.ids_stride;
    const int n_prompt = p.cfg[0], min_new_tokens = p.cfg[1], max_length = p.cfg[2], use_forced = p.cfg[3];
    const int t = p.pos[b] + 1, tb = p.timestamp_begin;
    const int n_gen = t - n_prompt;

    const bool was_finished = p.finished[b] != 0;
    int forced = (p.forced && use_forced) ? p.forced[(size_t)b * p.ids_stride + t] : -1;

    // timestamp grammar state from the generated suffix
    const bool last_ts = n_gen >= 1 && ids[t - 1] >= tb;
    const bool penult_ts = n_gen < 2 || ids[t - 2] >= tb;
    const int last_tok = p.last_ts_tok[b];
    const int ts_floor = (last_tok >= 0) ? ((last_ts && !penult_ts) ? last_tok : last_tok + 1) : tb;
    const bool at_begin = (n_gen == 0);
    const int ts_cap = (at_begin && p.max_initial_timestamp_index >= 0) ? tb + p.max_initial_timestamp_index
                                                                         : 0x7fffffff;
    // stage 2: merge the SAMPLE_NS slice records of this row (sample_partial_kernel)
    ArgPair bt = {-INFINITY, 0x7fffffff}, bs = {-INFINITY, 0x7fffffff};
    const SamplePart* pr = (const SamplePart*)p.partials + (size_t)b * SAMPLE_NS;
    for (int i = 0; i < SAMPLE_NS; ++i) {
        bt = arg_better(bt, ArgPair{pr[i].bt_v, pr[i].bt_i});
        bs = arg_better(bs, ArgPair{pr[i].bs_v, pr[i].bs_i});
    }
    const float M = fmaxf(bt.v, bs.v);
    float acc = 0.f;                                   // sum over allowed timestamps of exp(s - M)
    for (int i = 0; i < SAMPLE_NS; ++i)
        if (pr[i].bs_v > -INFINITY) acc += pr[i].ts_sum * expf(pr[i].bs_v - M);

    if (tid == 0) {
        bool force_ts = (acc > 0.f) && (logf(acc) > bt.v - M);
        int choice;
        if (force_ts || !(bt.v > -INFINITY)) choice = bs.i;
        else choice = arg_better(bt, bs).i;
        // no finite candidate at all (every token masked, or NaN logits): torch.argmax of an all -inf row is index 0;
        // never hand an out-of-range id to the next step's embedding gather
        if (!(M > -INFINITY) || choice < 0 || choice >= p.V) choice = 0;
        if (p.argmax_trace) p.argmax_trace[(size_t)b * p.ids_stride + t] = choice;
        int tok = (forced >= 0) ? forced : choice;
        if (p.lp_sum && !was_finished && n_gen >= 0) {
            // log_softmax of the PROCESSED scores at the chosen token (generation_whisper.py:1967-1971): suppressed tokens
            // are -inf there, and so is every text token once the timestamp rule has fired
            float st = 0.f;
            for (int i = 0; i < SAMPLE_NS; ++i)
                if (pr[i].bt_v > -INFINITY) st += pr[i].pad[0] * expf(pr[i].bt_v - M);
            const float tot = acc + (force_ts ? 0.f : st);
            if (tot > 0.f && tok >= 0 && tok < p.V) {
                p.lp_sum[b] += lg[tok] - (M + logf(tot));
                p.lp_cnt[b] += 1;
            }
        }
        if (was_finished) tok = p.pad;                                  // utils.py:2928-2929
        ids[t] = tok;
        if (tok >= tb && n_gen >= 0) p.last_ts_tok[b] = tok;
        int fin = was_finished || (n_gen >= 0 && tok == p.eos) || (t + 1 >= max_length);
        p.finished[b] = fin;
        if (!fin) atomicAdd(p.n_unfinished, 1);
        s_tok = tok;
        p.pos[b] = t;                                                   // next decoder input position
    }
    __syncthreads();
    // embedding for the next decoder step: token at sequence index t is fed at position t
    const int tok = s_tok;
    if (p.x_out && t < max_length) {
        const T* e = (const T*)p.embed + (size_t)tok * p.d;
        const float* pe = p.pos_embed + (size_t)t * p.d;
        for (int k = tid; k < p.d; k += blockDim.x) {
            const float v = Act<T>::ld(e + k) + pe[k];
            p.x_out[(size_t)b * p.d + k] = sizeof(T) == 2 ? resid_grid(v) : v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Beam search, device side (TF/generation/utils.py:3208-3520 keeps the bookkeeping on the host side of the C ABI too).
//
// beam_topk_kernel: one block per hypothesis row.  log_probs = log_softmax(raw logits) (:3402), the logits processors
// run on those log-probabilities (:3403; the same -inf predicate as the greedy kernel, the timestamp rule is invariant
// under the shift), and the n_cand best (value, token) pairs of the row are emitted in (value desc, token asc) order.
// The host adds the running beam scores and takes the top 2 * num_beams of the union: the global top-k over
// num_beams * vocab candidates (:3147) is contained in the union of the per-row top-k lists.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void beam_topk_kernel(SampleParams p, int n_cand, float* __restrict__ cand_val,
                                                         int* __restrict__ cand_id) {
    __shared__ float s_f[64];
    __shared__ int s_i[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* lg = p.logits + (size_t)b * p.ldv;
    const int* ids = p.ids + (size_t)b * p.ids_stride;
    const int n_prompt = p.cfg[0], min_new_tokens = p.cfg[1];
    const int t = p.pos[b] + 1, tb = p.timestamp_begin;
    const int n_gen = t - n_prompt;
    const bool last_ts = n_gen >= 1 && ids[t - 1] >= tb;
    const bool penult_ts = n_gen < 2 || ids[t - 2] >= tb;
    // last timestamp token generated so far (timestamps never decrease, so it is the maximum)
    float lt = -1.f;
    for (int k = n_prompt + tid; k < t; k += blockDim.x) if (ids[k] >= tb) lt = fmaxf(lt, (float)ids[k]);
    const int last_tok = (int)block_max(lt, s_f);
    __syncthreads();
    const int ts_floor = (last_tok >= 0) ? ((last_ts && !penult_ts) ? last_tok : last_tok + 1) : tb;
    const bool at_begin = (n_gen == 0);
    const int ts_cap = (at_begin && p.max_initial_timestamp_index >= 0) ? tb + p.max_initial_timestamp_index : 0x7fffffff;
    auto dead = [&](int v) -> bool {
        const unsigned char mk = p.mask[v];
        bool d = (mk & 1) || (at_begin && (mk & 2));
        d |= (v == p.eos && n_gen < min_new_tokens);
        if (last_ts) d |= penult_ts ? (v >= tb) : (v < p.eos);
        d |= (v >= tb && v < ts_floor);
        if (at_begin) d |= (v < tb) || (v > ts_cap);
        return d;
    };
    // raw max / sum (log_softmax normaliser over the whole vocabulary), best allowed text / timestamp value
    float rmax = -INFINITY, btext = -INFINITY, bts = -INFINITY;
    for (int v = tid; v < p.V; v += blockDim.x) {
        const float x = lg[v];
        rmax = fmaxf(rmax, x);
        if (!dead(v)) { if (v < tb) btext = fmaxf(btext, x); else bts = fmaxf(bts, x); }
    }
    rmax = block_max(rmax, s_f); __syncthreads();
    btext = block_max(btext, s_f); __syncthreads();
    bts = block_max(bts, s_f); __syncthreads();
    float rsum = 0.f, tsum = 0.f;
    const float M = fmaxf(btext, bts);
    for (int v = tid; v < p.V; v += blockDim.x) {
        const float x = lg[v];
        rsum += expf(x - rmax);
        if (v >= tb && !dead(v)) tsum += expf(x - M);
    }
    rsum = block_sum(rsum, s_f); __syncthreads();
    tsum = block_sum(tsum, s_f); __syncthreads();
    const float logz = logf(rsum);
    const bool force_ts = (tsum > 0.f) && (logf(tsum) > btext - M);
    // n_cand rounds of block-wide selection in (value desc, token asc) order
    ArgPair prev = {INFINITY, -1};
    for (int r = 0; r < n_cand; ++r) {
        ArgPair best = {-INFINITY, 0x7fffffff};
        for (int v = tid; v < p.V; v += blockDim.x) {
            if (dead(v) || (force_ts && v < tb)) continue;
            const float x = lg[v];
            if (!(x > -INFINITY)) continue;
            const bool after = (x < prev.v) || (x == prev.v && v > prev.i);
            if (after) best = arg_better(best, ArgPair{x, v});
        }
        best = wave_argmax(best);
        __syncthreads();
        if (lane == 0) { s_f[wave] = best.v; s_i[wave] = best.i; }
        __syncthreads();
        best = {-INFINITY, 0x7fffffff};
        for (int w = 0; w < nw; ++w) best = arg_better(best, ArgPair{s_f[w], s_i[w]});
        if (tid == 0) {
            const bool ok = best.v > -INFINITY;
            cand_val[(size_t)b * n_cand + r] = ok ? (best.v - rmax) - logz : -INFINITY;    // torch: (x - max) - log(sum)
            cand_id[(size_t)b * n_cand + r] = ok ? best.i : -1;
        }
        prev = best;
        if (!(best.v > -INFINITY)) {                      // fewer allowed tokens than n_cand: pad the rest
            for (int r2 = r + 1 + tid; r2 < n_cand; r2 += blockDim.x) {
                cand_val[(size_t)b * n_cand + r2] = -INFINITY; cand_id[(size_t)b * n_cand + r2] = -1;
            }
            break;
        }
    }
}

// Two-stage form of the same selection (the default): the single block per row above walks the 51 866-entry row n_cand + 2
// times with 40 blocks on 256 CUs -- 260 us per step at 8 items x 5 beams.  Stage 1 cuts every row into BT_NS slices
// (grid slices x rows, 13 logits per thread held in registers): per slice the log-softmax partials (raw max, sum of
// exponentials), the best allowed text / timestamp values with the timestamp mass relative to the slice's own timestamp
// maximum, and two sorted candidate lists (allowed text tokens, allowed timestamp tokens -- the row-wide "force a timestamp"
// decision is only known in stage 2, which then simply ignores the text lists).  Stage 2 (one wave per row) merges the
// partials exactly like the greedy sampler does and selects the n_cand best of the <= 2 * BT_NS * n_cand listed candidates.
// Record layout per (row, slice), BT_REC floats: [0] raw max, [1] sum exp(x - raw max), [2] best text, [3] best timestamp,
// [4] sum over allowed timestamps exp(x - best timestamp), then text values[64], text ids[64], timestamp values[64], ids[64].
#define BT_NS 16
#define BT_REC (8 + 4 * 64)
#define BT_PER_LANE 13      // ceil(ceil(51866 / 16) / 256); larger vocabularies fall back to the single-block kernel
__global__ __launch_bounds__(256) void beam_topk_partial_kernel(SampleParams p, int n_cand, float* __restrict__ scratch) {
    __shared__ float s_f[8];
    __shared__ int s_i[8];
    const int sl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lg = p.logits + (size_t)b * p.ldv;
    const int* ids = p.ids + (size_t)b * p.ids_stride;
    const int n_prompt = p.cfg[0], min_new_tokens = p.cfg[1];
    const int t = p.pos[b] + 1, tb = p.timestamp_begin;
    const int n_gen = t - n_prompt;
    const bool last_ts = n_gen >= 1 && ids[t - 1] >= tb;
    const bool penult_ts = n_gen < 2 || ids[t - 2] >= tb;
    float lt = -1.f;   // last timestamp token generated so far (timestamps never decrease, so it is the maximum)
    for (int k = n_prompt + tid; k < t; k += blockDim.x) if (ids[k] >= tb) lt = fmaxf(lt, (float)ids[k]);
    const int last_tok = (int)block_max(lt, s_f);
    __syncthreads();
    const int ts_floor = (last_tok >= 0) ? ((last_ts && !penult_ts) ? last_tok : last_tok + 1) : tb;
    const bool at_begin = (n_gen == 0);
    const int ts_cap = (at_begin && p.max_initial_timestamp_index >= 0) ? tb + p.max_initial_timestamp_index : 0x7fffffff;
    auto dead = [&](int v, unsigned char mk) -> bool {
        bool d = (mk & 1) || (at_begin && (mk & 2));
        d |= (v == p.eos && n_gen < min_new_tokens);
        if (last_ts) d |= penult_ts ? (v >= tb) : (v < p.eos);
        d |= (v >= tb && v < ts_floor);
        if (at_begin) d |= (v < tb) || (v > ts_cap);
        return d;
    };
    const int per = (p.V + BT_NS - 1) / BT_NS;
    const int lo = sl * per, hi = min(p.V, lo + per);
    float x[BT_PER_LANE];        // allowed value, -inf when dead or out of range
    float rawv[BT_PER_LANE];     // the logit itself (-inf out of range)
    unsigned char mkv[BT_PER_LANE];
    float rmax = -INFINITY, btext = -INFINITY, bts = -INFINITY;
    // every load of the block first, unconditional with a clamped index: behind `if (v < hi)` each logit and each mask byte was
    // its own exec-masked block with its own wait -- 26 memory round trips in a row, 46 us for a 200 KB slice
#pragma unroll
    for (int i = 0; i < BT_PER_LANE; ++i) {
        const int vc = min(lo + tid + i * 256, hi - 1);
        rawv[i] = lg[vc];
        mkv[i] = p.mask[vc];
    }
#pragma unroll
    for (int i = 0; i < BT_PER_LANE; ++i) {
        const int v = lo + tid + i * 256;
        float raw = -INFINITY;
        x[i] = -INFINITY;
        if (v < hi) {
            raw = rawv[i];
            if (!dead(v, mkv[i])) x[i] = raw;
        }
        rawv[i] = raw;
        rmax = fmaxf(rmax, raw);
        if (v < tb) btext = fmaxf(btext, x[i]); else bts = fmaxf(bts, x[i]);
    }
    // the three maxima in one exchange, the two sums in another (each was its own block reduction: 15 barriers of this
    // latency-bound kernel; the sums add the waves' partials in the same order as block_sum, so the records are unchanged)
    __shared__ float s_mx[3][4], s_sm[2][4];
    rmax = wave_max(rmax); btext = wave_max(btext); bts = wave_max(bts);
    if (lane == 0) { s_mx[0][wave] = rmax; s_mx[1][wave] = btext; s_mx[2][wave] = bts; }
    __syncthreads();
    rmax = fmaxf(fmaxf(s_mx[0][0], s_mx[0][1]), fmaxf(s_mx[0][2], s_mx[0][3]));
    btext = fmaxf(fmaxf(s_mx[1][0], s_mx[1][1]), fmaxf(s_mx[1][2], s_mx[1][3]));
    bts = fmaxf(fmaxf(s_mx[2][0], s_mx[2][1]), fmaxf(s_mx[2][2], s_mx[2][3]));
    float rsum = 0.f, tsum = 0.f;
#pragma unroll
    for (int i = 0; i < BT_PER_LANE; ++i) {
        const int v = lo + tid + i * 256;
        if (v < hi) {
            rsum += expf(rawv[i] - rmax);
            if (v >= tb && x[i] > -INFINITY) tsum += expf(x[i] - bts);
        }
    }
    rsum = wave_sum(rsum); tsum = wave_sum(tsum);
    if (lane == 0) { s_sm[0][wave] = rsum; s_sm[1][wave] = tsum; }
    __syncthreads();
    rsum = 0.f; tsum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { rsum += s_sm[0][w]; tsum += s_sm[1][w]; }
    float* rec = scratch + ((size_t)b * BT_NS + sl) * BT_REC;
    if (tid == 0) { rec[0] = rmax; rec[1] = rsum; rec[2] = btext; rec[3] = bts; rec[4] = tsum; }
    // two lists (text tokens, timestamp tokens), each n_cand rounds of block-wide selection in (value desc, token asc) order
    __shared__ float s_sf[2][4];                                             // selection rounds alternate between two exchange
    __shared__ int s_si[2][4];                                               // buffers: one barrier per round instead of two
    // the winners of the rounds are collected in LDS and written out once per list: a global store inside the round made the
    // round's barrier wait for it (vmcnt(0) in front of every s_barrier), a memory round trip per candidate
    __shared__ float s_lv[2][64];
    __shared__ int s_li[2][64];
    int rr = 0;
    for (int list = 0; list < 2; ++list) {
        float* lv = rec + 8 + list * 128;
        int* li = (int*)(lv + 64);
        if ((list == 0 && lo >= tb) || (list == 1 && hi <= tb)) {          // slice holds no token of this kind (block-uniform)
            for (int r = tid; r < n_cand; r += 256) { lv[r] = -INFINITY; li[r] = -1; }
            continue;
        }
        for (int r = tid; r < n_cand; r += 256) { s_lv[list][r] = -INFINITY; s_li[list][r] = -1; }   // rounds that never run: padding
        __syncthreads();
        ArgPair prev = {INFINITY, -1};
        for (int r = 0; r < n_cand; ++r) {
            ArgPair best = {-INFINITY, 0x7fffffff};
#pragma unroll
            for (int i = 0; i < BT_PER_LANE; ++i) {
                const int v = lo + tid + i * 256;
                const bool kind = (list == 0) ? (v < tb) : (v >= tb);
                const bool after = (x[i] < prev.v) || (x[i] == prev.v && v > prev.i);
                if (kind && after && x[i] > -INFINITY) best = arg_better(best, ArgPair{x[i], v});
            }
            best = wave_argmax(best);
            const int bufI = rr & 1; ++rr;
            if (lane == 0) { s_sf[bufI][wave] = best.v; s_si[bufI][wave] = best.i; }
            __syncthreads();
            best = {-INFINITY, 0x7fffffff};
#pragma unroll
            for (int w = 0; w < 4; ++w) best = arg_better(best, ArgPair{s_sf[bufI][w], s_si[bufI][w]});
            const bool ok = best.v > -INFINITY;
            if (tid == 0 && ok) { s_lv[list][r] = best.v; s_li[list][r] = best.i; }
            prev = best;
            if (!ok) break;                                                  // fewer allowed tokens than n_cand: the rest stays padded
        }
        __syncthreads();
        for (int r = tid; r < n_cand; r += 256) { lv[r] = s_lv[list][r]; li[r] = s_li[list][r]; }
    }
}

// stage 2: one wave per row
__global__ __launch_bounds__(64) void beam_topk_merge_kernel(const float* __restrict__ scratch, int n_cand,
                                                             float* __restrict__ cand_val, int* __restrict__ cand_id) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* rec0 = scratch + (size_t)b * BT_NS * BT_REC;
    float rmax = -INFINITY, btext = -INFINITY, bts = -INFINITY;
    if (lane < BT_NS) { rmax = rec0[lane * BT_REC]; btext = rec0[lane * BT_REC + 2]; bts = rec0[lane * BT_REC + 3]; }
    const float gmax = wave_max(rmax), gtext = wave_max(btext), gts = wave_max(bts);
    const float M = fmaxf(gtext, gts);
    float rsum = 0.f, tsum = 0.f;
    if (lane < BT_NS) {
        rsum = rec0[lane * BT_REC + 1] * expf(rmax - gmax);
        if (bts > -INFINITY) tsum = rec0[lane * BT_REC + 4] * expf(bts - M);
    }
    rsum = wave_sum(rsum); tsum = wave_sum(tsum);
    const float logz = logf(rsum);
    const bool force_ts = (tsum > 0.f) && (logf(tsum) > gtext - M);
    // candidates: lane l looks after list entries e = l, l + 64, ... of the 2 * BT_NS lists of n_cand entries
    const int total = 2 * BT_NS * n_cand;
    ArgPair prev = {INFINITY, -1};
    for (int r = 0; r < n_cand; ++r) {
        ArgPair best = {-INFINITY, 0x7fffffff};
        for (int e = lane; e < total; e += 64) {
            const int li = e / n_cand, k = e - li * n_cand;        // li = slice * 2 + list
            if (force_ts && !(li & 1)) continue;
            const float* lv = rec0 + (size_t)(li >> 1) * BT_REC + 8 + (li & 1) * 128;
            const float xv = lv[k];
            const int vi = ((const int*)(lv + 64))[k];
            if (!(xv > -INFINITY)) continue;
            const bool after = (xv < prev.v) || (xv == prev.v && vi > prev.i);
            if (after) best = arg_better(best, ArgPair{xv, vi});
        }
        best = wave_argmax(best);
        const bool ok = best.v > -INFINITY;
        if (lane == 0) {
            cand_val[(size_t)b * n_cand + r] = ok ? (best.v - gmax) - logz : -INFINITY;    // torch: (x - max) - log(sum)
            cand_id[(size_t)b * n_cand + r] = ok ? best.i : -1;
        }
        prev = best;
        if (!ok) {
            for (int r2 = r + 1 + lane; r2 < n_cand; r2 += 64) { cand_val[(size_t)b * n_cand + r2] = -INFINITY; cand_id[(size_t)b * n_cand + r2] = -1; }
            break;
        }
    }
}

static int g_topk_1block = -1;   // 1: the single-block kernel (differential tests / A-B); -1: from the environment
void cw_beam_topk_set_1block(int on) { g_topk_1block = on; }
size_t cw_beam_topk_scratch_floats(int rows) { return (size_t)rows * BT_NS * BT_REC; }

int cw_launch_beam_topk(const SampleParams& p, int n_cand, float* cand_val, int* cand_id, float* scratch, hipStream_t st) {
    if (n_cand < 1 || n_cand > 64) return CW_ERR_INVALID;
    if (g_topk_1block < 0) g_topk_1block = cw_sw::cw_switches().beam_topk_1block;
    const bool one_block = g_topk_1block != 0;
    if (scratch && !one_block && (p.V + BT_NS - 1) / BT_NS <= BT_PER_LANE * 256) {
        hipLaunchKernelGGL(beam_topk_partial_kernel, dim3(BT_NS, p.B), dim3(256), 0, st, p, n_cand, scratch);
        hipLaunchKernelGGL(beam_topk_merge_kernel, dim3(p.B), dim3(64), 0, st, (const float*)scratch, n_cand, cand_val, cand_id);
        return CW_OK;
    }
    if (p.embed_bf16) hipLaunchKernelGGL((beam_topk_kernel<bf16_t>), dim3(p.B), dim3(1024), 0, st, p, n_cand, cand_val, cand_id);
    else hipLaunchKernelGGL((beam_topk_kernel<float>), dim3(p.B), dim3(1024), 0, st, p, n_cand, cand_val, cand_id);
    return CW_OK;
}

// rows <- rows[parent]: token history, cache ancestry (+ the slot the parent just wrote), then the chosen token, its
// embedding for the next decoder step and the new position.  Two passes through a scratch copy (rows read each other).
__global__ void beam_gather_kernel(BeamAdvanceParams p) {
    const int r = blockIdx.x, q = p.parent[r];
    const int t = p.pos[q] + 1;                                       // all rows share the position
    for (int k = threadIdx.x; k < p.ids_stride; k += blockDim.x) p.ids_tmp[(size_t)r * p.ids_stride + k] = (k < t) ? p.ids[(size_t)q * p.ids_stride + k] : (k == t ? p.token[r] : p.ids[(size_t)r * p.ids_stride + k]);
    for (int k = threadIdx.x; k < p.cap; k += blockDim.x)
        p.anc_tmp[(size_t)r * p.cap + k] = (k < t - 1) ? p.anc[(size_t)q * p.cap + k] : (k == t - 1 ? q : r);
}
template <typename T>
__global__ void beam_commit_kernel(BeamAdvanceParams p) {
    const int r = blockIdx.x;
    const int t = p.pos[r] + 1;
    for (int k = threadIdx.x; k < p.ids_stride; k += blockDim.x) p.ids[(size_t)r * p.ids_stride + k] = p.ids_tmp[(size_t)r * p.ids_stride + k];
    for (int k = threadIdx.x; k < p.cap; k += blockDim.x) p.anc[(size_t)r * p.cap + k] = p.anc_tmp[(size_t)r * p.cap + k];
    const int tok = p.token[r];
    if (t < p.cap) {
        const T* e = (const T*)p.embed + (size_t)tok * p.d;
        const float* pe = p.pos_embed + (size_t)t * p.d;
        for (int k = threadIdx.x; k < p.d; k += blockDim.x) {
            const float v = Act<T>::ld(e + k) + pe[k];
            p.x_out[(size_t)r * p.d + k] = sizeof(T) == 2 ? resid_grid(v) : v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) p.pos[r] = t;
}
int cw_launch_beam_advance(const BeamAdvanceParams& p, hipStream_t st) {
    hipLaunchKernelGGL(beam_gather_kernel, dim3(p.rows), dim3(256), 0, st, p);
    if (p.embed_bf16) hipLaunchKernelGGL((beam_commit_kernel<bf16_t>), dim3(p.rows), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((beam_commit_kernel<float>), dim3(p.rows), dim3(256), 0, st, p);
    return CW_OK;
}

// out[i][a][p][:] = align[row_of_pos[i][p]][a][p][:]  (generation_whisper.py:262-303: the cross-attention row of output
// position p comes from the beam that produced it).  grid (L, n_align, n_items)
__global__ void align_gather_kernel(const float* __restrict__ align, const int* __restrict__ row_of_pos, int n_align,
                                    int align_rows, int L, int n_keys, float* __restrict__ out) {
    const int pI = blockIdx.x, a = blockIdx.y, i = blockIdx.z;
    const int src = row_of_pos[(size_t)i * L + pI];
    const float4* s4 = (const float4*)(align + (((size_t)src * n_align + a) * align_rows + pI) * n_keys);
    float4* d4 = (float4*)(out + (((size_t)i * n_align + a) * align_rows + pI) * n_keys);
    for (int k = threadIdx.x; k < n_keys / 4; k += blockDim.x) d4[k] = s4[k];
}
int cw_launch_align_gather(const float* align, const int* row_of_pos, int n_items, int n_align, int align_rows, int L,
                           int n_keys, float* out, hipStream_t st) {
    if (L <= 0 || n_keys % 4) return L <= 0 ? CW_OK : CW_ERR_INVALID;
    hipLaunchKernelGGL(align_gather_kernel, dim3(L, n_align, n_items), dim3(128), 0, st, align, row_of_pos, n_align, align_rows,
                       L, n_keys, out);
    return CW_OK;
}

int cw_launch_sample(const SampleParams& p, hipStream_t st) {
    if (!p.partials || (p.ldv >> 2) > SAMPLE_NS * 1024 || (p.ldv & 3)) return CW_ERR_INVALID;
    hipLaunchKernelGGL(sample_partial_kernel, dim3(p.B, SAMPLE_NS), dim3(256), 0, st, p, (SamplePart*)p.partials);
    if (p.embed_bf16)
        hipLaunchKernelGGL((sample_kernel<bf16_t>), dim3(p.B), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((sample_kernel<float>), dim3(p.B), dim3(256), 0, st, p);
    return CW_OK;
}

}  // namespace CW_NS
