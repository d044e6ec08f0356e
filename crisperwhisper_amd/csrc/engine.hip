// Engine + C ABI (include/crisperwhisper.h): owns device memory, the HIP stream, the weights and the
// per-stage orchestration of the CrisperWhisper hot path on one MI355X.  No CPU fallbacks: every entry
// point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <string>
#include <vector>
#include <functional>

#include "../../include/crisperwhisper.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

// dtype dispatch: the 16-bit kernels exist twice (namespace cw_bf16 / cw_f16, see kernels.h); the f32 parity kernels are
// identical in both builds
#define KD(c, fn, ...) ((c)->f16 ? cw_f16::fn(__VA_ARGS__) : cw_bf16::fn(__VA_ARGS__))

// launches of one decoder layer, as decode_step issues them (cw_time_decode_stage / cw_decode_stage_name)
#define CW_MAX_DEC_STAGES 24
enum DecStageKind { DST_OTHER = 0, DST_QKV_SELF, DST_QKV, DST_SELF_ATTN, DST_O_PROJ, DST_STACK, DST_STACK_CROSS, DST_CROSS_Q, DST_CROSS_ATTN,
                    DST_CROSS_O, DST_FC1, DST_FC2, DST_MLP_PAIR, DST_MLP_CHAIN, DST_N };
static const char* const kDecStageName[DST_N] = {
    "other (A/B path)", "LayerNorm + q/k/v projection + self-attention (qkv_self_kernel)", "LayerNorm + q/k/v projection + cache append",
    "self-attention", "self-attention out-projection", "fused out-projection + cross-query stage (gemv_stack_kernel)",
    "fused stage + cross-attention (dec_layer_a_kernel)", "LayerNorm + cross-attention query projection", "cross-attention",
    "cross-attention out-projection (combines the key-split partials)", "LayerNorm + fc1 + GELU", "fc2", "fc1 + fc2 (mlp_pair_kernel)",
    "LayerNorm + fc1 + GELU + fc2 in one launch (mlp_chain_kernel)"};

struct LayerW {
    void* wqkv = nullptr; float* bqkv = nullptr;
    void *wqkv8 = nullptr, *w18 = nullptr, *w28 = nullptr, *wkv_c8 = nullptr;   // e4m3 copies (option encoder_gemm_fp8) ...
    float *sqkv = nullptr, *s1 = nullptr, *s2 = nullptr, *skv_c = nullptr;        // ... and their per-output-row scales
    void* wo = nullptr; float* bo = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr;
    void* wq_c = nullptr; float* bq_c = nullptr;
    void* wkv_c = nullptr; float* bkv_c = nullptr;
    void* wo_c = nullptr; float* bo_c = nullptr;
    float *lnc_g = nullptr, *lnc_b = nullptr;
    void* w1 = nullptr; float* b1 = nullptr;
    void* w2 = nullptr; float* b2 = nullptr;
    float *ln2_g = nullptr, *ln2_b = nullptr;
    void *ck = nullptr, *cv = nullptr;   // cross K/V cache   [Bm][H][1500][64]
    void *ck8 = nullptr, *cv8 = nullptr; // opt-in fp8 (e4m3) copy of it, one byte per element
    float* kvs = nullptr;                // [Bm][H][2] dequantisation scales (K, V)
    void *sk = nullptr, *sv = nullptr;   // self  K/V cache   [Bm][H][448][64]
    // six-launch decoder layer (decfuse.hip), 16-bit engines: row-stacked weights [W'q_c ; W'q_c Wo ; Wo] and
    // [W'1 ; W'1 Wo_c ; Wo_c] -- wq_c / wo / w1 / wo_c above point into them -- and the per-column constants
    void *ws3 = nullptr, *ws5 = nullptr;
    float *qa_bias = nullptr, *q_wsum = nullptr;     // [D]  W'q_c bo ;  W'q_c 1
    float *u1_bias = nullptr, *u1_wsum = nullptr;    // [F]  W'1 bo_c ;  W'1 1
    float* qkv_wsum = nullptr;                       // [3D] W'qkv 1 (17..64-row path: LayerNorm through its linearity)
};

struct cw_ctx {
    cw_model_desc d;
    bool bf16 = false;         // 16-bit engine (bfloat16 or, with f16, IEEE binary16 -- the reference's GPU dtype)
    bool f16 = false;
    int device = 0;
    int Bm = 0, S_pad = 0, Vpad = 0;
    size_t esz = 4;
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr;          // side stream of the decode step: run-ahead prefetch of the next layer's streams (CW_PREFETCH)
    hipEvent_t ev_pf[2] = {nullptr, nullptr};
    int prefetch = 0;                   // 0 off; n > 0: blocks of the prefetch launch
    unsigned int* d_pf_sink = nullptr;
    std::vector<void*> allocs;
    char err[512] = "";
    bool err_set = false;      // a specific message is pending (set by fail(), cleared by cw_last_error)
    std::vector<int> align_layers, align_heads;
    // bf16 decode engine: projections fed by a LayerNorm keep their f32 checkpoint values on the device until every
    // tensor is in, then the LayerNorm's gamma / beta are folded into them (gemm.hip: fold_layernorm_kernel)
    struct Fold { float* stage; int N, K; void* w_dst; size_t w_off; float* b_dst; size_t b_off; float scale; int layer, which; };
    std::vector<Fold> folds;
    struct OutStage { float* stage; int layer, which; };   // f32 copies of the decoder out-projections (which: 0 self, 1 cross)
    std::vector<OutStage> out_stages;
    bool fuse6_ready = false;       // product matrices of the fused decoder stages are in place
    bool rows_ln_ready = false;     // row sums of the LayerNorm-folded q/k/v, cross-q and fc1 weights are in place (gemv_rows_kernel)
    bool rows_ln_enabled = false;   // CW_ROWS_LN=1 (A/B, measured slower): 17..64 rows without the preparation launch in front of every GEMV
    bool rows_hilo = true;          // CW_NO_ROWS_HILO=1: single 16-bit copy of the residual rows (A/B)
    bool mid16 = true;              // fc1 hands gelu(.) to fc2 in 16 bits (CW_NO_MID16=1: f32; A/B)
    bool dtw_block = false;         // CW_DTW_BLOCK=1: round-1 block-per-sequence DTW (A/B)
    bool stack_center = true;       // fused out-projection / cross-query stage rounds x - mean(x) (CW_NO_STACK_CENTER=1: x itself, round-3 behaviour; A/B)
    int skinny_mode = 0;            // 17..64 rows (skinny.hip / attention.hip: attn_cross_full_kernel); option "skinny" / CW_SKINNY=n:
                                    //   0 (default) round-3 path; 1 greedy rows: cross-attention query as a K-split skinny GEMM whose
                                    //   planes the one-block-per-(row, head) cross-attention finishes, which also writes the
                                    //   out-projection's rows (no LayerNorm preparation, no finish, no combine launch); 2 = 1 + q/k/v and
                                    //   fc1 through planes + finish launch.  Both measured slower in the step (profiles/r04_b64_*_rejected_*): A/B only
    float* d_planes = nullptr;      // [S][rows][N] K-split partial products of the LayerNorm projections (skinny.hip)
    size_t planes_cap = 0;          // floats
    float* d_sk_stats = nullptr;    // [16][64][2] slice statistics of the rows (skinny.hip)
    float* d_rstats = nullptr;      // [max(D, F) / 16][64][2] per-block LayerNorm partial sums of the 17..64-row producers
    bool fuse6_enabled = true;      // CW_NO_FUSE6=1: eight launches per layer (A/B)
    bool fuse_mlp = false;          // CW_FUSE_MLP=1: also fuse cross out-projection + fc1 (six launches; measured slower, A/B)
    bool mlp_pair_fence = false;    // CW_MLP_PAIR_FENCE=1: round-3 hand-over (agent-scope acquire fence behind the group barrier)
    bool mlp_pair = false;          // CW_MLP_PAIR=1: fc1 + fc2 in one launch with an in-kernel group barrier (A/B: 23 us against 12.8 for two launches)
    unsigned int* d_bar = nullptr; int* d_err = nullptr;   // group barriers of mlp_pair_kernel; "a block gave up waiting" flag
    int handoff_fallbacks = 0;      // times this context left the in-launch hand-offs for the launch-per-stage path because a wait gave up
    int handoff_resumes = 0;        // ... of which the decode call resumed at the failed position instead of starting over
    long long forwards_since_reset = 0;   // upper bound of the decoder forwards since the granule epoch last started at 1 (epoch_hygiene)
    int fail_pos = -1;              // test hook (option "handoff_fail_pos"): qkv_self_kernel gives up at this decoder position
    // persistent decoder-layer kernel (declayer.hip), rows <= 8: granule buffers, the epoch counter their tags carry, CU count
    bool declayer = false;          // CW_DECLAYER=1: stage A of declayer.hip (fused stage + cross-attention in one persistent launch; bit-identical,
                                    // measured SLOWER: 22-24 us against 18.4 for the two launches -- A/B and differential test only)
    bool qkv_self = true;           // q/k/v projection + self-attention in one launch (declayer.hip: qkv_self_kernel); CW_NO_QKV_SELF=1: two launches
    unsigned long long *d_gq = nullptr, *d_gps = nullptr, *d_gq2 = nullptr, *d_gkv = nullptr, *d_gflag = nullptr;
    bool mlp_chain = false;         // CW_MLP_CHAIN=1: fc1 + fc2 in one launch (declayer.hip: mlp_chain_kernel; flags instead of a kernel boundary);
                                    // bit-identical, measured SLOWER: 13.1 us against 5.7 + 4.9 for the two launches (profiles/r05_mlp_chain_phases.txt)
    unsigned int* d_epoch = nullptr;
    int n_cu = 0;
    int stack_nt3 = 0, stack_nt5 = 0;   // column tiles per block of the two stacked GEMVs (0 = launcher's choice; CW_STACK_NT3/5)
    bool ln_folded = false;         // decoder LN-GEMVs run plain normalisation (affine part is inside W / bias)
    bool fold_enabled = true;       // CW_NO_LN_FOLD=1: keep gamma / beta in the kernels
    std::set<std::string> loaded;   // HF tensor names received through cw_load_tensor
    bool weights_ok = false;        // every tensor of the geometry has been loaded (checked once, see cw_check_weights)
    bool wpacked = false;           // the decoder's GEMV matrices are in fragment-major order (gemm.hip: wfrag_pack_kernel)
    bool wpack_enabled = true;      // CW_NO_WPACK=1: keep them row-major (A/B)
    void* embed_pk = nullptr;       // fragment-major copy of the tied embedding for the logits GEMV (the row-major one serves the token lookup)

    // weights
    void *conv1_w = nullptr, *conv2_w = nullptr, *embed = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *dec_pos = nullptr;
    float *enc_ln_g = nullptr, *enc_ln_b = nullptr, *dec_ln_g = nullptr, *dec_ln_b = nullptr;
    std::vector<LayerW> enc, dec;

    // front end
    MelTables mel{};
    float *d_pcm = nullptr, *d_logspec = nullptr, *d_feats_hf = nullptr;
    unsigned int* d_gmax = nullptr;
    void* d_feats_tm = nullptr;
    std::vector<int> n_frames_items;

    // encoder workspace
    void *c1 = nullptr, *h = nullptr, *qb = nullptr, *kb = nullptr, *vb = nullptr, *ao = nullptr, *mid = nullptr,
         *enc_out = nullptr;
    float* x = nullptr;
    int *d_row_off = nullptr, *d_row_valid = nullptr, *d_row_off2 = nullptr, *d_row_valid2 = nullptr;
    int nb_encoded = 0;

    // decoder state
    float *dx = nullptr, *dxn = nullptr, *dq = nullptr, *dattn = nullptr, *dmid = nullptr, *dlogits = nullptr;
    float *dx1 = nullptr, *dx2c = nullptr, *d_qa = nullptr, *d_qb = nullptr, *d_u1 = nullptr, *d_pstats = nullptr;   // fused decoder stages (decfuse.hip)
    float *d_cvec = nullptr, *d_ostats = nullptr;             // 33..64 rows: row centres and per-(column pair, row) LayerNorm partial sums of the column-owning out-projection
    bool own_cols = true;                                     // CW_NO_OWN_COLS=1: K-split out-projection + LayerNorm preparation launch (A/B)
    void *d_xfrag = nullptr, *d_xfrag2 = nullptr;             // bf16 [64][5120] fragment-major activations of the 17..64-row GEMV path
    int *d_ids = nullptr, *d_forced = nullptr, *d_argmax = nullptr, *d_last_ts = nullptr, *d_finished = nullptr,
        *d_nunf = nullptr, *d_align_slot = nullptr;
    unsigned char* d_mask = nullptr;
    float* d_lp_sum = nullptr; int* d_lp_cnt = nullptr;      // per-row sum / count of chosen-token log-probabilities (score_tokens)
    bool score_tokens = false;
    float logprob_thr = NAN, no_speech_thr = NAN;             // cw_set_thresholds (NaN: unset)
    void* d_sample_part = nullptr;            // [Bm][16] 32-byte slice records of the two-stage sampler
    float* d_align = nullptr;
    float *d_part_o = nullptr, *d_part_ml = nullptr, *d_align_ml = nullptr;   // split cross-attention partials
    bool align_unnormalized = false;
    int* h_nunf = nullptr;  // pinned
    int *d_pos = nullptr, *d_cfg = nullptr;   // per-row decoder input position; [n_prompt, min_new, max_length, use_forced]
    hipGraphExec_t step_graph[130] = {};      // captured decode step (layers + logits + sampling) per batch size; + 65: the short-history form (hist_short)
    bool hist_short = false;                  // every row attends over <= 64 self-attention keys in the forward being launched (host knows the position)
    bool short_hist_enabled = true;           // CW_NO_SHORT_HIST=1: always request 128 history rows (A/B)
    bool use_graph = true;
    bool fuse_rows = true;                    // fused out-projection / cross-query stage at 17..64 greedy rows (CW_NO_FUSE_ROWS=1: off)
    bool fuse_beam = true;                    // ... and under beam search over the 16-bit cache (round 6; CW_NO_FUSE_BEAM=1: the twelve launches)
    bool fuse_rows8 = true;                   // ... over the e4m3 cache too (CW_NO_FUSE_ROWS8=1: that mode keeps its twelve launches)
    cw_gen_cfg gen{};
    bool gen_set = false;
    bool kv8 = false;                    // cross-attention reads the fp8 cache (cw_set_option "cross_kv_fp8")
    bool enc8 = false;                   // encoder linear layers + cross-K/V projection as e4m3 GEMMs ("encoder_gemm_fp8")
    int enc8_mask = 0;                   // which of them: 1 q/k/v, 2 fc1, 4 fc2, 8 cross-K/V projection (option value 1 = all, 16 + mask = subset)
    bool enc8_stale = false;             // a tensor was (re)loaded after the e4m3 copies were made
    void *h8 = nullptr, *mid8 = nullptr; float *sa8 = nullptr, *smid8 = nullptr;   // e4m3 activations and their row scales
    // beam search (cw_beam_*): rows = items x beams; self-attention keys are found through the ancestry table
    int beam_K = 0, beam_items = 0, beam_n_prompt = 0;
    int beam_pos = 0, beam_max_len = 0;   // position of the last decoder input / length limit (cw_beam_step refuses to run past it)
    int *d_anc = nullptr, *d_anc_tmp = nullptr, *d_ids_tmp = nullptr, *d_parent = nullptr, *d_tok = nullptr, *d_cand_id = nullptr,
        *d_rowmap = nullptr;
    float *d_cand_val = nullptr, *d_align_g = nullptr, *d_topk_scratch = nullptr;
    const float* align_cur = nullptr;     // alignment rows the timestamp stage reads (d_align, or the beam-gathered copy)
    float* logits_capture = nullptr;
    int logits_capture_steps = 0;
    int last_L = 0, last_nb = 0;

    // timestamps workspace
    float *d_mean = nullptr, *d_std = nullptr, *d_mat = nullptr, *d_skew = nullptr;
    size_t skew_cap = 0;
    unsigned char* d_trace = nullptr;
    int *d_first_col = nullptr, *d_path_text = nullptr, *d_path_time = nullptr, *d_path_len = nullptr,
        *d_ncols = nullptr;

    double* d_pause = nullptr;   // persistent scratch for cw_adjust_pauses: [4][pause_cap]
    int pause_cap = 0;

    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_step[2] = {nullptr, nullptr};
    // cw_time_decode_stage: one launch (stage_sel) of one layer (layer_sel) of decode_step; -1 = everything (the step itself)
    int stage_sel = -1, layer_sel = -1, stage_count = 0;
    int stage_kind[CW_MAX_DEC_STAGES] = {};
    int stage_launches[CW_MAX_DEC_STAGES] = {};   // kernel launches behind each stage (2 where gemv_prep_kernel precedes gemv_mt_kernel)
    float stage_ms[CW_N_STAGES] = {};
    int stage_calls[CW_N_STAGES] = {};
};

static int fail(cw_ctx* c, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_err, 512, fmt, ap);
    va_end(ap);
    if (c) c->err_set = true;
    return code;
}
// outer frames keep the innermost message and only add one when the callee reported a bare code
static int fail_ctx(cw_ctx* c, int code, const char* call, const char* file, int line) {
    if (c && c->err_set) return code;
    return fail(c, code, "%s -> %d (%s:%d)", call, code, file, line);
}

#define HIPCHK(c, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return fail(c, CW_ERR_HIP, "%s failed: %s (%s:%d)", #call,            \
                                          hipGetErrorString(e_), __FILE__, __LINE__);               \
    } while (0)
#define CWCHK(c, call)                                                          \
    do {                                                                        \
        int r_ = (call);                                                        \
        if (r_ != CW_OK) return fail_ctx(c, r_, #call, __FILE__, __LINE__); \
    } while (0)
#define KCHK(c) HIPCHK(c, hipGetLastError())

template <typename P>
static int dmalloc(cw_ctx* c, P** p, size_t bytes, bool zero = true) {
    void* q = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(c, CW_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) hipMemset(q, 0, bytes);
    c->allocs.push_back(q);
    *p = (P*)q;
    return CW_OK;
}

// host f32 -> device T (optionally scaled), at element offset `off` of dst
static int upload_T(cw_ctx* c, void* dst, size_t off, const float* src, size_t n, float scale = 1.0f) {
    if (c->bf16) {
        std::vector<bf16_t> tmp(n);
        if (c->f16) for (size_t i = 0; i < n; ++i) tmp[i] = cw_host_f32_to_f16(src[i] * scale);
        else for (size_t i = 0; i < n; ++i) tmp[i] = cw_host_f32_to_bf16(src[i] * scale);
        HIPCHK(c, hipMemcpy((bf16_t*)dst + off, tmp.data(), n * 2, hipMemcpyHostToDevice));
    } else if (scale != 1.0f) {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = src[i] * scale;
        HIPCHK(c, hipMemcpy((float*)dst + off, tmp.data(), n * 4, hipMemcpyHostToDevice));
    } else {
        HIPCHK(c, hipMemcpy((float*)dst + off, src, n * 4, hipMemcpyHostToDevice));
    }
    return CW_OK;
}
static int upload_f32(cw_ctx* c, float* dst, size_t off, const float* src, size_t n, float scale = 1.0f) {
    if (scale != 1.0f) {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) tmp[i] = src[i] * scale;
        HIPCHK(c, hipMemcpy(dst + off, tmp.data(), n * 4, hipMemcpyHostToDevice));
    } else {
        HIPCHK(c, hipMemcpy(dst + off, src, n * 4, hipMemcpyHostToDevice));
    }
    return CW_OK;
}
// device T -> host f32
static int download_T(cw_ctx* c, const void* src, size_t off, float* dst, size_t n) {
    if (c->bf16) {
        std::vector<bf16_t> tmp(n);
        HIPCHK(c, hipMemcpy(tmp.data(), (const bf16_t*)src + off, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) dst[i] = c->f16 ? cw_host_f16_to_f32(tmp[i]) : cw_host_bf16_to_f32(tmp[i]);
    } else {
        HIPCHK(c, hipMemcpy(dst, (const float*)src + off, n * 4, hipMemcpyDeviceToHost));
    }
    return CW_OK;
}

struct StageTimer {
    cw_ctx* c; int stage;
    StageTimer(cw_ctx* c_, int s) : c(c_), stage(s) { hipEventRecord(c->ev0, c->st); }
    void stop() {
        hipEventRecord(c->ev1, c->st);
        hipEventSynchronize(c->ev1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->stage_ms[stage] += ms;
        c->stage_calls[stage] += 1;
    }
};

#ifdef CW_EXPERIMENTS   // measured slower (profiles/r04_prefetch_side_stream_rejected.txt)
// Run-ahead prefetch of a decoder layer's HBM streams (weights + this batch's cross-attention K/V) into the Infinity Cache.
// A decode kernel's first bytes arrive 1.5-2 us after its launch when they come from HBM; on operands the previous launch left
// in the 256 MB Infinity Cache the same kernels run 0.7-1.5 us shorter each (tests/gpu_microbench.py).  The
// layer's 110 MB are therefore touched one layer ahead -- one 4-byte load per 64-byte sector, so the lines go HBM -> Infinity
// Cache and only 1/16 of the bytes travel on to the CU -- by a light launch on a side stream (a parallel branch of the captured
// step graph) that shares the CUs with the layer's own launches.
struct PrefetchRanges { const char* p[8]; unsigned long long n[8]; int count; };
template <int WIDE>
__global__ __launch_bounds__(256) void prefetch_kernel(PrefetchRanges r, unsigned int* sink) {
    unsigned int acc = 0;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
    for (int k = 0; k < r.count; ++k) {
        const char* base = r.p[k];
        if (WIDE) {                                               // every byte, 16 B per lane (A/B)
            const size_t chunks = r.n[k] >> 4;
            size_t s = tid;
            for (; s + 7 * nthr < chunks; s += 8 * nthr) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(base + ((s + u * nthr) << 4));
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
            }
            for (; s < chunks; s += nthr) acc ^= ((const uint4*)(base))[s].x;
        } else {
            const size_t sectors = r.n[k] >> 6;
            size_t s = tid;
            for (; s + 7 * nthr < sectors; s += 8 * nthr) {      // eight independent loads in flight per lane
                unsigned int v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const unsigned int*)(base + ((s + u * nthr) << 6));
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u];
            }
            for (; s < sectors; s += nthr) acc ^= *(const unsigned int*)(base + (s << 6));
        }
    }
    if (acc == 0x9e3779b9u && sink) *sink = acc;                  // keeps the loads alive; practically never taken
}


#endif

// Device buffers of a test / bench hook: everything allocated through `get` is freed when the scope ends, on every exit path
// (the hooks return through HIPCHK / CWCHK from many places; tools call them in loops).
struct DevScope {
    std::vector<void*> ptrs;
    template <typename P> hipError_t get(P** p, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs.push_back(q);
        *p = (P*)q;
        return e;
    }
    ~DevScope() { for (void* q : ptrs) hipFree(q); }
};

extern "C" {

int32_t cw_abi_version(void) { return 1; }

const char* cw_last_error(cw_ctx* ctx) {
    if (!ctx) return g_err;
    ctx->err_set = false;
    return ctx->err;
}

int32_t cw_sync(cw_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->st));
    return CW_OK;
}

static int create_impl(cw_ctx* c) {
    const cw_model_desc& d = c->d;
    const int D = d.d_model, H = d.n_heads, F = d.ffn_dim, V = d.vocab_size, Bm = d.max_batch;
    if (D != H * 64) return fail(c, CW_ERR_INVALID, "head_dim must be 64 (d_model=%d heads=%d)", D, H);
    if (D % 128 || F % 128 || d.n_mels % 64) return fail(c, CW_ERR_INVALID, "d_model/ffn must be multiples of 128, n_mels of 64");
    if (Bm < 1 || Bm > 64) return fail(c, CW_ERR_INVALID, "max_batch must be in 1..64");
    if (d.max_target_positions > 512) return fail(c, CW_ERR_INVALID, "max_target_positions > 512 unsupported");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamCreate(&c->st));
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_step[0], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_step[1], hipEventDisableTiming));
#ifdef CW_EXPERIMENTS   // side stream of the rejected run-ahead prefetch (CW_PREFETCH)
    HIPCHK(c, hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pf[0], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pf[1], hipEventDisableTiming));
#endif
    HIPCHK(c, hipHostMalloc((void**)&c->h_nunf, 2 * 1024 * sizeof(int), hipHostMallocDefault));   // [step]{running rows, hand-off failure word}
    if (d.dtype != CW_DTYPE_F32 && d.dtype != CW_DTYPE_BF16 && d.dtype != CW_DTYPE_F16) return fail(c, CW_ERR_INVALID, "unknown dtype %d", d.dtype);
    c->f16 = d.dtype == CW_DTYPE_F16;
    c->bf16 = d.dtype == CW_DTYPE_BF16 || c->f16;
    const cw_sw::Switches sw = cw_sw::read_switches();   // the environment as it is now: a context's switches are fixed at creation
#ifndef CW_EXPERIMENTS
    {
        const struct { const char* name; bool set; } rejected[] = {{"CW_ROWS_LN", sw.rows_ln}, {"CW_FUSE_MLP", sw.fuse_mlp}, {"CW_MLP_PAIR", sw.mlp_pair},
                                                                   {"CW_SKINNY", sw.skinny != 0}, {"CW_PREFETCH", sw.prefetch != 0},
                                                                   {"CW_DECLAYER", sw.declayer}, {"CW_MLP_CHAIN", sw.mlp_chain}};
        for (const auto& r : rejected)
            if (r.set)
                return fail(c, CW_ERR_INVALID, "%s selects a measured-and-rejected kernel variant that is not in this build: make EXTRA=-DCW_EXPERIMENTS", r.name);
    }
#endif
    c->use_graph = !sw.no_graph;
    c->fuse_rows = !sw.no_fuse_rows;
    c->fuse_rows8 = !sw.no_fuse_rows8;
    c->fuse_beam = !sw.no_fuse_beam;
    c->own_cols = !sw.no_own_cols;
    c->short_hist_enabled = !sw.no_short_hist;
    c->fold_enabled = !sw.no_ln_fold;
    c->fuse6_enabled = !sw.no_fuse6;
    c->rows_ln_enabled = sw.rows_ln;
    c->rows_hilo = !sw.no_rows_hilo;
    c->skinny_mode = sw.skinny;
    c->stack_center = !sw.no_stack_center;
    c->mid16 = !sw.no_mid16;
    c->dtw_block = sw.dtw_block;
    c->fuse_mlp = sw.fuse_mlp;
    c->wpack_enabled = !sw.no_wpack;
    c->mlp_pair = sw.mlp_pair;
    c->mlp_pair_fence = sw.mlp_pair_fence;
    c->declayer = sw.declayer;
    c->qkv_self = !sw.no_qkv_self;
    c->mlp_chain = sw.mlp_chain;
    c->stack_nt3 = sw.stack_nt3;
    c->stack_nt5 = sw.stack_nt5;
    c->prefetch = sw.prefetch;   // experiments builds only (refused above otherwise)
    c->esz = c->bf16 ? 2 : 4;
    c->Bm = Bm;
    c->S_pad = 1536;
    const size_t e = c->esz;
    const int TGT = d.max_target_positions;

    // ---- weights
    CWCHK(c, dmalloc(c, &c->conv1_w, (size_t)D * 3 * d.n_mels * e));
    CWCHK(c, dmalloc(c, &c->conv2_w, (size_t)D * 3 * D * e));
    CWCHK(c, dmalloc(c, &c->conv1_b, (size_t)D * 4));
    CWCHK(c, dmalloc(c, &c->conv2_b, (size_t)D * 4));
    CWCHK(c, dmalloc(c, &c->enc_pos, (size_t)CW_N_CTX * D * 4));
    CWCHK(c, dmalloc(c, &c->dec_pos, (size_t)TGT * D * 4));
    CWCHK(c, dmalloc(c, &c->embed, (size_t)V * D * e));
    CWCHK(c, dmalloc(c, &c->enc_ln_g, D * 4)); CWCHK(c, dmalloc(c, &c->enc_ln_b, D * 4));
    CWCHK(c, dmalloc(c, &c->dec_ln_g, D * 4)); CWCHK(c, dmalloc(c, &c->dec_ln_b, D * 4));
    c->enc.resize(d.enc_layers);
    c->dec.resize(d.dec_layers);
    auto alloc_common = [&](LayerW& L, bool stacked) -> int {
        CWCHK(c, dmalloc(c, &L.wqkv, (size_t)3 * D * D * e)); CWCHK(c, dmalloc(c, &L.bqkv, (size_t)3 * D * 4));
        if (!stacked) CWCHK(c, dmalloc(c, &L.wo, (size_t)D * D * e));
        CWCHK(c, dmalloc(c, &L.bo, D * 4));
        CWCHK(c, dmalloc(c, &L.ln1_g, D * 4)); CWCHK(c, dmalloc(c, &L.ln1_b, D * 4));
        if (!stacked) CWCHK(c, dmalloc(c, &L.w1, (size_t)F * D * e));
        CWCHK(c, dmalloc(c, &L.b1, F * 4));
        CWCHK(c, dmalloc(c, &L.w2, (size_t)D * F * e)); CWCHK(c, dmalloc(c, &L.b2, D * 4));
        CWCHK(c, dmalloc(c, &L.ln2_g, D * 4)); CWCHK(c, dmalloc(c, &L.ln2_b, D * 4));
        return CW_OK;
    };
    for (auto& L : c->enc) CWCHK(c, alloc_common(L, false));
    for (auto& L : c->dec) {
        const bool stacked = c->bf16;   // 16-bit engines: the four projections live inside the two row-stacked matrices
        CWCHK(c, alloc_common(L, stacked));
        if (stacked) {
            CWCHK(c, dmalloc(c, &L.ws3, (size_t)3 * D * D * e));
            CWCHK(c, dmalloc(c, &L.ws5, ((size_t)2 * F + D) * D * e));
            L.wq_c = L.ws3; L.wo = (char*)L.ws3 + (size_t)2 * D * D * e;
            L.w1 = L.ws5;   L.wo_c = (char*)L.ws5 + (size_t)2 * F * D * e;
            CWCHK(c, dmalloc(c, &L.qa_bias, D * 4)); CWCHK(c, dmalloc(c, &L.q_wsum, D * 4));
            CWCHK(c, dmalloc(c, &L.u1_bias, F * 4)); CWCHK(c, dmalloc(c, &L.u1_wsum, F * 4));
            CWCHK(c, dmalloc(c, &L.qkv_wsum, (size_t)3 * D * 4));
        } else {
            CWCHK(c, dmalloc(c, &L.wq_c, (size_t)D * D * e));
            CWCHK(c, dmalloc(c, &L.wo_c, (size_t)D * D * e));
        }
        CWCHK(c, dmalloc(c, &L.bq_c, D * 4));
        CWCHK(c, dmalloc(c, &L.wkv_c, (size_t)2 * D * D * e)); CWCHK(c, dmalloc(c, &L.bkv_c, (size_t)2 * D * 4));
        CWCHK(c, dmalloc(c, &L.bo_c, D * 4));
        CWCHK(c, dmalloc(c, &L.lnc_g, D * 4)); CWCHK(c, dmalloc(c, &L.lnc_b, D * 4));
        CWCHK(c, dmalloc(c, &L.ck, (size_t)Bm * H * CW_N_CTX * 64 * e));
        CWCHK(c, dmalloc(c, &L.cv, (size_t)Bm * H * CW_N_CTX * 64 * e));
        CWCHK(c, dmalloc(c, &L.sk, (size_t)Bm * H * TGT * 64 * e));
        CWCHK(c, dmalloc(c, &L.sv, (size_t)Bm * H * TGT * 64 * e));
    }

    // ---- mel tables (host-computed: identical on every box)
    {
        std::vector<double> ct(400), sn(400), win(400);
        for (int i = 0; i < 400; ++i) {
            ct[i] = cos(2.0 * M_PI * i / 400.0);
            sn[i] = sin(2.0 * M_PI * i / 400.0);
            win[i] = (double)(float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 400.0));  // torch.hann_window is f32
        }
        double *dct, *dsn, *dwin;
        CWCHK(c, dmalloc(c, &dct, 400 * 8)); CWCHK(c, dmalloc(c, &dsn, 400 * 8)); CWCHK(c, dmalloc(c, &dwin, 400 * 8));
        HIPCHK(c, hipMemcpy(dct, ct.data(), 400 * 8, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dsn, sn.data(), 400 * 8, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dwin, win.data(), 400 * 8, hipMemcpyHostToDevice));
        // slaney mel filterbank, TF/audio_utils.py:448-560, 700-720 (float64 maths, cast to f32 at use)
        const int nm = d.n_mels, nb = 201;
        auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; };
        auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
        std::vector<double> ff(nm + 2);
        const double m_lo = hz2mel(0.0), m_hi = hz2mel(8000.0);
        for (int i = 0; i < nm + 2; ++i) {
            double step = (m_hi - m_lo) / (nm + 1);          // np.linspace
            double mv = (i == nm + 1) ? m_hi : m_lo + step * i;
            ff[i] = mel2hz(mv);
        }
        std::vector<float> fb((size_t)nb * nm);
        for (int k = 0; k < nb; ++k) {
            double fk = (k == nb - 1) ? 8000.0 : (8000.0 / (nb - 1)) * k;
            for (int m = 0; m < nm; ++m) {
                double down = -(ff[m] - fk) / (ff[m + 1] - ff[m]);
                double up = (ff[m + 2] - fk) / (ff[m + 2] - ff[m + 1]);
                double v = fmax(0.0, fmin(down, up));
                v *= 2.0 / (ff[m + 2] - ff[m]);
                fb[(size_t)k * nm + m] = (float)v;
            }
        }
        float* dfb;
        CWCHK(c, dmalloc(c, &dfb, fb.size() * 4));
        HIPCHK(c, hipMemcpy(dfb, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
        c->mel.cos_t = dct; c->mel.sin_t = dsn; c->mel.window = dwin; c->mel.filters = dfb;
        // non-zero bin range of every mel filter (mel.hip: mel_mfma_kernel walks only those)
        {
            std::vector<int> lo(nm, nb), hi(nm, -1);
            for (int m = 0; m < nm; ++m)
                for (int k = 0; k < nb; ++k)
                    if (fb[(size_t)k * nm + m] != 0.f) { if (k < lo[m]) lo[m] = k; if (k > hi[m]) hi[m] = k; }
            int *dlo, *dhi;
            CWCHK(c, dmalloc(c, &dlo, nm * 4)); CWCHK(c, dmalloc(c, &dhi, nm * 4));
            HIPCHK(c, hipMemcpy(dlo, lo.data(), nm * 4, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(dhi, hi.data(), nm * 4, hipMemcpyHostToDevice));
            c->mel.fb_lo = dlo; c->mel.fb_hi = dhi;
        }
    }
    CWCHK(c, dmalloc(c, &c->d_pcm, (size_t)Bm * CW_N_SAMPLES * 4));
    CWCHK(c, dmalloc(c, &c->d_logspec, (size_t)Bm * CW_N_FRAMES * d.n_mels * 4));
    CWCHK(c, dmalloc(c, &c->d_gmax, Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_feats_tm, (size_t)Bm * CW_N_FRAMES * d.n_mels * e));
    c->n_frames_items.assign(Bm, CW_N_FRAMES);

    // ---- encoder workspace
    const size_t MR = (size_t)Bm * CW_N_CTX;
    CWCHK(c, dmalloc(c, &c->c1, (size_t)Bm * CW_N_FRAMES * D * e));
    CWCHK(c, dmalloc(c, &c->x, MR * D * 4));
    CWCHK(c, dmalloc(c, &c->h, MR * D * e));
    CWCHK(c, dmalloc(c, &c->ao, MR * D * e));
    CWCHK(c, dmalloc(c, &c->enc_out, MR * D * e));
    CWCHK(c, dmalloc(c, &c->mid, MR * F * e));
    CWCHK(c, dmalloc(c, &c->qb, (size_t)Bm * H * c->S_pad * 64 * e));
    CWCHK(c, dmalloc(c, &c->kb, (size_t)Bm * H * c->S_pad * 64 * e));
    CWCHK(c, dmalloc(c, &c->vb, (size_t)Bm * H * c->S_pad * 64 * e));
    CWCHK(c, dmalloc(c, &c->d_row_off, Bm * 4)); CWCHK(c, dmalloc(c, &c->d_row_valid, Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_row_off2, Bm * 4)); CWCHK(c, dmalloc(c, &c->d_row_valid2, Bm * 4));
    {
        std::vector<int> off2(Bm), val2(Bm, CW_N_FRAMES);
        for (int i = 0; i < Bm; ++i) off2[i] = i * CW_N_FRAMES;
        HIPCHK(c, hipMemcpy(c->d_row_off2, off2.data(), Bm * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_row_valid2, val2.data(), Bm * 4, hipMemcpyHostToDevice));
    }

    // ---- decoder state
    CWCHK(c, dmalloc(c, &c->dx, (size_t)Bm * D * 4)); CWCHK(c, dmalloc(c, &c->dxn, (size_t)Bm * D * 4));
    CWCHK(c, dmalloc(c, &c->dq, (size_t)Bm * D * 4)); CWCHK(c, dmalloc(c, &c->dattn, (size_t)Bm * D * 4));
    CWCHK(c, dmalloc(c, &c->dmid, (size_t)Bm * F * 4));
    CWCHK(c, dmalloc(c, &c->dx1, (size_t)Bm * D * 4)); CWCHK(c, dmalloc(c, &c->dx2c, (size_t)Bm * D * 4));
    CWCHK(c, dmalloc(c, &c->d_qa, (size_t)Bm * D * 4)); CWCHK(c, dmalloc(c, &c->d_qb, (size_t)Bm * D * 4));
    CWCHK(c, dmalloc(c, &c->d_u1, (size_t)Bm * F * 4)); CWCHK(c, dmalloc(c, &c->d_pstats, (size_t)128 * 64 * 2 * 4));   // [group of 16 rows 4][tile <= 128][16][2]
    CWCHK(c, dmalloc(c, &c->d_rstats, (size_t)((D > F ? D : F) / 16 + 1) * 64 * 2 * 4));
    CWCHK(c, dmalloc(c, &c->d_bar, 64 * 4));
    {   // declayer.hip: granules are valid by tag only (never cleared); epoch 0 is never used
        CWCHK(c, dmalloc(c, &c->d_gq, (size_t)2 * 16 * D * 8)); CWCHK(c, dmalloc(c, &c->d_gps, (size_t)(D / 16) * 16 * 2 * 8));
        CWCHK(c, dmalloc(c, &c->d_gq2, (size_t)16 * D * 8)); CWCHK(c, dmalloc(c, &c->d_gkv, (size_t)2 * 16 * (D / 2) * 8));
        CWCHK(c, dmalloc(c, &c->d_gflag, (size_t)512 * 8));
        CWCHK(c, dmalloc(c, &c->d_epoch, 4));
        const unsigned int one = 1;
        HIPCHK(c, hipMemcpy(c->d_epoch, &one, 4, hipMemcpyHostToDevice));
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
        c->n_cu = prop.multiProcessorCount;
    }
    CWCHK(c, dmalloc(c, &c->d_xfrag, (size_t)64 * 5120 * 2));
#ifdef CW_EXPERIMENTS   // skinny.hip planes (rejected A/B, CW_SKINNY): four K slices of the widest LayerNorm projection for 64 rows (5.2 MB at large-v3)
    if (Bm > 16) {
        c->planes_cap = (size_t)4 * 64 * (3 * D > F ? 3 * D : F);
        CWCHK(c, dmalloc(c, &c->d_planes, c->planes_cap * 4, false));
        CWCHK(c, dmalloc(c, &c->d_sk_stats, (size_t)16 * 64 * 2 * 4));
    }
#endif
    CWCHK(c, dmalloc(c, &c->d_xfrag2, (size_t)64 * 5120 * 2));
    CWCHK(c, dmalloc(c, &c->d_cvec, 64 * 4)); CWCHK(c, dmalloc(c, &c->d_ostats, (size_t)96 * 64 * 2 * 4));   // column-owning out-projection of 33..64 rows
    c->Vpad = (V + 3) & ~3;
    CWCHK(c, dmalloc(c, &c->dlogits, (size_t)Bm * c->Vpad * 4));
    CWCHK(c, dmalloc(c, &c->d_ids, (size_t)Bm * TGT * 4)); CWCHK(c, dmalloc(c, &c->d_forced, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_argmax, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_last_ts, Bm * 4)); CWCHK(c, dmalloc(c, &c->d_finished, Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_nunf, 8));   // {rows still running, hand-off failure word}: one 8-byte copy per step brings both to the host
    c->d_err = c->d_nunf + 1;
#ifdef CW_EXPERIMENTS
    CWCHK(c, dmalloc(c, &c->d_pf_sink, 4));
#endif
    CWCHK(c, dmalloc(c, &c->d_pos, 64 * 4)); CWCHK(c, dmalloc(c, &c->d_cfg, 4 * 4));
    CWCHK(c, dmalloc(c, &c->d_mask, (size_t)V + 16));
    CWCHK(c, dmalloc(c, &c->d_sample_part, (size_t)Bm * 16 * 32));
    CWCHK(c, dmalloc(c, &c->d_lp_sum, (size_t)Bm * 4)); CWCHK(c, dmalloc(c, &c->d_lp_cnt, (size_t)Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_align_slot, (size_t)d.dec_layers * H * 4));
    {
        std::vector<int> slot((size_t)d.dec_layers * H, -1);
        for (int a = 0; a < d.n_align; ++a) {
            int l = c->align_layers[a], hh = c->align_heads[a];
            if (l < 0 || l >= d.dec_layers || hh < 0 || hh >= H) return fail(c, CW_ERR_INVALID, "bad alignment head (%d,%d)", l, hh);
            slot[(size_t)l * H + hh] = a;
        }
        HIPCHK(c, hipMemcpy(c->d_align_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice));
    }
    const int Ha = d.n_align > 0 ? d.n_align : 1;
    CWCHK(c, dmalloc(c, &c->d_align, (size_t)Bm * Ha * TGT * CW_N_CTX * 4));
    CWCHK(c, dmalloc(c, &c->d_part_o, (size_t)ATT_NS * Bm * D * 4));
    CWCHK(c, dmalloc(c, &c->d_part_ml, (size_t)Bm * H * ATT_NS * 2 * 4));
    CWCHK(c, dmalloc(c, &c->d_align_ml, (size_t)Bm * Ha * TGT * ATT_NS * 2 * 4));

    // ---- timestamps workspace
    CWCHK(c, dmalloc(c, &c->d_mean, (size_t)Bm * Ha * CW_N_CTX * 4));
    CWCHK(c, dmalloc(c, &c->d_std, (size_t)Bm * Ha * CW_N_CTX * 4));
    CWCHK(c, dmalloc(c, &c->d_mat, (size_t)Bm * TGT * CW_N_CTX * 4));
    CWCHK(c, dmalloc(c, &c->d_trace, (size_t)Bm * TGT * CW_N_CTX));
    CWCHK(c, dmalloc(c, &c->d_first_col, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_path_text, (size_t)Bm * (TGT + CW_N_CTX + 2) * 4));
    CWCHK(c, dmalloc(c, &c->d_path_time, (size_t)Bm * (TGT + CW_N_CTX + 2) * 4));
    CWCHK(c, dmalloc(c, &c->d_path_len, Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_ncols, Bm * 4));
    HIPCHK(c, hipDeviceSynchronize());
    return CW_OK;
}

cw_ctx* cw_create(const cw_model_desc* desc, int32_t device) {
    if (!desc) { fail(nullptr, CW_ERR_INVALID, "null desc"); return nullptr; }
    cw_ctx* c = new cw_ctx();
    c->d = *desc;
    c->device = device;
    c->align_layers.assign(desc->align_layers, desc->align_layers + desc->n_align);
    c->align_heads.assign(desc->align_heads, desc->align_heads + desc->n_align);
    c->d.align_layers = c->align_layers.data();
    c->d.align_heads = c->align_heads.data();
    int r = create_impl(c);
    if (r != CW_OK) {
        snprintf(g_err, sizeof(g_err), "%s", c->err);
        cw_destroy(c);
        return nullptr;
    }
    return c;
}

void cw_destroy(cw_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->st) hipStreamSynchronize(c->st);
    for (auto& ge : c->step_graph) if (ge) hipGraphExecDestroy(ge);
    for (auto& f : c->folds) hipFree(f.stage);
    for (auto& o : c->out_stages) hipFree(o.stage);
    for (void* p : c->allocs) hipFree(p);
    if (c->h_nunf) hipHostFree(c->h_nunf);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (auto& e : c->ev_step) if (e) hipEventDestroy(e);
    for (auto& e : c->ev_pf) if (e) hipEventDestroy(e);
    if (c->st2) hipStreamDestroy(c->st2);
    if (c->st) hipStreamDestroy(c->st);
    delete c;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static int load_tensor_impl(cw_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim);
static int apply_folds(cw_ctx* c);
static int pack_decoder_weights(cw_ctx* c);

int32_t cw_load_tensor(cw_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!name || !data || !shape) return fail(c, CW_ERR_INVALID, "cw_load_tensor: null argument");
    const int r = load_tensor_impl(c, name, data, shape, ndim);
    if (r == CW_OK) { c->loaded.insert(name); c->weights_ok = false; c->enc8_stale = true; }   // e4m3 copies follow the weights
    return r;
}

// Every tensor of WhisperForConditionalGeneration.state_dict() for this geometry (k_proj has no bias,
// modeling_whisper.py:279; proj_out is tied).  Device buffers start zero-filled, so a checkpoint with a missing shard
// would otherwise run and return garbage: the first cw_encode refuses to start until the list is complete.
int32_t cw_check_weights(cw_ctx* c) {
    if (c->weights_ok) return CW_OK;
    std::vector<std::string> want = {
        "model.encoder.conv1.weight", "model.encoder.conv1.bias", "model.encoder.conv2.weight", "model.encoder.conv2.bias",
        "model.encoder.embed_positions.weight", "model.encoder.layer_norm.weight", "model.encoder.layer_norm.bias",
        "model.decoder.embed_tokens.weight", "model.decoder.embed_positions.weight", "model.decoder.layer_norm.weight",
        "model.decoder.layer_norm.bias"};
    auto attn = [&](const std::string& p) {
        for (const char* t : {".q_proj.weight", ".q_proj.bias", ".k_proj.weight", ".v_proj.weight", ".v_proj.bias",
                              ".out_proj.weight", ".out_proj.bias"}) want.push_back(p + t);
    };
    auto pair = [&](const std::string& p) { want.push_back(p + ".weight"); want.push_back(p + ".bias"); };
    for (int stack = 0; stack < 2; ++stack) {
        const int nl = stack ? c->d.dec_layers : c->d.enc_layers;
        for (int l = 0; l < nl; ++l) {
            const std::string p = std::string(stack ? "model.decoder.layers." : "model.encoder.layers.") + std::to_string(l);
            attn(p + ".self_attn"); pair(p + ".self_attn_layer_norm");
            if (stack) { attn(p + ".encoder_attn"); pair(p + ".encoder_attn_layer_norm"); }
            pair(p + ".fc1"); pair(p + ".fc2"); pair(p + ".final_layer_norm");
        }
    }
    std::string missing; int n_missing = 0;
    for (const auto& w : want)
        if (!c->loaded.count(w)) { if (n_missing < 4) missing += (n_missing ? ", " : "") + w; ++n_missing; }
    if (n_missing) return fail(c, CW_ERR_STATE, "%d of %zu weight tensors were never loaded (e.g. %s): incomplete checkpoint", n_missing, want.size(), missing.c_str());
    CWCHK(c, apply_folds(c));
    CWCHK(c, pack_decoder_weights(c));
    c->weights_ok = true;
    return CW_OK;
}

// which: 0 = self-attention LN (q/k/v), 1 = cross-attention LN (q), 2 = final LN (fc1)
static int stage_fold(cw_ctx* c, int layer, int which, const float* data, int N, int K, void* w_dst, size_t w_off, float* b_dst,
                      size_t b_off, float scale) {
    cw_ctx::Fold f{nullptr, N, K, w_dst, w_off, b_dst, b_off, scale, layer, which};
    HIPCHK(c, hipMalloc((void**)&f.stage, (size_t)N * K * 4));
    HIPCHK(c, hipMemcpy(f.stage, data, (size_t)N * K * 4, hipMemcpyHostToDevice));
    c->folds.push_back(f);
    return CW_OK;
}

static int apply_folds(cw_ctx* c) {
    for (auto& f : c->folds) {
        LayerW& L = c->dec[f.layer];
        const float* g = f.which == 0 ? L.ln1_g : (f.which == 1 ? L.lnc_g : L.ln2_g);
        const float* b = f.which == 0 ? L.ln1_b : (f.which == 1 ? L.lnc_b : L.ln2_b);
        CWCHK(c, KD(c, cw_launch_fold_layernorm, f.stage, f.N, f.K, g, b, f.scale, (bf16_t*)f.w_dst + f.w_off, f.b_dst + f.b_off, c->st));
    }
    // six-launch layer: product matrices W'q_c Wo and W'1 Wo_c (f32 from the checkpoint values, one 16-bit rounding), the
    // constants W'q_c bo / W'1 bo_c and the row sums of the folded 16-bit weights (what the MFMAs actually multiply by)
    const int D = c->d.d_model, F = c->d.ffn_dim;
    bool all = !c->folds.empty() && c->fuse6_enabled && F % D == 0 && D <= 1280 && D % 64 == 0 && F % 64 == 0 && c->d.n_heads <= 20;
    if (all) {
        for (int l = 0; l < c->d.dec_layers && all; ++l) {
            const cw_ctx::Fold *fq = nullptr, *f1 = nullptr;
            const float *so = nullptr, *sc = nullptr;
            for (auto& f : c->folds) if (f.layer == l) { if (f.which == 1) fq = &f; else if (f.which == 2) f1 = &f; }
            for (auto& o : c->out_stages) if (o.layer == l) { if (o.which == 0) so = o.stage; else sc = o.stage; }
            if (!fq || !f1 || !so || !sc) { all = false; break; }
            LayerW& L = c->dec[l];
            const size_t e = c->esz;
            CWCHK(c, KD(c, cw_launch_fold_product, fq->stage, L.lnc_g, fq->scale, so, D, D, D, (char*)L.ws3 + (size_t)D * D * e, c->st));
            CWCHK(c, KD(c, cw_launch_fold_rowvec, fq->stage, L.lnc_g, fq->scale, L.bo, L.wq_c, D, D, L.qa_bias, L.q_wsum, c->st));
            CWCHK(c, KD(c, cw_launch_fold_product, f1->stage, L.ln2_g, f1->scale, sc, F, D, D, (char*)L.ws5 + (size_t)F * D * e, c->st));
            CWCHK(c, KD(c, cw_launch_fold_rowvec, f1->stage, L.ln2_g, f1->scale, L.bo_c, L.w1, F, D, L.u1_bias, L.u1_wsum, c->st));
        }
    }
    // 17..64-row path: row sums of the folded 16-bit matrices (row-major here: packing comes after the folds)
    bool rows_ok = !c->folds.empty() && c->bf16 && D % 128 == 0 && F % 128 == 0 && F <= 5120;
    for (int l = 0; l < c->d.dec_layers && rows_ok; ++l) {
        LayerW& L = c->dec[l];
        if (!L.qkv_wsum || !L.q_wsum || !L.u1_wsum) { rows_ok = false; break; }
        CWCHK(c, KD(c, cw_launch_fold_rowvec, nullptr, nullptr, 1.0f, nullptr, L.wqkv, 3 * D, D, nullptr, L.qkv_wsum, c->st));
        if (!all) {
            CWCHK(c, KD(c, cw_launch_fold_rowvec, nullptr, nullptr, 1.0f, nullptr, L.wq_c, D, D, nullptr, L.q_wsum, c->st));
            CWCHK(c, KD(c, cw_launch_fold_rowvec, nullptr, nullptr, 1.0f, nullptr, L.w1, F, D, nullptr, L.u1_wsum, c->st));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    KCHK(c);
    c->fuse6_ready = all;
    c->rows_ln_ready = rows_ok;
    for (auto& f : c->folds) hipFree(f.stage);
    for (auto& o : c->out_stages) hipFree(o.stage);
    c->out_stages.clear();
    if (!c->folds.empty()) c->ln_folded = true;
    c->folds.clear();
    return CW_OK;
}

// 16-bit engines: the decoder's GEMV weight matrices go fragment-major once every tensor is in and folded (gemm.hip:
// wfrag_pack_kernel) -- in place through one scratch buffer -- and the tied embedding gets a packed copy for the logits GEMV.
// A tensor loaded afterwards would land row-major in a packed matrix: refused like a load after folding.
static int pack_decoder_weights(cw_ctx* c) {
    if (!c->bf16 || !c->wpack_enabled || c->wpacked) return CW_OK;
    const int D = c->d.d_model, F = c->d.ffn_dim, V = c->d.vocab_size;
    // packed weights are read by gemv2_bf16_kernel / gemv_mt_kernel / gemv_stack_kernel only (the first-generation GEMV reads
    // row-major): pack exactly when those kernels take every decoder shape -- K = d_model in one K slice (<= 1280), fc2's
    // K = ffn_dim in power-of-two slices of <= 1280 that are multiples of 128 (gemm.hip: gemv2_ok).  Wider geometries
    // (d_model up to 2048 is accepted by cw_create) keep row-major weights and the fallback kernel.
    int fks = 1;
    while (F / fks > 1280) fks *= 2;
    if (D % 128 || D > 1280 || F % 128 || F > 5120 || F % fks || (F / fks) % 128) return CW_OK;
    const size_t emb_elems = KD(c, cw_wfrag_elems, V, D);
    size_t tmp_elems = (size_t)(2 * F + D) * D;
    if ((size_t)3 * D * D > tmp_elems) tmp_elems = (size_t)3 * D * D;
    void* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, tmp_elems * 2));
    auto pack = [&](void* w, int N, int K) -> int {
        CWCHK(c, KD(c, cw_launch_wfrag_pack, w, N, K, tmp, c->st));
        HIPCHK(c, hipMemcpyAsync(w, tmp, (size_t)N * K * 2, hipMemcpyDeviceToDevice, c->st));
        return CW_OK;
    };
    int r = CW_OK;
    for (auto& L : c->dec) {
        if (r == CW_OK) r = pack(L.wqkv, 3 * D, D);
        if (r == CW_OK) r = pack(L.ws3, 3 * D, D);            // [W'q_c ; W'q_c Wo ; Wo]: wq_c / wo point into it
        if (r == CW_OK) r = pack(L.ws5, 2 * F + D, D);        // [W'1 ; W'1 Wo_c ; Wo_c]: w1 / wo_c point into it
        if (r == CW_OK) r = pack(L.w2, D, F);
    }
    if (r == CW_OK && !c->embed_pk) r = dmalloc(c, &c->embed_pk, emb_elems * 2, false);
    if (r == CW_OK) r = KD(c, cw_launch_wfrag_pack, c->embed, V, D, c->embed_pk, c->st);
    hipStreamSynchronize(c->st);
    hipFree(tmp);
    if (r != CW_OK) return r;
    KCHK(c);
    c->wpacked = true;
    return CW_OK;
}

static int load_tensor_impl(cw_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    const int D = c->d.d_model, F = c->d.ffn_dim, V = c->d.vocab_size, NM = c->d.n_mels;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    auto expect = [&](size_t want) -> int {
        if (n != want) return fail(c, CW_ERR_INVALID, "tensor %s: %zu elements, expected %zu", name, n, want);
        return CW_OK;
    };
    std::string s(name);
    if (s == "proj_out.weight") return CW_OK;  // tied to embed_tokens (modeling_whisper.py:965)
    if (s == "model.encoder.conv1.weight" || s == "model.encoder.conv2.weight") {
        const bool first = s == "model.encoder.conv1.weight";
        const int C = first ? NM : D;
        CWCHK(c, expect((size_t)D * C * 3));
        std::vector<float> re((size_t)D * 3 * C);  // [O][C][3] -> [O][3][C]
        for (int o = 0; o < D; ++o)
            for (int ci = 0; ci < C; ++ci)
                for (int k = 0; k < 3; ++k) re[((size_t)o * 3 + k) * C + ci] = data[((size_t)o * C + ci) * 3 + k];
        return upload_T(c, first ? c->conv1_w : c->conv2_w, 0, re.data(), re.size());
    }
    if (s == "model.encoder.conv1.bias") { CWCHK(c, expect(D)); return upload_f32(c, c->conv1_b, 0, data, n); }
    if (s == "model.encoder.conv2.bias") { CWCHK(c, expect(D)); return upload_f32(c, c->conv2_b, 0, data, n); }
    if (s == "model.encoder.embed_positions.weight") { CWCHK(c, expect((size_t)CW_N_CTX * D)); return upload_f32(c, c->enc_pos, 0, data, n); }
    if (s == "model.decoder.embed_positions.weight") { CWCHK(c, expect((size_t)c->d.max_target_positions * D)); return upload_f32(c, c->dec_pos, 0, data, n); }
    if (s == "model.decoder.embed_tokens.weight") {
        CWCHK(c, expect((size_t)V * D));
        if (c->wpacked) return fail(c, CW_ERR_STATE, "decoder weights were already packed for the decode GEMVs: create a new context to load another checkpoint");
        return upload_T(c, c->embed, 0, data, n);
    }
    if (s == "model.encoder.layer_norm.weight") { CWCHK(c, expect(D)); return upload_f32(c, c->enc_ln_g, 0, data, n); }
    if (s == "model.encoder.layer_norm.bias") { CWCHK(c, expect(D)); return upload_f32(c, c->enc_ln_b, 0, data, n); }
    if (s == "model.decoder.layer_norm.weight") { CWCHK(c, expect(D)); return upload_f32(c, c->dec_ln_g, 0, data, n); }
    if (s == "model.decoder.layer_norm.bias") { CWCHK(c, expect(D)); return upload_f32(c, c->dec_ln_b, 0, data, n); }

    int li = -1; char rest[128] = "";
    bool is_dec = false;
    if (sscanf(name, "model.encoder.layers.%d.%127s", &li, rest) == 2) is_dec = false;
    else if (sscanf(name, "model.decoder.layers.%d.%127s", &li, rest) == 2) is_dec = true;
    else return fail(c, CW_ERR_INVALID, "unknown tensor name %s", name);
    if (li < 0 || li >= (is_dec ? c->d.dec_layers : c->d.enc_layers)) return fail(c, CW_ERR_INVALID, "layer index out of range in %s", name);
    LayerW& L = is_dec ? c->dec[li] : c->enc[li];
    std::string r(rest);
    const float qs = 0.125f;  // head_dim ** -0.5, folded into q projections (exact: power of two)
    const size_t DD = (size_t)D * D;
    if (c->wpacked && is_dec)
        return fail(c, CW_ERR_STATE, "decoder weights were already packed for the decode GEMVs: create a new context to load another checkpoint");
    if (is_dec && c->bf16 && c->fold_enabled) {   // LayerNorm-fed decode projections: folded once all tensors are in
        if (c->ln_folded) return fail(c, CW_ERR_STATE, "weights were already folded: create a new context to load another checkpoint");
        if (r == "self_attn.q_proj.weight") { CWCHK(c, expect(DD)); return stage_fold(c, li, 0, data, D, D, L.wqkv, 0, L.bqkv, 0, qs); }
        if (r == "self_attn.k_proj.weight") { CWCHK(c, expect(DD)); return stage_fold(c, li, 0, data, D, D, L.wqkv, DD, L.bqkv, (size_t)D, 1.f); }
        if (r == "self_attn.v_proj.weight") { CWCHK(c, expect(DD)); return stage_fold(c, li, 0, data, D, D, L.wqkv, 2 * DD, L.bqkv, 2 * (size_t)D, 1.f); }
        if (r == "encoder_attn.q_proj.weight") { CWCHK(c, expect(DD)); return stage_fold(c, li, 1, data, D, D, L.wq_c, 0, L.bq_c, 0, qs); }
        if (r == "fc1.weight") { CWCHK(c, expect((size_t)F * D)); return stage_fold(c, li, 2, data, F, D, L.w1, 0, L.b1, 0, 1.f); }
        if (c->fuse6_enabled && (r == "self_attn.out_proj.weight" || r == "encoder_attn.out_proj.weight")) {   // f32 copy for the products
            CWCHK(c, expect(DD));
            cw_ctx::OutStage o{nullptr, li, r[0] == 's' ? 0 : 1};
            HIPCHK(c, hipMalloc((void**)&o.stage, DD * 4));
            HIPCHK(c, hipMemcpy(o.stage, data, DD * 4, hipMemcpyHostToDevice));
            c->out_stages.push_back(o);
        }
    }
    if (r == "self_attn.q_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wqkv, 0, data, n, qs); }
    if (r == "self_attn.k_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wqkv, DD, data, n); }
    if (r == "self_attn.v_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wqkv, 2 * DD, data, n); }
    if (r == "self_attn.q_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bqkv, 0, data, n, qs); }
    if (r == "self_attn.v_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bqkv, 2 * (size_t)D, data, n); }
    if (r == "self_attn.out_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wo, 0, data, n); }
    if (r == "self_attn.out_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bo, 0, data, n); }
    if (r == "self_attn_layer_norm.weight") { CWCHK(c, expect(D)); return upload_f32(c, L.ln1_g, 0, data, n); }
    if (r == "self_attn_layer_norm.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.ln1_b, 0, data, n); }
    if (r == "fc1.weight") { CWCHK(c, expect((size_t)F * D)); return upload_T(c, L.w1, 0, data, n); }
    if (r == "fc1.bias") { CWCHK(c, expect(F)); return upload_f32(c, L.b1, 0, data, n); }
    if (r == "fc2.weight") { CWCHK(c, expect((size_t)F * D)); return upload_T(c, L.w2, 0, data, n); }
    if (r == "fc2.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.b2, 0, data, n); }
    if (r == "final_layer_norm.weight") { CWCHK(c, expect(D)); return upload_f32(c, L.ln2_g, 0, data, n); }
    if (r == "final_layer_norm.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.ln2_b, 0, data, n); }
    if (is_dec) {
        if (r == "encoder_attn.q_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wq_c, 0, data, n, qs); }
        if (r == "encoder_attn.q_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bq_c, 0, data, n, qs); }
        if (r == "encoder_attn.k_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wkv_c, 0, data, n); }
        if (r == "encoder_attn.v_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wkv_c, DD, data, n); }
        if (r == "encoder_attn.v_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bkv_c, (size_t)D, data, n); }
        if (r == "encoder_attn.out_proj.weight") { CWCHK(c, expect(DD)); return upload_T(c, L.wo_c, 0, data, n); }
        if (r == "encoder_attn.out_proj.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.bo_c, 0, data, n); }
        if (r == "encoder_attn_layer_norm.weight") { CWCHK(c, expect(D)); return upload_f32(c, L.lnc_g, 0, data, n); }
        if (r == "encoder_attn_layer_norm.bias") { CWCHK(c, expect(D)); return upload_f32(c, L.lnc_b, 0, data, n); }
    }
    return fail(c, CW_ERR_INVALID, "unknown tensor name %s", name);
}

int32_t cw_set_generation(cw_ctx* c, const cw_gen_cfg* g) {
    const int V = c->d.vocab_size;
    std::vector<unsigned char> mask(V, 0);
    for (int i = 0; i < g->n_suppress; ++i) {
        int t = g->suppress_tokens[i];
        if (t < 0 || t >= V) return fail(c, CW_ERR_INVALID, "suppress token %d out of range", t);
        mask[t] |= 1;
    }
    for (int i = 0; i < g->n_begin_suppress; ++i) {
        int t = g->begin_suppress_tokens[i];
        if (t < 0 || t >= V) return fail(c, CW_ERR_INVALID, "begin-suppress token %d out of range", t);
        mask[t] |= 2;
    }
    if (g->no_timestamps_token_id < 0 || g->no_timestamps_token_id + 1 >= V) return fail(c, CW_ERR_INVALID, "no_timestamps_token_id out of range");
    mask[g->no_timestamps_token_id] |= 1;  // logits_process.py:2003
    HIPCHK(c, hipMemcpy(c->d_mask, mask.data(), V, hipMemcpyHostToDevice));
    for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // token ids are baked into captured kernel args
    c->gen = *g;
    c->gen.suppress_tokens = nullptr;
    c->gen.begin_suppress_tokens = nullptr;
    c->gen_set = true;
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// front end
// ------------------------------------------------------------------------------------------------
int32_t cw_upload_pcm(cw_ctx* c, const float* pcm, int32_t B, const int32_t* n_samples) {
    if (B < 1 || B > c->Bm) return fail(c, CW_ERR_INVALID, "B=%d out of range (max_batch %d)", B, c->Bm);
    HIPCHK(c, hipMemsetAsync(c->d_pcm, 0, (size_t)B * CW_N_SAMPLES * 4, c->st));
    size_t off = 0;
    for (int b = 0; b < B; ++b) {
        int n = n_samples[b];
        if (n < 0 || n > CW_N_SAMPLES) return fail(c, CW_ERR_INVALID, "n_samples[%d]=%d out of range", b, n);
        HIPCHK(c, hipMemcpyAsync(c->d_pcm + (size_t)b * CW_N_SAMPLES, pcm + off, (size_t)n * 4, hipMemcpyHostToDevice, c->st));
        off += n;
        c->n_frames_items[b] = (n + 159) / 160;  // attention_mask[:, ::hop].sum()  (feature_extraction_whisper.py:332-341)
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    return CW_OK;
}

int32_t cw_mel_resident(cw_ctx* c, int32_t B) {
    if (B < 1 || B > c->Bm) return fail(c, CW_ERR_INVALID, "B=%d out of range", B);
    StageTimer tm(c, CW_STAGE_MEL);
    CWCHK(c, KD(c, cw_launch_mel, c->mel, c->d_pcm, B, c->d.n_mels, c->d_logspec, c->d_gmax, c->st));
    CWCHK(c, KD(c, cw_launch_mel_finish, c->d_logspec, c->d_gmax, B, c->d.n_mels, c->d_feats_tm, c->bf16 ? 1 : 0, c->d_feats_hf, c->st));
    KCHK(c);
    tm.stop();
    return CW_OK;
}

int32_t cw_mel(cw_ctx* c, const float* pcm, int32_t B, const int32_t* n_samples, float* feats_out, int32_t* n_frames_out) {
    CWCHK(c, cw_upload_pcm(c, pcm, B, n_samples));
    if (feats_out && !c->d_feats_hf) CWCHK(c, dmalloc(c, &c->d_feats_hf, (size_t)c->Bm * CW_N_FRAMES * c->d.n_mels * 4));
    float* keep = c->d_feats_hf;
    if (!feats_out) c->d_feats_hf = nullptr;
    int r = cw_mel_resident(c, B);
    c->d_feats_hf = keep;
    if (r != CW_OK) return r;
    if (feats_out) HIPCHK(c, hipMemcpy(feats_out, c->d_feats_hf, (size_t)B * CW_N_FRAMES * c->d.n_mels * 4, hipMemcpyDeviceToHost));
    if (n_frames_out) for (int b = 0; b < B; ++b) n_frames_out[b] = c->n_frames_items[b];
    return CW_OK;
}

int32_t cw_set_features(cw_ctx* c, const float* feats, int32_t B) {
    if (B < 1 || B > c->Bm) return fail(c, CW_ERR_INVALID, "B=%d out of range", B);
    const int NM = c->d.n_mels;
    std::vector<float> tmaj((size_t)B * CW_N_FRAMES * NM);
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < NM; ++m)
            for (int t = 0; t < CW_N_FRAMES; ++t)
                tmaj[((size_t)b * CW_N_FRAMES + t) * NM + m] = feats[((size_t)b * NM + m) * CW_N_FRAMES + t];
    return upload_T(c, c->d_feats_tm, 0, tmaj.data(), tmaj.size());
}

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
static EpiParams epi0() { EpiParams p; memset(&p, 0, sizeof(p)); return p; }
static DecAttnParams dec_attn(const float* q, const void* K, const void* V, int cap, int n_keys, const int* pos, float* out,
                              int B, int H) {
    DecAttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.K = K; p.V = V; p.cap = cap; p.n_keys = n_keys; p.pos = pos; p.out = out; p.B = B; p.H = H;
    return p;
}

static int enc8_quantise(cw_ctx* c);

int32_t cw_encode(cw_ctx* c, int32_t nb, const int32_t* item, const int32_t* seek, const int32_t* n_frames) {
    const int D = c->d.d_model, H = c->d.n_heads, F = c->d.ffn_dim, NM = c->d.n_mels, S = CW_N_CTX;
    if (nb < 1 || nb > c->Bm) return fail(c, CW_ERR_INVALID, "nb=%d out of range", nb);
    std::vector<int> off(nb), val(nb);
    for (int i = 0; i < nb; ++i) {
        if (item[i] < 0 || item[i] >= c->Bm || seek[i] < 0 || n_frames[i] < 0 || seek[i] + n_frames[i] > CW_N_FRAMES)
            return fail(c, CW_ERR_INVALID, "bad window %d: item=%d seek=%d n=%d", i, item[i], seek[i], n_frames[i]);
        off[i] = item[i] * CW_N_FRAMES + seek[i];
        val[i] = n_frames[i];
    }
    CWCHK(c, cw_check_weights(c));
    if (c->enc8 && c->enc8_stale) CWCHK(c, enc8_quantise(c));
    HIPCHK(c, hipMemcpyAsync(c->d_row_off, off.data(), nb * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(c->d_row_valid, val.data(), nb * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));  // host vectors go out of scope
    const bool bf = c->bf16;
    StageTimer tm(c, CW_STAGE_ENCODER);
    {   // conv1 + GELU (modeling_whisper.py:618) as implicit GEMM over time-major features
        AParams ap{c->d_feats_tm, 0, 1, CW_N_FRAMES, NM, 1, c->d_row_off, c->d_row_valid};
        EpiParams ep = epi0(); ep.out = c->c1; ep.bias = c->conv1_b; ep.ldo = D;
        CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_GELU, ap, c->conv1_w, nb * CW_N_FRAMES, D, 3 * NM, ep, c->st));
    }
    {   // conv2 (stride 2) + GELU + sinusoidal positions (:619-624) -> f32 residual stream
        AParams ap{c->c1, 0, 1, S, D, 2, c->d_row_off2, c->d_row_valid2};
        EpiParams ep = epi0(); ep.outf = c->x; ep.bias = c->conv2_b; ep.ldo = D; ep.pos = c->enc_pos; ep.T = S;
        CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_GELU_POS_F32, ap, c->conv2_w, nb * S, D, 3 * D, ep, c->st));
    }
    const int M = nb * S;
    for (int l = 0; l < c->d.enc_layers; ++l) {
        LayerW& L = c->enc[l];
        if (c->enc8_mask & 1) CWCHK(c, KD(c, cw_launch_layernorm_fp8, c->x, L.ln1_g, L.ln1_b, c->h8, c->sa8, M, D, c->st));
        else CWCHK(c, KD(c, cw_launch_layernorm, bf, c->x, L.ln1_g, L.ln1_b, c->h, M, D, c->st));
        {
            AParams ap{c->h, D, 0, 0, 0, 0, nullptr, nullptr};
            EpiParams ep = epi0(); ep.out = c->qb; ep.out1 = c->kb; ep.out2 = c->vb; ep.bias = L.bqkv;
            ep.T = S; ep.S_pad = c->S_pad; ep.H = H; ep.d_model = D;
            if (c->enc8_mask & 1) CWCHK(c, KD(c, cw_launch_gemm_fp8, EPI_HEADS, c->h8, D, L.wqkv8, M, 3 * D, D, c->sa8, L.sqkv, ep, c->st));
            else CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_HEADS, ap, L.wqkv, M, 3 * D, D, ep, c->st));
        }
        CWCHK(c, KD(c, cw_launch_attn_encoder, bf, c->qb, c->kb, c->vb, c->ao, nb, H, S, c->S_pad, c->st));
        {
            AParams ap{c->ao, D, 0, 0, 0, 0, nullptr, nullptr};
            EpiParams ep = epi0(); ep.outf = c->x; ep.resid = c->x; ep.bias = L.bo; ep.ldo = D;
            CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_RESID_F32, ap, L.wo, M, D, D, ep, c->st));
        }
        if (c->enc8_mask & 2) CWCHK(c, KD(c, cw_launch_layernorm_fp8, c->x, L.ln2_g, L.ln2_b, c->h8, c->sa8, M, D, c->st));
        else CWCHK(c, KD(c, cw_launch_layernorm, bf, c->x, L.ln2_g, L.ln2_b, c->h, M, D, c->st));
        {
            AParams ap{c->h, D, 0, 0, 0, 0, nullptr, nullptr};
            EpiParams ep = epi0(); ep.out = c->mid; ep.bias = L.b1; ep.ldo = F;
            if (c->enc8_mask & 2) CWCHK(c, KD(c, cw_launch_gemm_fp8, EPI_GELU, c->h8, D, L.w18, M, F, D, c->sa8, L.s1, ep, c->st));
            else CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_GELU, ap, L.w1, M, F, D, ep, c->st));
        }
        {
            AParams ap{c->mid, F, 0, 0, 0, 0, nullptr, nullptr};
            EpiParams ep = epi0(); ep.outf = c->x; ep.resid = c->x; ep.bias = L.b2; ep.ldo = D;
            if (c->enc8_mask & 4) {   // GELU output: its row maxima are only known once fc1 is complete -> one row-wise quantisation pass
                CWCHK(c, KD(c, cw_launch_quant_rows_fp8, c->mid, M, F, c->mid8, c->smid8, c->st));
                CWCHK(c, KD(c, cw_launch_gemm_fp8, EPI_RESID_F32, c->mid8, F, L.w28, M, D, F, c->smid8, L.s2, ep, c->st));
            } else CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_RESID_F32, ap, L.w2, M, D, F, ep, c->st));
        }
    }
    CWCHK(c, KD(c, cw_launch_layernorm, bf, c->x, c->enc_ln_g, c->enc_ln_b, c->enc_out, M, D, c->st));   // cw_get_encoder_output
    if (c->enc8_mask & 8) CWCHK(c, KD(c, cw_launch_layernorm_fp8, c->x, c->enc_ln_g, c->enc_ln_b, c->h8, c->sa8, M, D, c->st));
    KCHK(c);
    tm.stop();
    StageTimer tk(c, CW_STAGE_CROSS_KV);
    for (int l = 0; l < c->d.dec_layers; ++l) {   // cross-attention K/V, once per window (:322-335)
        LayerW& L = c->dec[l];
        AParams ap{c->enc_out, D, 0, 0, 0, 0, nullptr, nullptr};
        EpiParams ep = epi0(); ep.out = L.ck; ep.out1 = L.cv; ep.bias = L.bkv_c;
        ep.T = S; ep.S_pad = S; ep.H = H; ep.d_model = D;
        if (c->enc8_mask & 8) CWCHK(c, KD(c, cw_launch_gemm_fp8, EPI_HEADS, c->h8, D, L.wkv_c8, M, 2 * D, D, c->sa8, L.skv_c, ep, c->st));
        else CWCHK(c, KD(c, cw_launch_gemm, bf, EPI_HEADS, ap, L.wkv_c, M, 2 * D, D, ep, c->st));
        if (c->kv8) CWCHK(c, KD(c, cw_launch_kv_quant_fp8, L.ck, L.cv, L.ck8, L.cv8, L.kvs, nb, H, S, c->st));
    }
    KCHK(c);
    tk.stop();
    c->nb_encoded = nb;
    return CW_OK;
}

int32_t cw_get_encoder_output(cw_ctx* c, float* out, int32_t nb) {
    if (nb < 1 || nb > c->nb_encoded) return fail(c, CW_ERR_STATE, "only %d windows encoded", c->nb_encoded);
    HIPCHK(c, hipStreamSynchronize(c->st));
    return download_T(c, c->enc_out, 0, out, (size_t)nb * CW_N_CTX * c->d.d_model);
}

// ------------------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------------------
static int gemv_ln(cw_ctx* c, int epi, const float* x, int Mb, int K, const void* W, int N, const float* g,
                   const float* b, const EpiParams& ep) {
    // (17..64 rows over row-major weights -- CW_NO_WPACK=1, or a geometry pack_decoder_weights does not take: no scratch, i.e. groups of
    // 16 rows on the <= 16-row kernels; gemv_mt_kernel reads fragment-major weights only)
    if (c->bf16 || !g) return KD(c, cw_launch_gemv, c->bf16, epi, x, Mb, K, W, N, g, b, ep, c->st, nullptr, (Mb <= 16 || c->wpacked) ? c->d_xfrag : nullptr, c->bf16 && c->wpacked);
    int r = KD(c, cw_launch_layernorm_f32, x, g, b, c->dxn, Mb, K, c->st);   // f32 parity mode: unfused LN
    if (r != CW_OK) return r;
    return KD(c, cw_launch_gemv, false, epi, c->dxn, Mb, K, W, N, nullptr, nullptr, ep, c->st);
}


#ifdef CW_EXPERIMENTS
static int launch_prefetch_layer(cw_ctx* c, int l, int nb) {
    const int D = c->d.d_model, F = c->d.ffn_dim, H = c->d.n_heads;
    const size_t e = c->esz;
    LayerW& L = c->dec[l];
    PrefetchRanges r;
    memset(&r, 0, sizeof(r));
    auto add = [&](const void* p, size_t bytes) { if (p && bytes && r.count < 8) { r.p[r.count] = (const char*)p; r.n[r.count] = bytes; ++r.count; } };
    add(L.wqkv, (size_t)3 * D * D * e);
    if (c->fuse6_ready && c->fuse6_enabled && nb <= 16 && c->beam_K == 0) add(L.ws3, (size_t)3 * D * D * e);
    else { add(L.wo, (size_t)D * D * e); add(L.wq_c, (size_t)D * D * e); }
    add(L.ck, (size_t)nb * H * CW_N_CTX * 64 * e);
    add(L.cv, (size_t)nb * H * CW_N_CTX * 64 * e);
    add(L.wo_c, (size_t)D * D * e);
    add(L.w1, (size_t)F * D * e);
    add(L.w2, (size_t)D * F * e);
    const int wide = cw_sw::cw_switches().prefetch_wide;
    const int what = cw_sw::cw_switches().prefetch_what;   // 1 weights, 2 cross K/V, 3 both
    if (!(what & 1)) { int k = 0; for (int i = 0; i < r.count; ++i) if (r.p[i] == (const char*)L.ck || r.p[i] == (const char*)L.cv) { r.p[k] = r.p[i]; r.n[k] = r.n[i]; ++k; } r.count = k; }
    if (!(what & 2)) { int k = 0; for (int i = 0; i < r.count; ++i) if (r.p[i] != (const char*)L.ck && r.p[i] != (const char*)L.cv) { r.p[k] = r.p[i]; r.n[k] = r.n[i]; ++k; } r.count = k; }
    if (wide) hipLaunchKernelGGL(prefetch_kernel<1>, dim3(c->prefetch), dim3(256), 0, c->st2, r, c->d_pf_sink);
    else hipLaunchKernelGGL(prefetch_kernel<0>, dim3(c->prefetch), dim3(256), 0, c->st2, r, c->d_pf_sink);
    return CW_OK;
}
#else
static int launch_prefetch_layer(cw_ctx*, int, int) { return CW_ERR_INVALID; }
#endif

// persistent decoder-layer launch of layer l (declayer.hip): fused out-projection / cross-query stage + cross-attention
static DecLayerParams dec_layer_params(cw_ctx* c, int l, int nb, const float* xin, float* xalt) {
    const int D = c->d.d_model, H = c->d.n_heads, TGT = c->d.max_target_positions;
    LayerW& L = c->dec[l];
    DecLayerParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.Ws = L.ws3; dp.x = xin; dp.a = c->dattn; dp.qa_bias = L.qa_bias; dp.q_wsum = c->stack_center ? L.q_wsum : nullptr;
    dp.bo = L.bo; dp.x1 = xalt;
    dp.K = L.ck; dp.V = L.cv; dp.n_keys = CW_N_CTX; dp.part_o = c->d_part_o; dp.part_ml = c->d_part_ml;
    dp.align_out = c->d.n_align > 0 ? c->d_align : nullptr; dp.align_ml = c->d_align_ml;
    dp.align_slot = c->d_align_slot + (size_t)l * H; dp.pos = c->d_pos; dp.n_align = c->d.n_align; dp.align_rows = TGT;
    dp.qw = L.q_wsum; dp.qbias = L.bq_c;
    dp.gq = c->d_gq; dp.gps = c->d_gps; dp.epoch = c->d_epoch; dp.layer = l; dp.err = c->d_err;
    dp.Mb = nb; dp.D = D; dp.H = H;
    dp.kv_wait = cw_sw::cw_switches().dl_kvwait;
    return dp;
}

// One decoder forward for the nb rows at the positions held in c->d_pos (device): 8 launches per layer.
static int decode_step(cw_ctx* c, int nb, bool want_logits) {
    const int D = c->d.d_model, H = c->d.n_heads, F = c->d.ffn_dim, V = c->d.vocab_size;
    const int TGT = c->d.max_target_positions;
    // 17..64 rows (bf16): producers hand activations to the next GEMV already in MFMA fragment order (no prep launch)
    const bool frag = c->bf16 && nb > 16 && c->wpacked;      // (row-major weights: groups of 16 rows on the <= 16-row kernels, see gemv_ln)
    // fused out-projection / cross-query stage (decfuse.hip): greedy rows of one MFMA half tile, 16-bit caches.  The residual
    // stream then alternates between two buffers: a layer reads x from `xin` and leaves x1, x2, x3 in `xalt`.
    // (e4m3 cache: the fp8 matrix-core kernel finishes the fused query too; its VALU fallback does not)
    // 17..64 greedy rows (round 5): the same stage in groups of 16 rows (gemv_stack_kernel's grid y) replaces out-projection,
    // LayerNorm preparation and query projection -- three of the layer's twelve launches -- by one; the rest of the layer keeps its
    // preparation + gemv_mt launches, on the alternating buffers.  CW_NO_FUSE_ROWS=1: the twelve launches (A/B).  Batch 64: decode
    // 4.43 -> 4.28 ms per token step, 64 / 64 clips; over the e4m3 cache 3.47 -> 3.41 once the cross-attention kernel lets wave 0 alone
    // finish the query (attn_cross_mfma8_kernel<1, 2>; every wave doing it, as at <= 16 rows, measured flat).  CW_NO_FUSE_ROWS8=1: A/B
    // (the rejected 17..64-row A/B variants of -DCW_EXPERIMENTS builds -- rows path, skinny GEMMs, full-key cross-attention -- keep their
    // own twelve / nine launches: they read c->dx and d_xfrag, which the fused stage's alternating buffers would leave stale)
    const bool ab17 = nb > 16 && (c->rows_ln_enabled || c->skinny_mode != 0);
    const bool fuse = c->bf16 && c->fuse6_ready && c->fuse6_enabled && c->ln_folded && !ab17 && (nb <= 16 || frag) && (nb <= 16 || (c->fuse_rows && nb <= 64 && (!c->kv8 || c->fuse_rows8))) &&
                      // beam search (round 6): the hypotheses of an item share a cross-attention block that finishes their queries
                      // (attn_cross_mfma_kernel<.., FUSED>, 16-bit cache); CW_NO_FUSE_BEAM=1: the twelve launches
                      (c->beam_K == 0 || (c->fuse_beam && !c->kv8 && c->beam_K <= 16 && !c->fuse_mlp)) &&
                      (!c->kv8 || (KD(c, cw_cross8_is_mfma, CW_N_CTX) && !c->fuse_mlp)) && !((c->fuse_mlp || c->mlp_pair) && nb > 8);
    float *xin = c->dx, *xalt = c->dx1;
    // 17..64 rows without preparation launches (decfuse.hip: gemv_rows_kernel): the residual GEMVs own whole columns and leave
    // the stream in f32, its 16-bit fragment-major copy in d_xfrag and LayerNorm partial sums in d_rstats; the LayerNorm
    // GEMVs read that copy and normalise their outputs.  d_xfrag2 carries attention outputs and the GELU'd MLP rows.
    // 17..64 rows, DEFAULT (skinny_mode 0): the round-3 path -- gemv_prep_kernel (LayerNorm / combine -> fragment-major rows) +
    // gemv_mt_kernel per projection.  A/B only (-DCW_EXPERIMENTS builds, CW_SKINNY=1|2; measured slower in the step,
    // profiles/r04_b64_skinny_planes_rejected_*): the LayerNorm projections as skinny-M GEMMs (skinny.hip): f32 rows -> K-split
    // partial planes, finished (statistics, bias, cache / GELU epilogue) by one light launch.  The residual projections stay on
    // the 16-column K-split GEMV either way: their cost is the f32 atomics (7 ps each), which a wider tile with more K slices multiplies.
    const bool sk_ok = frag && c->wpacked && c->ln_folded && c->rows_ln_ready && c->d_planes;
    const bool skinny = sk_ok && c->skinny_mode >= 2;
    // greedy rows over the 16-bit cache: one cross-attention block per (row, head) over all keys writes the out-projection's
    // 16-bit rows itself (1280 blocks at 64 rows: the chip is full without key splits)
    const bool xfull = frag && c->skinny_mode >= 1 && c->beam_K == 0 && !c->kv8 && !c->rows_ln_enabled && H <= 20 && CW_N_CTX <= 24 * 64;
    auto sk_ln = [&](int epi, const void* W, int N, const float* wsum, const EpiParams& ep) -> int {
        SkinnyParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.x = c->dx; sp.W = W; sp.Mb = nb; sp.K = D; sp.N = N; sp.planes = c->d_planes;
        int smax = (int)(c->planes_cap / ((size_t)nb * N));
        if (smax > 16) smax = 16;
        const int nks = KD(c, cw_skinny_pick_nks, N, D, smax);
        if (nks < 1) return CW_ERR_INVALID;
        sp.stats = c->d_sk_stats;
        int r = KD(c, cw_launch_skinny, 0, sp, nks, c->st);
        if (r != CW_OK) return r;
        SkinnyFinishParams fp;
        memset(&fp, 0, sizeof(fp));
        fp.planes = c->d_planes; fp.S = (D / 32) / nks; fp.Mb = nb; fp.N = N; fp.x = c->dx; fp.K = D; fp.wsum = wsum; fp.ep = ep;
        fp.stats = c->d_sk_stats;
        return KD(c, cw_launch_skinny_finish, epi, fp, c->st);
    };
    const bool rows = !skinny && frag && c->rows_ln_ready && c->rows_ln_enabled && c->ln_folded && D <= 1280;
    int ln_nblk = 1;
    // bf16 only: the 16-bit copy of the residual rows keeps 8 mantissa bits of values that are NOT normalised yet; carried as
    // hi + lo halves (two MFMAs per fragment) the LayerNorm GEMVs see 16 bits, more than the rounded LN(x) of the prepared path
    const int lo_off = (c->rows_hilo && !c->f16) ? 64 * (D / 8) : 0;
    auto rows_consume = [&](int epi, const void* W, int N, const EpiParams& ep, const float* wsum) -> int {
        RowsParams rp;
        memset(&rp, 0, sizeof(rp));
        rp.xf = c->d_xfrag; rp.W = W; rp.Mb = nb; rp.K = D; rp.N = N; rp.wpk = c->wpacked ? 1 : 0; rp.ep = ep;
        rp.ln_pstats = c->d_rstats; rp.ln_nblk = ln_nblk; rp.ln_wsum = wsum; rp.lo_off = lo_off;
        return KD(c, cw_launch_gemv_rows, epi, false, rp, c->st);
    };
    auto rows_produce = [&](const void* W, int K, const float* bias) -> int {   // x += W a + b over the rows in d_xfrag2
        RowsParams rp;
        memset(&rp, 0, sizeof(rp));
        rp.xf = c->d_xfrag2; rp.W = W; rp.Mb = nb; rp.K = K; rp.N = D; rp.wpk = c->wpacked ? 1 : 0;
        rp.ep = epi0(); rp.ep.outf = c->dx; rp.ep.resid = c->dx; rp.ep.bias = bias; rp.ep.ldo = D;
        rp.xf_out = c->d_xfrag; rp.pstats_out = c->d_rstats; rp.lo_off = lo_off;
        ln_nblk = D / 16;
        return KD(c, cw_launch_gemv_rows, (int)EPI_RESID_F32, true, rp, c->st);
    };
    if (rows) CWCHK(c, KD(c, cw_launch_rows_prep, c->dx, nb, D, c->d_xfrag, c->d_rstats, lo_off, c->st));
    // 33..64 rows behind the fused stage (round 6): the cross-attention out-projection owns whole columns (no K split, no atomics) and
    // leaves fc1 its 16-bit rows and LayerNorm partial sums -- one preparation launch less per layer (gemm.hip: gemv_mt_kernel OWN / LNA)
    const bool own = c->own_cols && fuse && frag && nb > 32 && !c->fuse_mlp && !c->mlp_pair && c->rows_ln_ready && D % 32 == 0 && D / 16 <= 96 && D <= 1536 &&
                     D % 128 == 0 && F % 32 == 0;
    const bool pf = c->prefetch > 0 && c->bf16 && c->beam_K == 0 && !c->kv8;
    // cw_time_decode_stage runs ONE launch (stage_sel) of ONE layer (layer_sel) through this very code: the kernels it times are
    // the ones the step launches, with the step's arguments
#define STG(kind, call)                                                                             \
    do {                                                                                            \
        if (c->stage_sel < 0 || c->stage_sel == stage_no) CWCHK(c, call);                           \
        if (stage_no < CW_MAX_DEC_STAGES) {                                                         \
            c->stage_kind[stage_no] = (kind);                                                       \
            /* 17..64 rows: a LayerNorm / combining GEMV is gemv_prep_kernel + gemv_mt_kernel */    \
            c->stage_launches[stage_no] = 1 + ((frag && !skinny && !rows && !xfull && ((kind) == DST_QKV || (kind) == DST_CROSS_Q || (kind) == DST_FC1 || (kind) == DST_CROSS_O) && \
                                                !(own && ((kind) == DST_FC1 || (kind) == DST_CROSS_O))) ? 1 : 0); \
        }                                                                                           \
        c->stage_count = ++stage_no;                                                                \
    } while (0)
    const int l_lo = c->layer_sel >= 0 ? c->layer_sel : 0, l_hi = c->layer_sel >= 0 ? c->layer_sel + 1 : c->d.dec_layers;
    for (int l = l_lo; l < l_hi; ++l) {
        LayerW& L = c->dec[l];
        int stage_no = 0;
        if (pf && l + 1 < c->d.dec_layers) {   // layer l + 1's streams are touched while layer l runs (side stream: a parallel graph branch)
            HIPCHK(c, hipEventRecord(c->ev_pf[0], c->st));
            HIPCHK(c, hipStreamWaitEvent(c->st2, c->ev_pf[0], 0));
            CWCHK(c, launch_prefetch_layer(c, l + 1, nb));
        }
        // rows <= 8, 16-bit engines: LN + q/k/v projection + self-attention as ONE launch (declayer.hip: qkv_self_kernel; the tiles
        // reach the attention blocks as granules, the history rows of the cache are requested at kernel entry); bit-identical
        // (its 3 D / 16 blocks wait for each other: only where the whole grid is resident at once on this part; layer < 64: granule tag)
        const bool qs = c->qkv_self && c->bf16 && c->ln_folded && c->wpacked && nb <= 8 && c->beam_K == 0 && D <= 1280 && nb * H <= 3 * (D / 16) && TGT <= 512 &&
                        c->d.dec_layers <= 64 && 3 * (D / 16) <= c->n_cu * KD(c, cw_qkv_self_blocks_per_cu, D, TGT);
        if (qs) {
            QkvSelfParams qp;
            memset(&qp, 0, sizeof(qp));
            qp.x = xin; qp.W = L.wqkv; qp.bias = L.bqkv; qp.sk = L.sk; qp.sv = L.sv; qp.cap = TGT; qp.pos = c->d_pos; qp.out = c->dattn;
            qp.gq = c->d_gq2; qp.gkv = c->d_gkv; qp.epoch = c->d_epoch; qp.layer = l; qp.err = c->d_err; qp.Mb = nb; qp.D = D; qp.H = H; qp.fail_pos = c->fail_pos;
            const int dbg = cw_sw::cw_switches().qkv_self_dbg;   // 1: tiles fused, attention by attn_decode_kernel behind it (bisecting aid)
            if (dbg) { qp.q_plain = c->dq; qp.no_attn = 1; }
            STG(DST_QKV_SELF, KD(c, cw_launch_qkv_self, qp, c->st));
            if (dbg) {
                DecAttnParams p = dec_attn(c->dq, L.sk, L.sv, TGT, 0, c->d_pos, c->dattn, nb, H);
                STG(DST_SELF_ATTN, KD(c, cw_launch_attn_decode, c->bf16, p, c->st));
            }
        } else {
        {   // LN + fused q/k/v projection; k,v appended to the self-attention cache at pos[b]
            EpiParams ep = epi0(); ep.outf = c->dq; ep.out1 = L.sk; ep.out2 = L.sv; ep.bias = L.bqkv;
            ep.H = H; ep.S_pad = TGT; ep.d_model = D; ep.row_pos = c->d_pos;
            if (skinny) STG(DST_QKV, sk_ln(EPI_QKV_CACHE, L.wqkv, 3 * D, L.qkv_wsum, ep));
            else if (rows) STG(DST_QKV, rows_consume(EPI_QKV_CACHE, L.wqkv, 3 * D, ep, L.qkv_wsum));
            else STG(DST_QKV, gemv_ln(c, EPI_QKV_CACHE, xin, nb, D, L.wqkv, 3 * D, L.ln1_g, c->ln_folded ? nullptr : L.ln1_b, ep));
        }
        {
            DecAttnParams p = dec_attn(c->dq, L.sk, L.sv, TGT, 0, c->d_pos, c->dattn, nb, H);
            if (frag && !fuse) p.out_frag = (unsigned short*)c->d_xfrag2;   // the fused stage reads the f32 rows
            if (c->beam_K > 0) p.anc = c->d_anc;
            p.short_hist = (c->hist_short && c->short_hist_enabled && nb > 8) ? 1 : 0;   // bytes matter from 9 rows up (<= 8: latency-bound, qkv_self)
            STG(DST_SELF_ATTN, KD(c, cw_launch_attn_decode, c->bf16, p, c->st));
        }
        }
        if (fuse) {
            const int TD = D / 16, TF = F / 16;
            // column tiles per X1 block (3 * TD blocks at 1: 240 at large-v3).  17..64 rows: two -- a block's 16 f32 rows are twice the bytes
            // of a weight tile, and a pair of tiles shares them (64 rows, 4 groups: 11.2 us at one tile, 10.5 at two, 15.5 at three)
            const int nt3 = c->stack_nt3 > 0 ? c->stack_nt3 : (nb > 16 ? 2 : 1);
            // rows <= 8: X1 and the cross-attention as ONE persistent launch (declayer.hip): the K/V rows are requested at kernel
            // entry and stream under the GEMV tile; qa / qb / the partial sums cross CUs as granules.  Bit-identical to the two launches.
            const bool dl = c->declayer && !c->kv8 && !c->fuse_mlp && nt3 == 1 && c->wpacked && nb <= 8 && 3 * TD <= c->n_cu &&
                            nb * H * ATT_NS <= 4 * c->n_cu && TD <= 128 && KD(c, cw_dec_layer_lds, D) <= (size_t)160 * 1024;
            if (dl) {
                const DecLayerParams dp = dec_layer_params(c, l, nb, xin, xalt);
                STG(DST_STACK_CROSS, KD(c, cw_launch_dec_layer, dp, c->n_cu, c->st));
            } else
            {   // X1 over [W'q_c ; W'q_c Wo ; Wo]:  qa = W'q_c x + W'q_c bo,  qb = (W'q_c Wo) a,  x1 = x + Wo a + bo
                StackParams sp;
                memset(&sp, 0, sizeof(sp));
                sp.W = L.ws3; sp.wpk = c->wpacked ? 1 : 0; sp.K = D; sp.Mb = nb; sp.nseg = 3;
                sp.seg[0].x = xin;      sp.seg[0].bias = L.qa_bias; sp.seg[0].out = c->d_qa; sp.seg[0].tile0 = 0;      sp.seg[0].n_tiles = TD; sp.seg[0].epi = 0;
                sp.seg[0].wsum = c->stack_center ? L.q_wsum : nullptr;   // x is rounded as x - mean(x): the operand the cross-attention LayerNorm is sensitive to
                sp.seg[1].x = c->dattn; sp.seg[1].bias = nullptr;   sp.seg[1].out = c->d_qb; sp.seg[1].tile0 = TD;     sp.seg[1].n_tiles = TD; sp.seg[1].epi = 0;
                sp.seg[2].x = c->dattn; sp.seg[2].bias = L.bo;      sp.seg[2].out = xalt;    sp.seg[2].tile0 = 2 * TD; sp.seg[2].n_tiles = TD; sp.seg[2].epi = 1;
                sp.seg[2].resid = xin; sp.seg[2].pstats = c->d_pstats;   // LayerNorm partial sums of x1 for the cross-attention
                if (c->fuse_mlp) { sp.zero = c->d_u1; sp.zero_n4 = nb * F / 4; }   // X2 below accumulates its two halves into u1
                STG(DST_STACK, KD(c, cw_launch_gemv_stack, sp, nt3, c->st));
            }
            // cross-attention; the kernel finishes q_c = rstd(x1) (qa + qb - mean(x1) W'q_c 1) + b'q_c
            CrossSplitParams p{nullptr, L.ck, L.cv, CW_N_CTX, c->d_part_o, c->d_part_ml,
                               c->d.n_align > 0 ? c->d_align : nullptr, c->d_align_ml, c->d_align_slot + (size_t)l * H,
                               c->d_pos, c->d.n_align, TGT, nb, H};
            p.kv_div = c->beam_K > 0 ? c->beam_K : 1;
            p.qa = c->d_qa; p.qb = c->d_qb; p.qw = L.q_wsum; p.qbias = L.bq_c;
            p.pstats = c->d_pstats; p.n_pstats = (TD + nt3 - 1) / nt3;
            if (!c->fuse_mlp) {
                if (c->kv8) {
                    p.K = L.ck8; p.V = L.cv8; p.kv_scale = L.kvs;
                    STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split_fp8, p, c->st));
                } else
                if (!dl) STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split, true, p, c->st));
                if (own) {
                    // combine (+ the rows' centres) -> d_xfrag;  x2 = x1 + Wo_c a_c + bo_c in place, y = x2 - c -> d_xfrag2, partial sums;
                    // fc1 normalises on its accumulator, GELU rows -> d_xfrag;  fc2 as before
                    CombineParams cb{c->d_part_ml, H, nb * D, c->d_pstats, p.n_pstats, c->d_cvec};
                    STG(DST_OTHER, KD(c, cw_launch_rows_combine, c->d_part_o, nb, D, cb, c->d_xfrag, c->st));
                    {
                        EpiParams ep = epi0(); ep.outf = xalt; ep.resid = xalt; ep.bias = L.bo_c; ep.ldo = D;
                        STG(DST_CROSS_O, KD(c, cw_launch_gemv_own, c->d_xfrag, nb, D, L.wo_c, D, ep, c->d_cvec, c->d_xfrag2, c->d_ostats, c->st, c->wpacked));
                    }
                    {
                        EpiParams ep = epi0(); ep.out = c->d_xfrag; ep.bias = L.b1; ep.ldo = F;
                        STG(DST_FC1, KD(c, cw_launch_gemv_lna, c->d_xfrag2, nb, D, L.w1, F, ep, c->d_ostats, D / (16 * KD(c, cw_gemv_own_nt, D)), L.u1_wsum, c->st, c->wpacked));
                    }
                    {
                        EpiParams ep = epi0(); ep.outf = xalt; ep.resid = xalt; ep.bias = L.b2; ep.ldo = D;
                        STG(DST_FC2, KD(c, cw_launch_gemv, true, EPI_RESID_F32, nullptr, nb, F, L.w2, D, nullptr, nullptr, ep, c->st, nullptr, c->d_xfrag, c->wpacked));
                    }
                    float* t = xin; xin = xalt; xalt = t;
                    continue;
                }
                {   // out-projection combines the key-split partials; x2 = x1 + Wo_c a_c + bo_c in place
                    EpiParams ep = epi0(); ep.outf = xalt; ep.resid = xalt; ep.bias = L.bo_c; ep.ldo = D;
                    CombineParams cb{c->d_part_ml, H, nb * D};
                    STG(DST_CROSS_O, KD(c, cw_launch_gemv, true, EPI_RESID_F32, c->d_part_o, nb, D, L.wo_c, D, nullptr, nullptr, ep, c->st, &cb, c->d_xfrag, c->wpacked));
                }
                if (c->mlp_pair && F % D == 0 && D % 32 == 0 && F / 32 <= 256 && F / D <= 32) {
                    // LN + fc1 + GELU, group barrier, fc2 + residual in one launch (decfuse.hip: mlp_pair_kernel)
                    MlpPairParams mp{xalt, L.w1, L.b1, L.w2, L.b2, c->d_xfrag2, c->d_bar, c->d_err, nb, D, F, c->wpacked ? 1 : 0, c->mlp_pair_fence ? 1 : 0};
                    STG(DST_MLP_PAIR, KD(c, cw_launch_mlp_pair, mp, c->st));
                } else if (c->mlp_chain && c->wpacked && c->mid16 && nb <= 8 && KD(c, cw_mlp_chain_ok, nb, D, F)) {
                    // LN + fc1 + GELU and fc2 + residual in ONE launch: fc2's blocks request their weights at kernel entry and wait for
                    // the fc1 blocks' flags instead of a kernel boundary (declayer.hip: mlp_chain_kernel); bit-identical
                    MlpChainParams mp;
                    memset(&mp, 0, sizeof(mp));
                    mp.x = xalt; mp.W1 = L.w1; mp.b1 = L.b1; mp.W2 = L.w2; mp.b2 = L.b2; mp.xio = xalt; mp.mid = c->d_xfrag2;
                    mp.flags = c->d_gflag; mp.epoch = c->d_epoch; mp.layer = l; mp.err = c->d_err; mp.Mb = nb; mp.D = D; mp.F = F;
                    STG(DST_MLP_CHAIN, KD(c, cw_launch_mlp_chain, mp, c->st));
                } else if (frag) {
                    // 17..64 rows: LayerNorm preparation + gemv_mt launches of the twelve-launch layer, on the alternating buffer
                    {
                        EpiParams ep = epi0(); ep.outf = c->dmid; ep.out = c->d_xfrag2; ep.bias = L.b1; ep.ldo = F;
                        STG(DST_FC1, gemv_ln(c, EPI_GELU_FRAG, xalt, nb, D, L.w1, F, L.ln2_g, c->ln_folded ? nullptr : L.ln2_b, ep));
                    }
                    {
                        EpiParams ep = epi0(); ep.outf = xalt; ep.resid = xalt; ep.bias = L.b2; ep.ldo = D;
                        STG(DST_FC2, KD(c, cw_launch_gemv, true, EPI_RESID_F32, nullptr, nb, F, L.w2, D, nullptr, nullptr, ep, c->st, nullptr, c->d_xfrag2, c->wpacked));
                    }
                } else {
                    // fc1 writes gelu(.) in the 16-bit type fc2 would round it to anyway (bit-identical): fc2's activation load halves
                    const bool mid16 = F > 1280 && c->mid16;
                    {
                        EpiParams ep = epi0(); ep.outf = c->dmid; ep.out = c->d_xfrag2; ep.bias = L.b1; ep.ldo = F;
                        STG(DST_FC1, gemv_ln(c, mid16 ? EPI_GELU : EPI_GELU_F32, xalt, nb, D, L.w1, F, L.ln2_g, nullptr, ep));
                    }
                    {
                        EpiParams ep = epi0(); ep.outf = xalt; ep.resid = xalt; ep.bias = L.b2; ep.ldo = D; ep.x16 = mid16 ? 1 : 0;
                        STG(DST_FC2, gemv_ln(c, EPI_RESID_F32, mid16 ? (const float*)c->d_xfrag2 : c->dmid, nb, F, L.w2, D, nullptr, nullptr, ep));
                    }
                }
                float* t = xin; xin = xalt; xalt = t;
                continue;
            }
            // A/B (CW_FUSE_MLP=1), measured slower: cross out-projection + fc1 fused the same way.  Needs the
            // finished attention output in one plane (one block per (row, head): 17 us against 12) and makes every fc2 block
            // redo the statistics and the GELU of its K slice.
            p.a_out = c->dattn; p.xstat = xalt;
            STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split, true, p, c->st));
            {   // X2 over [W'1 ; W'1 Wo_c ; Wo_c]:  u1 = W'1 x1 + W'1 bo_c + (W'1 Wo_c) a_c (two halves, atomics into the zeros X1
                // left: two commutative additions, so the order does not matter),  x2 = x1 + Wo_c a_c + bo_c (+ a copy that
                // fc2 takes the LayerNorm statistics of while its atomics are already modifying the stream)
                StackParams sp;
                memset(&sp, 0, sizeof(sp));
                sp.W = L.ws5; sp.wpk = c->wpacked ? 1 : 0; sp.K = D; sp.Mb = nb; sp.nseg = 3;
                sp.seg[0].x = xalt;     sp.seg[0].bias = L.u1_bias; sp.seg[0].out = c->d_u1; sp.seg[0].tile0 = 0;      sp.seg[0].n_tiles = TF; sp.seg[0].epi = 2;
                sp.seg[0].wsum = c->stack_center ? L.u1_wsum : nullptr;
                sp.seg[1].x = c->dattn; sp.seg[1].bias = nullptr;   sp.seg[1].out = c->d_u1; sp.seg[1].tile0 = TF;     sp.seg[1].n_tiles = TF; sp.seg[1].epi = 2;
                sp.seg[2].x = c->dattn; sp.seg[2].bias = L.bo_c;    sp.seg[2].out = xin;     sp.seg[2].tile0 = 2 * TF; sp.seg[2].n_tiles = TD; sp.seg[2].epi = 1;
                sp.seg[2].resid = xalt; sp.seg[2].out2 = c->dx2c;
                STG(DST_STACK, KD(c, cw_launch_gemv_stack, sp, c->stack_nt5, c->st));
            }
            {   // fc2: mid = gelu(rstd(x2) (u1 - mean(x2) W'1 1) + b'1) on load; x3 = x2 + W2 mid + b2 in place
                Fc2xParams fp{c->dx2c, c->d_u1, L.u1_wsum, L.b1, L.w2, L.b2, xin, nb, D, F, c->wpacked ? 1 : 0};
                STG(DST_FC2, KD(c, cw_launch_gemv_fc2x, fp, c->st));
            }
            continue;
        }
        {
            EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.bo; ep.ldo = D;
            if (rows) STG(DST_O_PROJ, rows_produce(L.wo, D, L.bo));
            else if (frag) STG(DST_O_PROJ, KD(c, cw_launch_gemv, true, EPI_RESID_F32, nullptr, nb, D, L.wo, D, nullptr, nullptr, ep, c->st, nullptr, c->d_xfrag2, c->wpacked));
            else STG(DST_O_PROJ, gemv_ln(c, EPI_RESID_F32, c->dattn, nb, D, L.wo, D, nullptr, nullptr, ep));
        }
        if (xfull) {
            CrossSplitParams p{c->dq, L.ck, L.cv, CW_N_CTX, c->d_part_o, c->d_part_ml,
                               c->d.n_align > 0 ? c->d_align : nullptr, c->d_align_ml, c->d_align_slot + (size_t)l * H,
                               c->d_pos, c->d.n_align, TGT, nb, H};
            p.kv_div = 1; p.a_frag = c->d_xfrag;
            if (sk_ok) {   // W'q_c (x - c) over K slices -> planes; the attention blocks apply rstd (sum - mean W'q_c 1) + b'q_c
                SkinnyParams sp;
                memset(&sp, 0, sizeof(sp));
                sp.x = c->dx; sp.W = L.wq_c; sp.Mb = nb; sp.K = D; sp.N = D; sp.planes = c->d_planes;
                const int nks = KD(c, cw_skinny_pick_nks, D, D, (int)(c->planes_cap / ((size_t)nb * D)));
                if (nks < 1) return fail(c, CW_ERR_INVALID, "no K split for the cross-attention query GEMM");
                STG(DST_OTHER, KD(c, cw_launch_skinny, 0, sp, nks, c->st));
                p.q = nullptr; p.xstat = c->dx; p.qa = c->d_planes; p.q_planes = (D / 32) / nks; p.q_plane_stride = nb * D;
                p.qw = L.q_wsum; p.qbias = L.bq_c;
            } else {
                EpiParams ep = epi0(); ep.outf = c->dq; ep.bias = L.bq_c; ep.ldo = D;
                STG(DST_CROSS_Q, gemv_ln(c, EPI_STORE_F32, c->dx, nb, D, L.wq_c, D, L.lnc_g, c->ln_folded ? nullptr : L.lnc_b, ep));
            }
            STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split, true, p, c->st));
            EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.bo_c; ep.ldo = D;
            STG(DST_CROSS_O, KD(c, cw_launch_gemv, true, EPI_RESID_F32, nullptr, nb, D, L.wo_c, D, nullptr, nullptr, ep, c->st, nullptr, c->d_xfrag, c->wpacked));
        } else {
        {   // cross-attention: LN + q projection, attention over the cached encoder K/V
            EpiParams ep = epi0(); ep.outf = c->dq; ep.bias = L.bq_c; ep.ldo = D;
            if (skinny) STG(DST_CROSS_Q, sk_ln(EPI_STORE_F32, L.wq_c, D, L.q_wsum, ep));
            else if (rows) STG(DST_CROSS_Q, rows_consume(EPI_STORE_F32, L.wq_c, D, ep, L.q_wsum));
            else STG(DST_CROSS_Q, gemv_ln(c, EPI_STORE_F32, c->dx, nb, D, L.wq_c, D, L.lnc_g, c->ln_folded ? nullptr : L.lnc_b, ep));
        }
        if (c->bf16) {
            // keys split over ATT_NS blocks per (row, head); the out-projection GEMV combines the partials
            CrossSplitParams p{c->dq, L.ck, L.cv, CW_N_CTX, c->d_part_o, c->d_part_ml,
                               c->d.n_align > 0 ? c->d_align : nullptr, c->d_align_ml, c->d_align_slot + (size_t)l * H,
                               c->d_pos, c->d.n_align, TGT, nb, H};
            p.kv_div = c->beam_K > 0 ? c->beam_K : 1;
            if (c->kv8) {
                p.K = L.ck8; p.V = L.cv8; p.kv_scale = L.kvs;
                STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split_fp8, p, c->st));
            } else STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_cross_split, true, p, c->st));
            EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.bo_c; ep.ldo = D;
            CombineParams cb{c->d_part_ml, H, nb * D};
            if (rows) {
                STG(DST_OTHER, KD(c, cw_launch_rows_combine, c->d_part_o, nb, D, cb, c->d_xfrag2, c->st));
                STG(DST_CROSS_O, rows_produce(L.wo_c, D, L.bo_c));
            } else STG(DST_CROSS_O, KD(c, cw_launch_gemv, true, EPI_RESID_F32, c->d_part_o, nb, D, L.wo_c, D, nullptr, nullptr, ep, c->st, &cb, (nb <= 16 || frag) ? c->d_xfrag : nullptr, c->wpacked));
        } else {
            DecAttnParams p = dec_attn(c->dq, L.ck, L.cv, CW_N_CTX, CW_N_CTX, c->d_pos, c->dattn, nb, H);
            p.align_out = c->d.n_align > 0 ? c->d_align : nullptr; p.align_slot = c->d_align_slot + (size_t)l * H;
            p.n_align = c->d.n_align; p.align_rows = TGT;
            p.kv_div = c->beam_K > 0 ? c->beam_K : 1;
            STG(DST_CROSS_ATTN, KD(c, cw_launch_attn_decode, c->bf16, p, c->st));
            EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.bo_c; ep.ldo = D;
            STG(DST_CROSS_O, gemv_ln(c, EPI_RESID_F32, c->dattn, nb, D, L.wo_c, D, nullptr, nullptr, ep));
        }
        }
        {
            EpiParams ep = epi0(); ep.outf = c->dmid; ep.out = c->d_xfrag2; ep.bias = L.b1; ep.ldo = F;
            if (skinny) STG(DST_FC1, sk_ln(EPI_GELU_FRAG, L.w1, F, L.u1_wsum, ep));
            else if (rows) STG(DST_FC1, rows_consume(EPI_GELU_FRAG, L.w1, F, ep, L.u1_wsum));
            else STG(DST_FC1, gemv_ln(c, frag ? EPI_GELU_FRAG : EPI_GELU_F32, c->dx, nb, D, L.w1, F, L.ln2_g, c->ln_folded ? nullptr : L.ln2_b, ep));
        }
        {
            EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.b2; ep.ldo = D;
            if (rows) STG(DST_FC2, rows_produce(L.w2, F, L.b2));
            else if (frag) STG(DST_FC2, KD(c, cw_launch_gemv, true, EPI_RESID_F32, nullptr, nb, F, L.w2, D, nullptr, nullptr, ep, c->st, nullptr, c->d_xfrag2, c->wpacked));
            else STG(DST_FC2, gemv_ln(c, EPI_RESID_F32, c->dmid, nb, F, L.w2, D, nullptr, nullptr, ep));
        }
    }
#undef STG
    if (pf && c->d.dec_layers > 1) {           // join the side stream (required before the capture ends; the last prefetch is long done)
        HIPCHK(c, hipEventRecord(c->ev_pf[1], c->st2));
        HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_pf[1], 0));
    }
    if (want_logits) {   // final LN + tied proj_out (:790, :1080), logits in f32 (utils.py:2894)
        EpiParams ep = epi0(); ep.outf = c->dlogits; ep.ldo = c->Vpad;
        CWCHK(c, gemv_ln(c, EPI_STORE_F32, xin, nb, D, (c->bf16 && c->wpacked) ? c->embed_pk : c->embed, V, c->dec_ln_g, c->dec_ln_b, ep));
    }
    return CW_OK;
}

static int launch_sample(cw_ctx* c, int nb, bool forced) {
    SampleParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.logits = c->dlogits; sp.V = c->d.vocab_size; sp.ldv = c->Vpad; sp.B = nb; sp.mask = c->d_mask;
    sp.eos = c->gen.eos_token_id; sp.pad = c->gen.pad_token_id;
    sp.timestamp_begin = c->gen.no_timestamps_token_id + 1;
    sp.max_initial_timestamp_index = c->gen.max_initial_timestamp_index;
    sp.cfg = c->d_cfg; sp.pos = c->d_pos;
    sp.ids_stride = c->d.max_target_positions; sp.ids = c->d_ids; sp.forced = c->d_forced;
    sp.argmax_trace = c->d_argmax; sp.last_ts_tok = c->d_last_ts; sp.finished = c->d_finished;
    sp.n_unfinished = c->d_nunf;
    sp.embed = c->embed; sp.pos_embed = c->dec_pos; sp.x_out = c->dx; sp.d = c->d.d_model; sp.embed_bf16 = c->bf16 ? 1 : 0;
    sp.partials = c->d_sample_part;
    if (c->score_tokens) { sp.lp_sum = c->d_lp_sum; sp.lp_cnt = c->d_lp_cnt; }
    sp.epoch = c->d_epoch;
    (void)forced;
    return KD(c, cw_launch_sample, sp, c->st);
}

// decoder forward + logits + sampling for one position, captured once per batch size as a hipGraph
// (~260 launches -> one replay; every per-step value lives in device memory: d_pos, d_cfg).
static int run_step(cw_ctx* c, int nb) {
    const bool graph_ok = c->use_graph && !c->logits_capture;
    if (!graph_ok) {
        CWCHK(c, decode_step(c, nb, true));
        return launch_sample(c, nb, false);
    }
    const int gi = nb + ((c->hist_short && c->short_hist_enabled && nb > 8) ? 65 : 0);
    if (!c->step_graph[gi]) {
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
        int r = decode_step(c, nb, true);
        if (r == CW_OK) r = launch_sample(c, nb, false);
        hipError_t e = hipStreamEndCapture(c->st, &g);
        if (r != CW_OK) { if (g) hipGraphDestroy(g); return r; }
        if (e != hipSuccess) return fail(c, CW_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&c->step_graph[gi], g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) return fail(c, CW_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    HIPCHK(c, hipGraphLaunch(c->step_graph[gi], c->st));
    return CW_OK;
}

// ---- in-launch hand-offs that gave up -------------------------------------------------------------------------------------
// The kernels of declayer.hip (and the A/B kernel mlp_pair_kernel) let blocks of ONE launch wait for each other.  That needs the
// waiting blocks and the ones they wait for on the chip together -- always true when the context has the GPU to itself, not
// when other processes fill it (eight ranks on one device: tests/test_dist_gloo.py).  A block that polls DL_SPIN_LIMIT times
// without success sets d_err and carries on with garbage.  The entry points below check the flag when their work has drained
// and, if it is set, switch every such kernel off for this context (for good) and run the call again: the launch-per-stage
// kernels are bit-identical, so the caller sees the result it would have had, late.
#define CW_HANDOFF_RETRY 0x7e57
static void drop_step_graphs(cw_ctx* c);
static bool handoffs_on(const cw_ctx* c) { return c->qkv_self || c->mlp_chain || c->declayer || c->mlp_pair; }
static int handoff_gave_up(cw_ctx* c, bool* gave_up, int* word = nullptr) {
    int e = 0;
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipMemcpy(&e, c->d_err, 4, hipMemcpyDeviceToHost));
    *gave_up = e != 0;
    if (word) *word = e;
    if (e) { HIPCHK(c, hipMemset(c->d_err, 0, 4)); HIPCHK(c, hipMemset(c->d_bar, 0, 64 * 4)); }
    return CW_OK;
}
static int handoffs_off(cw_ctx* c, const char* where) {
    if (!handoffs_on(c)) return fail(c, CW_ERR_HIP, "%s: an in-launch hand-off gave up although none is enabled", where);
    c->qkv_self = c->mlp_chain = c->declayer = c->mlp_pair = false;
    drop_step_graphs(c);
    if (c->handoff_fallbacks++ == 0)
        fprintf(stderr, "crisperwhisper: %s: blocks of one launch waited for each other in vain (GPU shared with other work?); "
                        "this context continues on the launch-per-stage decoder kernels (same results)\n", where);
    return CW_OK;
}
int32_t cw_handoff_fallbacks(cw_ctx* c) { return c->handoff_fallbacks; }
int32_t cw_handoff_resumes(cw_ctx* c) { return c->handoff_resumes; }

// The granule tag is (epoch << 6) | layer in 32 bits: 26 bits of the device's forward counter.  Granules are never cleared, so a
// stale one from exactly 2^26 forwards ago (rows a smaller batch did not rewrite) would pass for valid -- about a day of continuous
// decoding.  Entry points that run decoder forwards announce an upper bound here; long before the tag can repeat the granules are
// zeroed (tag 0 = epoch 0, never used) and the counter starts again at 1.  Called with the stream idle or about to be synchronised.
static int epoch_hygiene(cw_ctx* c, long long upcoming_forwards) {
    c->forwards_since_reset += upcoming_forwards;
    if (c->forwards_since_reset < (1ll << 25)) return CW_OK;
    const int D = c->d.d_model;
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipMemset(c->d_gq, 0, (size_t)2 * 16 * D * 8)); HIPCHK(c, hipMemset(c->d_gps, 0, (size_t)(D / 16) * 16 * 2 * 8));
    HIPCHK(c, hipMemset(c->d_gq2, 0, (size_t)16 * D * 8)); HIPCHK(c, hipMemset(c->d_gkv, 0, (size_t)2 * 16 * (D / 2) * 8));
    HIPCHK(c, hipMemset(c->d_gflag, 0, (size_t)512 * 8));
    const unsigned int one = 1;
    HIPCHK(c, hipMemcpy(c->d_epoch, &one, 4, hipMemcpyHostToDevice));
    c->forwards_since_reset = upcoming_forwards;
    return CW_OK;
}

static int decode_once(cw_ctx* c, int32_t nb, const int32_t* prompt, int32_t n_prompt, int32_t max_length,
                       int32_t min_new_tokens, const int32_t* forced, int32_t* sequences, int32_t* lengths,
                       int32_t* argmax_out);
int32_t cw_decode(cw_ctx* c, int32_t nb, const int32_t* prompt, int32_t n_prompt, int32_t max_length,
                  int32_t min_new_tokens, const int32_t* forced, int32_t* sequences, int32_t* lengths,
                  int32_t* argmax_out) {
    int r = decode_once(c, nb, prompt, n_prompt, max_length, min_new_tokens, forced, sequences, lengths, argmax_out);
    if (r != CW_HANDOFF_RETRY) return r;
    CWCHK(c, handoffs_off(c, "decode"));
    r = decode_once(c, nb, prompt, n_prompt, max_length, min_new_tokens, forced, sequences, lengths, argmax_out);
    return r == CW_HANDOFF_RETRY ? fail(c, CW_ERR_HIP, "decode: an in-kernel wait timed out on the launch-per-stage path") : r;
}

static int decode_once(cw_ctx* c, int32_t nb, const int32_t* prompt, int32_t n_prompt, int32_t max_length,
                       int32_t min_new_tokens, const int32_t* forced, int32_t* sequences, int32_t* lengths,
                       int32_t* argmax_out) {
    const int D = c->d.d_model, V = c->d.vocab_size, TGT = c->d.max_target_positions;
    if (!c->gen_set) return fail(c, CW_ERR_STATE, "cw_set_generation not called");
    if (nb < 1 || nb > c->nb_encoded) return fail(c, CW_ERR_STATE, "nb=%d but %d windows encoded", nb, c->nb_encoded);
    if (n_prompt < 1 || n_prompt >= TGT) return fail(c, CW_ERR_INVALID, "n_prompt=%d out of range", n_prompt);
    if (max_length <= n_prompt || max_length > TGT) return fail(c, CW_ERR_INVALID, "max_length=%d out of range (n_prompt %d, max_target %d)", max_length, n_prompt, TGT);
    c->beam_K = 0;
    c->align_cur = c->d_align;
    CWCHK(c, epoch_hygiene(c, 2 * (long long)max_length + 4));   // (a resumed call runs some positions twice)
    std::vector<int> ids((size_t)nb * TGT, c->gen.pad_token_id);
    for (int b = 0; b < nb; ++b)
        for (int t = 0; t < n_prompt; ++t) {
            int tok = prompt[(size_t)b * n_prompt + t];
            if (tok < 0 || tok >= V) return fail(c, CW_ERR_INVALID, "prompt token %d out of range", tok);
            ids[(size_t)b * TGT + t] = tok;
        }
    HIPCHK(c, hipMemcpyAsync(c->d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->st));
    if (forced) HIPCHK(c, hipMemcpyAsync(c->d_forced, forced, (size_t)nb * TGT * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_finished, 0, nb * 4, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_lp_sum, 0, nb * 4, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_lp_cnt, 0, nb * 4, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_last_ts, 0xff, nb * 4, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_argmax, 0xff, (size_t)nb * TGT * 4, c->st));
    const int cfg[4] = {n_prompt, min_new_tokens, max_length, forced ? 1 : 0};
    HIPCHK(c, hipMemcpyAsync(c->d_cfg, cfg, sizeof(cfg), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    StageTimer tm(c, CW_STAGE_DECODE);

    // prompt positions 0 .. n_prompt-2: forward only (their alignment rows are recorded, :254-256)
    for (int pos = 0; pos + 1 < n_prompt; ++pos) {
        c->hist_short = pos + 1 <= 64;
        CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, pos, nb, c->st, c->d_epoch));
        CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, pos, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, nb, D, c->st));
        CWCHK(c, decode_step(c, nb, false));
    }
    CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, n_prompt - 1, nb, c->st, c->d_epoch));
    CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, n_prompt - 1, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, nb, D, c->st));
    int t = n_prompt, step = 0;
    // Host/device overlap: the "rows still running" counter of step s lands in its own pinned slot and is only
    // *read* after step s+1 has been queued (one step of lag, so the GPU never waits for the host; at most one
    // harmless extra step runs: it only writes cache / alignment rows beyond the ones that are used), and it
    // is not even copied while no row can finish yet (fewer than min_new_tokens generated: eos is masked).
    // The same 8-byte copy carries the hand-off failure word (d_err sits behind d_nunf): a wait that gave up (GPU shared
    // with other work) is seen one step later, and the call RESUMES at the position of the first failed forward on the
    // launch-per-stage kernels instead of running to the end on garbage and starting over.
    int first_copied = -1;
    for (;;) {
        int gave_up_word = 0;
        for (;;) {
            // forward at position t-1, logits, fused processors + argmax -> ids[t], x for position t, pos := t
            c->hist_short = t <= 64;               // keys 0 .. t-1
            CWCHK(c, run_step(c, nb));
            if (c->logits_capture && step < c->logits_capture_steps)
                HIPCHK(c, hipMemcpy2DAsync(c->logits_capture + (size_t)step * nb * V, (size_t)V * 4, c->dlogits, (size_t)c->Vpad * 4, (size_t)V * 4, nb, hipMemcpyDeviceToHost, c->st));
            ++t;                               // sequence length is now t
            const bool can_finish = (t - n_prompt) >= min_new_tokens;
            if (can_finish) {
                if (first_copied < 0) first_copied = step;
                c->h_nunf[2 * step] = 1; c->h_nunf[2 * step + 1] = 0;
                HIPCHK(c, hipMemcpyAsync(c->h_nunf + 2 * step, c->d_nunf, 8, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, hipEventRecord(c->ev_step[step & 1], c->st));
            }
            ++step;
            if (t >= max_length) break;
            if (first_copied >= 0 && step - 2 >= first_copied) {   // counter of the step before the one just queued
                HIPCHK(c, hipEventSynchronize(c->ev_step[(step - 2) & 1]));
                if (c->h_nunf[2 * (step - 2) + 1] != 0) { gave_up_word = c->h_nunf[2 * (step - 2) + 1]; break; }
                if (c->h_nunf[2 * (step - 2)] == 0) break;
            }
        }
        HIPCHK(c, hipStreamSynchronize(c->st));
        {   // a block of a launch with an in-kernel wait gave up (GPU shared with other work)
            bool gave_up = false;
            int word = 0;
            CWCHK(c, handoff_gave_up(c, &gave_up, &word));
            if (gave_up) gave_up_word = word;
        }
        if (!gave_up_word) break;
        // word = 1 + decoder position P of the first forward that ran on a missed hand-off: ids[0 .. P], the cache rows and alignment
        // rows below P are good.  Resume there when the sampler's state can be rebuilt from the ids alone (no log-probability sums, no
        // logits capture, P inside the generated part); otherwise cw_decode repeats the whole call.
        const int P = gave_up_word - 1;
        if (P < n_prompt || P + 1 >= max_length || c->score_tokens || c->logits_capture) { tm.stop(); return CW_HANDOFF_RETRY; }
        CWCHK(c, handoffs_off(c, "decode"));
        HIPCHK(c, hipMemcpy(ids.data(), c->d_ids, ids.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int> fin(nb), lts(nb);
        const int tb = c->gen.no_timestamps_token_id + 1;
        for (int b = 0; b < nb; ++b) {                                  // sample_kernel's per-row state after it wrote ids[P]
            fin[b] = 0; lts[b] = -1;
            for (int k = n_prompt; k <= P; ++k) {
                const int tok = ids[(size_t)b * TGT + k];
                if (!fin[b] && tok >= tb) lts[b] = tok;
                if (tok == c->gen.eos_token_id) fin[b] = 1;
            }
        }
        HIPCHK(c, hipMemcpy(c->d_finished, fin.data(), nb * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_last_ts, lts.data(), nb * 4, hipMemcpyHostToDevice));
        CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, P, nb, c->st, c->d_epoch));
        CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, P, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, nb, D, c->st));
        t = P + 1; step = t - n_prompt; first_copied = -1;
        ++c->handoff_resumes;
    }
    if (first_copied >= 0)                 // true end = first step after which no row was running
        for (int s2 = first_copied; s2 < step; ++s2)
            if (c->h_nunf[2 * s2] == 0) { t = n_prompt + s2 + 1; break; }
    KCHK(c);
    tm.stop();
    HIPCHK(c, hipMemcpy(ids.data(), c->d_ids, ids.size() * 4, hipMemcpyDeviceToHost));
    memcpy(sequences, ids.data(), ids.size() * 4);
    // per-row length: up to and including the first eos among generated tokens, else t
    for (int b = 0; b < nb; ++b) {
        int len = t;
        for (int k = n_prompt; k < t; ++k)
            if (ids[(size_t)b * TGT + k] == c->gen.eos_token_id) { len = k + 1; break; }
        lengths[b] = len;
    }
    if (argmax_out) HIPCHK(c, hipMemcpy(argmax_out, c->d_argmax, (size_t)nb * TGT * 4, hipMemcpyDeviceToHost));
    c->last_L = t - 1;   // attention rows retained: one per decoder input position
    c->align_unnormalized = c->bf16 && c->d.n_align > 0;
    c->last_nb = nb;
    return CW_OK;
}

// ---- deterministic half of generate_with_fallback (generation_whisper.py:970-1116, 1243-1287)
static void drop_step_graphs(cw_ctx* c) {   // (declared above cw_decode)
    for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // kernel arguments are baked into the graphs
}

int32_t cw_set_thresholds(cw_ctx* c, float logprob_threshold, float no_speech_threshold) {
    if (!isnan(no_speech_threshold) && isnan(logprob_threshold))
        return fail(c, CW_ERR_INVALID, "no_speech_threshold needs logprob_threshold (generation_whisper.py:1275-1285 compares both)");
    c->logprob_thr = logprob_threshold; c->no_speech_thr = no_speech_threshold;
    const bool want = !isnan(logprob_threshold);
    if (want != c->score_tokens) { c->score_tokens = want; drop_step_graphs(c); }
    return CW_OK;
}

// average log_softmax(processed scores)[token] over the generated tokens of every row of the last cw_decode, the eos
// included (generation_whisper.py:1958-1974); rows that generated nothing report 0
int32_t cw_get_avg_logprobs(cw_ctx* c, float* out, int32_t nb) {
    if (!c->score_tokens) return fail(c, CW_ERR_STATE, "token scores are only tracked while a logprob threshold is set (cw_set_thresholds)");
    if (nb < 1 || nb > c->last_nb) return fail(c, CW_ERR_INVALID, "nb=%d but the last decode had %d rows", nb, c->last_nb);
    std::vector<float> s(nb); std::vector<int> n(nb);
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipMemcpy(s.data(), c->d_lp_sum, nb * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(n.data(), c->d_lp_cnt, nb * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) out[b] = n[b] > 0 ? s[b] / (float)n[b] : 0.f;
    return CW_OK;
}

// WhisperNoSpeechDetection (logits_process.py:2050-2112): softmax of the raw decoder logits at the <|startoftranscript|>
// position of every encoded window, at token no_timestamps_token_id - 1 (generation_whisper.py:1801-1806).  One decoder
// forward at position 0; call it before cw_decode (which decodes position 0 again).
int32_t cw_no_speech_probs(cw_ctx* c, int32_t nb, int32_t sot_token, float* out) {
    const int D = c->d.d_model, V = c->d.vocab_size, TGT = c->d.max_target_positions;
    if (!c->gen_set) return fail(c, CW_ERR_STATE, "cw_set_generation not called");
    if (nb < 1 || nb > c->nb_encoded) return fail(c, CW_ERR_STATE, "nb=%d but %d windows encoded", nb, c->nb_encoded);
    const int tok = c->gen.no_timestamps_token_id - 1;
    if (sot_token < 0 || sot_token >= V || tok < 0) return fail(c, CW_ERR_INVALID, "no_speech_probs: token out of range");
    CWCHK(c, epoch_hygiene(c, 4));
    c->beam_K = 0;
    c->align_cur = c->d_align;
    std::vector<int> ids((size_t)nb * TGT, c->gen.pad_token_id);
    for (int b = 0; b < nb; ++b) ids[(size_t)b * TGT] = sot_token;
    HIPCHK(c, hipMemcpyAsync(c->d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    for (int attempt = 0;; ++attempt) {
        CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, 0, nb, c->st, c->d_epoch));
        CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, 0, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, nb, D, c->st));
        c->hist_short = false;
        CWCHK(c, decode_step(c, nb, true));
        KCHK(c);
        bool gave_up = false;
        CWCHK(c, handoff_gave_up(c, &gave_up));
        if (!gave_up) break;
        if (attempt) return fail(c, CW_ERR_HIP, "no_speech_probs: an in-kernel wait timed out on the launch-per-stage path");
        CWCHK(c, handoffs_off(c, "no_speech_probs"));
    }
    std::vector<float> lg((size_t)nb * V);
    CWCHK(c, cw_get_logits(c, lg.data(), nb));
    for (int b = 0; b < nb; ++b) {
        const float* r = lg.data() + (size_t)b * V;
        float m = r[0];
        for (int v = 1; v < V; ++v) m = r[v] > m ? r[v] : m;
        double z = 0.0;
        for (int v = 0; v < V; ++v) z += exp((double)(r[v] - m));
        out[b] = (float)(exp((double)(r[tok] - m)) / z);
    }
    return CW_OK;
}

int32_t cw_get_logits(cw_ctx* c, float* out, int32_t nb) {
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipMemcpy2D(out, (size_t)c->d.vocab_size * 4, c->dlogits, (size_t)c->Vpad * 4, (size_t)c->d.vocab_size * 4, nb, hipMemcpyDeviceToHost));
    return CW_OK;
}

int32_t cw_set_logits_capture(cw_ctx* c, float* host_buf, int32_t max_steps) {
    c->logits_capture = host_buf;
    c->logits_capture_steps = host_buf ? max_steps : 0;
    return CW_OK;
}

static int normalize_alignment(cw_ctx* c) {
    if (!c->align_unnormalized) return CW_OK;
    CWCHK(c, KD(c, cw_launch_align_normalize, c->d_align, c->d_align_ml, c->last_nb, c->d.n_align, c->d.max_target_positions,
                                       c->last_L, CW_N_CTX, c->st));
    c->align_unnormalized = false;
    return CW_OK;
}

int32_t cw_get_alignment(cw_ctx* c, float* out, int32_t nb, int32_t L) {
    const int Ha = c->d.n_align, TGT = c->d.max_target_positions;
    if (Ha <= 0) return fail(c, CW_ERR_STATE, "no alignment heads configured");
    if (nb > c->last_nb || L > c->last_L) return fail(c, CW_ERR_STATE, "only %d x %d rows retained", c->last_nb, c->last_L);
    CWCHK(c, normalize_alignment(c));
    HIPCHK(c, hipStreamSynchronize(c->st));
    for (int b = 0; b < nb; ++b)
        for (int a = 0; a < Ha; ++a)
            HIPCHK(c, hipMemcpy(out + (((size_t)b * Ha + a) * L) * CW_N_CTX,
                                c->align_cur + (((size_t)b * Ha + a) * TGT) * CW_N_CTX, (size_t)L * CW_N_CTX * 4,
                                hipMemcpyDeviceToHost));
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// beam search: device half of GenerationMixin._beam_search (TF/generation/utils.py:3208-3520).  The host side of the
// ABI (crisperwhisper_amd/generation.py) keeps the hypothesis bookkeeping -- running / finished beams, length penalty,
// early-stopping heuristic -- exactly as HF does; the device runs the decoder on items x beams rows, reduces every row
// to its best candidates and re-orders the per-row state by rewriting an ancestry table instead of copying KV caches.
// ------------------------------------------------------------------------------------------------
static int beam_alloc(cw_ctx* c) {
    if (c->d_anc) return CW_OK;
    const int TGT = c->d.max_target_positions, Bm = c->Bm;
    CWCHK(c, dmalloc(c, &c->d_anc, (size_t)Bm * TGT * 4)); CWCHK(c, dmalloc(c, &c->d_anc_tmp, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_ids_tmp, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_parent, Bm * 4)); CWCHK(c, dmalloc(c, &c->d_tok, Bm * 4));
    CWCHK(c, dmalloc(c, &c->d_cand_val, (size_t)Bm * 64 * 4)); CWCHK(c, dmalloc(c, &c->d_cand_id, (size_t)Bm * 64 * 4));
    CWCHK(c, dmalloc(c, &c->d_rowmap, (size_t)Bm * TGT * 4));
    CWCHK(c, dmalloc(c, &c->d_topk_scratch, KD(c, cw_beam_topk_scratch_floats, Bm) * 4));
    return CW_OK;
}

int32_t cw_beam_begin(cw_ctx* c, int32_t n_items, int32_t num_beams, const int32_t* prompt, int32_t n_prompt,
                      int32_t max_length, int32_t min_new_tokens) {
    const int D = c->d.d_model, V = c->d.vocab_size, TGT = c->d.max_target_positions;
    if (!c->gen_set) return fail(c, CW_ERR_STATE, "cw_set_generation not called");
    if (num_beams < 1 || n_items < 1 || n_items > c->nb_encoded) return fail(c, CW_ERR_STATE, "beam_begin: %d items but %d windows encoded", n_items, c->nb_encoded);
    const int rows = n_items * num_beams;
    if (rows > c->Bm) return fail(c, CW_ERR_INVALID, "beam search needs max_batch >= items x beams = %d (context has %d)", rows, c->Bm);
    if (n_prompt < 1 || n_prompt >= TGT || max_length <= n_prompt || max_length > TGT) return fail(c, CW_ERR_INVALID, "beam_begin: n_prompt=%d max_length=%d out of range", n_prompt, max_length);
    CWCHK(c, beam_alloc(c));
    CWCHK(c, epoch_hygiene(c, 2 * (long long)max_length + 4));
    std::vector<int> ids((size_t)rows * TGT, c->gen.pad_token_id), anc((size_t)rows * TGT);
    for (int r = 0; r < rows; ++r) {
        for (int t = 0; t < n_prompt; ++t) {
            const int tok = prompt[(size_t)(r / num_beams) * n_prompt + t];
            if (tok < 0 || tok >= V) return fail(c, CW_ERR_INVALID, "prompt token %d out of range", tok);
            ids[(size_t)r * TGT + t] = tok;
        }
        for (int t = 0; t < TGT; ++t) anc[(size_t)r * TGT + t] = r;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(c->d_anc, anc.data(), anc.size() * 4, hipMemcpyHostToDevice, c->st));
    const int cfg[4] = {n_prompt, min_new_tokens, max_length, 0};
    HIPCHK(c, hipMemcpyAsync(c->d_cfg, cfg, sizeof(cfg), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    c->beam_K = num_beams; c->beam_items = n_items; c->beam_n_prompt = n_prompt;
    c->beam_pos = n_prompt - 1; c->beam_max_len = max_length;
    c->align_cur = c->d_align;
    for (int pos = 0; pos + 1 < n_prompt; ++pos) {          // prompt positions: forward only
        CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, pos, rows, c->st));
        CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, pos, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, rows, D, c->st));
        c->hist_short = pos + 1 <= 64;
        CWCHK(c, decode_step(c, rows, false));
    }
    CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, n_prompt - 1, rows, c->st));
    CWCHK(c, KD(c, cw_launch_embed, c->d_ids, TGT, n_prompt - 1, c->embed, c->bf16 ? 1 : 0, c->dec_pos, c->dx, rows, D, c->st));
    KCHK(c);
    c->last_nb = rows; c->last_L = n_prompt - 1;
    return CW_OK;
}

int32_t cw_beam_step(cw_ctx* c, int32_t n_cand, float* cand_logprob, int32_t* cand_token) {
    if (c->beam_K <= 0) return fail(c, CW_ERR_STATE, "cw_beam_begin not called");
    if (n_cand < 1 || n_cand > 64) return fail(c, CW_ERR_INVALID, "n_cand=%d out of range", n_cand);
    // the token chosen from this step sits at index beam_pos + 1: one step too many would write cache and alignment rows
    // beyond the limit the context was sized for
    if (c->beam_pos + 1 >= c->beam_max_len) return fail(c, CW_ERR_STATE, "beam_step: sequence already has max_length=%d tokens", c->beam_max_len);
    const int rows = c->beam_items * c->beam_K;
    StageTimer tm(c, CW_STAGE_DECODE);
    c->hist_short = c->beam_pos + 1 <= 64;   // keys 0 .. beam_pos
    CWCHK(c, decode_step(c, rows, true));
    SampleParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.logits = c->dlogits; sp.V = c->d.vocab_size; sp.ldv = c->Vpad; sp.B = rows; sp.mask = c->d_mask;
    sp.eos = c->gen.eos_token_id; sp.pad = c->gen.pad_token_id;
    sp.timestamp_begin = c->gen.no_timestamps_token_id + 1;
    sp.max_initial_timestamp_index = c->gen.max_initial_timestamp_index;
    sp.cfg = c->d_cfg; sp.pos = c->d_pos; sp.ids_stride = c->d.max_target_positions; sp.ids = c->d_ids;
    sp.embed_bf16 = c->bf16 ? 1 : 0;
    CWCHK(c, KD(c, cw_launch_beam_topk, sp, n_cand, c->d_cand_val, c->d_cand_id, c->d_topk_scratch, c->st));
    KCHK(c);
    tm.stop();
    HIPCHK(c, hipMemcpy(cand_logprob, c->d_cand_val, (size_t)rows * n_cand * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cand_token, c->d_cand_id, (size_t)rows * n_cand * 4, hipMemcpyDeviceToHost));
    c->last_L += 1;                       // one more decoder input position has its alignment rows
    c->align_unnormalized = c->bf16 && c->d.n_align > 0;
    return CW_OK;
}

int32_t cw_beam_advance(cw_ctx* c, const int32_t* parent, const int32_t* token) {
    if (c->beam_K <= 0) return fail(c, CW_ERR_STATE, "cw_beam_begin not called");
    const int rows = c->beam_items * c->beam_K, V = c->d.vocab_size;
    for (int r = 0; r < rows; ++r) {
        if (parent[r] < 0 || parent[r] >= rows || parent[r] / c->beam_K != r / c->beam_K) return fail(c, CW_ERR_INVALID, "beam_advance: row %d cannot descend from row %d", r, parent[r]);
        if (token[r] < 0 || token[r] >= V) return fail(c, CW_ERR_INVALID, "beam_advance: token %d out of range", token[r]);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_parent, parent, rows * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(c->d_tok, token, rows * 4, hipMemcpyHostToDevice, c->st));
    BeamAdvanceParams p;
    memset(&p, 0, sizeof(p));
    p.ids = c->d_ids; p.ids_tmp = c->d_ids_tmp; p.ids_stride = c->d.max_target_positions;
    p.anc = c->d_anc; p.anc_tmp = c->d_anc_tmp; p.cap = c->d.max_target_positions;
    p.parent = c->d_parent; p.token = c->d_tok; p.pos = c->d_pos;
    p.embed = c->embed; p.pos_embed = c->dec_pos; p.x_out = c->dx; p.d = c->d.d_model; p.embed_bf16 = c->bf16 ? 1 : 0;
    p.rows = rows;
    CWCHK(c, KD(c, cw_launch_beam_advance, p, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));                   // parent / token are caller-owned host buffers
    c->beam_pos += 1;
    return CW_OK;
}

int32_t cw_beam_finish(cw_ctx* c, int32_t n_items, int32_t L, const int32_t* row_of_pos) {
    if (c->beam_K <= 0) return fail(c, CW_ERR_STATE, "cw_beam_begin not called");
    const int rows = c->beam_items * c->beam_K, TGT = c->d.max_target_positions, Ha = c->d.n_align;
    if (n_items != c->beam_items || L < 0 || L > c->last_L) return fail(c, CW_ERR_INVALID, "beam_finish: %d items x %d rows requested, %d x %d decoded", n_items, L, c->beam_items, c->last_L);
    for (size_t i = 0; i < (size_t)n_items * L; ++i)
        if (row_of_pos[i] < 0 || row_of_pos[i] >= rows) return fail(c, CW_ERR_INVALID, "beam_finish: row %d out of range", row_of_pos[i]);
    if (Ha > 0 && L > 0) {
        if (!c->d_align_g) CWCHK(c, dmalloc(c, &c->d_align_g, (size_t)c->Bm * Ha * TGT * CW_N_CTX * 4, false));
        c->last_nb = rows;
        CWCHK(c, normalize_alignment(c));
        HIPCHK(c, hipMemcpyAsync(c->d_rowmap, row_of_pos, (size_t)n_items * L * 4, hipMemcpyHostToDevice, c->st));
        CWCHK(c, KD(c, cw_launch_align_gather, c->d_align, c->d_rowmap, n_items, Ha, TGT, L, CW_N_CTX, c->d_align_g, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        c->align_cur = c->d_align_g;
    }
    c->last_nb = n_items; c->last_L = L;
    c->beam_K = 0;                                            // back to one row per item
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// token timestamps
// ------------------------------------------------------------------------------------------------
static int run_alignment(cw_ctx* c, const float* w, int B, int Ha, int rows_cap, int S, int row0, int N,
                         const int* d_ncols, int width, float* mean, float* stdv, float* mat) {
    CWCHK(c, cw_launch_align_stats(w, B, Ha, rows_cap, S, row0, N, d_ncols, mean, stdv, c->st));
    CWCHK(c, cw_launch_align_filter(w, B, Ha, rows_cap, S, row0, N, d_ncols, mean, stdv, width, mat, c->st));
    return CW_OK;
}

int32_t cw_token_timestamps(cw_ctx* c, int32_t nb, int32_t L, int32_t n_prompt, const int32_t* num_frames, float* ts_out) {
    const int Ha = c->d.n_align, TGT = c->d.max_target_positions, S = CW_N_CTX;
    if (Ha <= 0) return fail(c, CW_ERR_STATE, "no alignment heads configured");
    if (nb < 1 || nb > c->last_nb || L != c->last_L) return fail(c, CW_ERR_STATE, "rows retained: %d x %d, asked %d x %d", c->last_nb, c->last_L, nb, L);
    for (size_t i = 0; i < (size_t)nb * (L + 1); ++i) ts_out[i] = 0.f;
    const int N = L - n_prompt;
    if (N <= 0) return CW_OK;                     // generation_whisper.py:336-338
    // Columns reaching the DTW, with HF's exact slicing semantics (:318-323 + :354): python-style
    // `[..., : nf // 2]`, applied twice when every item has the same num_frames, once otherwise.
    // nf <= 0 happens when the seek loop ran past the valid audio; it may leave zero columns.
    std::vector<int> ncols(nb);
    bool uniform = true;
    for (int b = 1; b < nb; ++b) uniform = uniform && (num_frames[b] == num_frames[0]);
    auto floordiv2 = [](int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); };
    auto slice_len = [](int n, int h) { return h >= 0 ? (h < n ? h : n) : (n + h > 0 ? n + h : 0); };
    bool any_cols = false;
    for (int b = 0; b < nb; ++b) {
        int h = floordiv2(num_frames[b]);
        int n1 = slice_len(S, h);
        ncols[b] = uniform ? slice_len(n1, h) : n1;
        any_cols = any_cols || ncols[b] > 0;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ncols, ncols.data(), nb * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    StageTimer tm(c, CW_STAGE_TIMESTAMPS);
    CWCHK(c, normalize_alignment(c));
    CWCHK(c, run_alignment(c, c->align_cur ? c->align_cur : c->d_align, nb, Ha, TGT, S, n_prompt, N, c->d_ncols, c->d.median_filter_width, c->d_mean, c->d_std, c->d_mat));
    {   // workspace of the wave-local DTW, grown on demand (diagonal-major copy of the cost matrices)
        const size_t need = cw_dtw_skew_floats(nb, N, S);
        if (need > c->skew_cap) {
            // grown at most a few times: sized for the worst case of this batch size (N = max_target_positions) and the
            // previous buffer is released, so a token count that creeps upwards cannot pile up dead workspaces
            const size_t want = cw_dtw_skew_floats(nb, c->d.max_target_positions, S) > need ? cw_dtw_skew_floats(nb, c->d.max_target_positions, S) : need;
            if (c->d_skew) {
                HIPCHK(c, hipStreamSynchronize(c->st));
                for (auto it = c->allocs.begin(); it != c->allocs.end(); ++it)
                    if (*it == (void*)c->d_skew) { c->allocs.erase(it); break; }
                hipFree(c->d_skew);
                c->d_skew = nullptr; c->skew_cap = 0;
            }
            CWCHK(c, dmalloc(c, &c->d_skew, want * 4, false));
            c->skew_cap = want;
        }
    }
    CWCHK(c, cw_launch_dtw(c->d_mat, nb, N, S, c->d_ncols, c->d_trace, c->d_first_col, c->dtw_block ? c->d_path_text : nullptr, c->dtw_block ? c->d_path_time : nullptr, c->d_path_len, c->st, c->dtw_block ? nullptr : c->d_skew));
    KCHK(c);
    tm.stop();
    std::vector<int> fc((size_t)nb * N);
    HIPCHK(c, hipMemcpy(fc.data(), c->d_first_col, fc.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) {
        float* ts = ts_out + (size_t)b * (L + 1);
        for (int i = 0; i < N; ++i) {
            // zero columns: the reference's backtrace walks up column 0 -> time index -1 for every token
            int col = ncols[b] > 0 ? fc[(size_t)b * N + i] : -1;
            ts[n_prompt + i] = (float)((double)col * 0.02);                                          // :369
        }
        ts[L] = ts[L - 1];                                                                          // :377-379
    }
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// native seek loop
// ------------------------------------------------------------------------------------------------
int32_t cw_transcribe(cw_ctx* c, int32_t B, const int32_t* num_frames, const cw_transcribe_cfg* cfg, int32_t* tokens,
                      float* token_ts, int32_t* lens, int32_t cap, int32_t* n_passes) {
    const int TGT = c->d.max_target_positions, V = c->d.vocab_size;
    if (B < 1 || B > c->Bm) return fail(c, CW_ERR_INVALID, "B=%d out of range (max_batch %d)", B, c->Bm);
    if (!c->gen_set) return fail(c, CW_ERR_STATE, "cw_set_generation not called");
    const int tb = c->gen.no_timestamps_token_id + 1;
    const int eos = c->gen.eos_token_id, pad = c->gen.pad_token_id;
    std::vector<std::vector<int>> prompts(B);
    std::vector<int> all(B), zeros(B, 0), full(B, CW_N_FRAMES);
    for (int i = 0; i < B; ++i) all[i] = i;
    bool pre_encoded = false;
    if (cfg->language_token >= 0) {
        for (int i = 0; i < B; ++i) {
            prompts[i] = {cfg->sot_token, cfg->language_token};
            if (cfg->task_token >= 0) prompts[i].push_back(cfg->task_token);
        }
    } else {   // detect_language (:1610-1673): one decoder step on <|startoftranscript|>, argmax over the language ids
        if (!cfg->lang_ids || cfg->n_lang_ids <= 0) return fail(c, CW_ERR_INVALID, "language detection needs lang_ids");
        CWCHK(c, cw_encode(c, B, all.data(), zeros.data(), full.data()));
        std::vector<int> sotp(B, cfg->sot_token), seq((size_t)B * TGT), ln(B);
        CWCHK(c, cw_decode(c, B, sotp.data(), 1, 2, 0, nullptr, seq.data(), ln.data(), nullptr));
        std::vector<float> lg((size_t)B * V);
        CWCHK(c, cw_get_logits(c, lg.data(), B));
        for (int i = 0; i < B; ++i) {
            int best = cfg->lang_ids[0];
            for (int k = 1; k < cfg->n_lang_ids; ++k) {
                const int id = cfg->lang_ids[k];
                const float a = lg[(size_t)i * V + id], bv = lg[(size_t)i * V + best];
                if (a > bv || (a == bv && id < best)) best = id;
            }
            prompts[i] = {cfg->sot_token, best};
            if (cfg->task_token >= 0) prompts[i].push_back(cfg->task_token);
        }
        pre_encoded = true;
    }
    const int n_prompt = (int)prompts[0].size();
    int max_new = cfg->max_new_tokens;
    if (max_new >= 0 && max_new + n_prompt > TGT) max_new = TGT - n_prompt;            // :1937-1942
    const int max_length = max_new >= 0 ? n_prompt + max_new : (cfg->max_length < TGT ? cfg->max_length : TGT);
    std::vector<long> seek(B, 0);
    std::vector<std::vector<int>> out_tok(B);
    std::vector<std::vector<float>> out_ts(B);
    int passes = 0;
    for (;;) {
        std::vector<int> active, a_seek, a_n, a_nf;
        for (int i = 0; i < B; ++i)
            if (seek[i] < CW_N_FRAMES) {                                               // _maybe_reduce_batch
                active.push_back(i); a_seek.push_back((int)seek[i]);
                a_n.push_back((int)(CW_N_FRAMES - seek[i]));
                a_nf.push_back((int)(num_frames[i] - seek[i]));
            }
        if (active.empty()) break;
        const int nb = (int)active.size();
        if (!(pre_encoded && passes == 0)) CWCHK(c, cw_encode(c, nb, active.data(), a_seek.data(), a_n.data()));
        std::vector<int> prm((size_t)nb * n_prompt), seq((size_t)nb * TGT), ln(nb);
        for (int r = 0; r < nb; ++r) for (int k = 0; k < n_prompt; ++k) prm[(size_t)r * n_prompt + k] = prompts[active[r]][k];
        // deterministic half of generate_with_fallback: a window with a low average log-probability AND a high no-speech
        // probability is skipped (seek moves on by the whole window, no segment; :879-881, :1275-1285)
        const bool thr = !isnan(c->logprob_thr) && !isnan(c->no_speech_thr);
        std::vector<float> nsp(nb, 0.f), alp(nb, 0.f);
        if (thr) CWCHK(c, cw_no_speech_probs(c, nb, cfg->sot_token, nsp.data()));
        CWCHK(c, cw_decode(c, nb, prm.data(), n_prompt, max_length, cfg->min_new_tokens, nullptr, seq.data(), ln.data(), nullptr));
        if (thr) CWCHK(c, cw_get_avg_logprobs(c, alp.data(), nb));
        int total = 0;
        for (int r = 0; r < nb; ++r) total = ln[r] > total ? ln[r] : total;
        const int L = total - 1;
        std::vector<float> ts((size_t)nb * (L + 1));
        CWCHK(c, cw_token_timestamps(c, nb, L, n_prompt, a_nf.data(), ts.data()));
        ++passes;
        for (int r = 0; r < nb; ++r) {
            const int i = active[r];
            if (thr && alp[r] < c->logprob_thr && nsp[r] > c->no_speech_thr) { seek[i] += a_n[r]; continue; }
            const int* s = seq.data() + (size_t)r * TGT + n_prompt;
            int n = total - n_prompt;
            if (n > 0 && s[n - 1] == pad) {                                            // strip right padding (:1060-1067)
                int npad = 0;
                for (int k = 0; k < n; ++k) npad += (s[k] == pad);
                if (pad == eos) npad -= 1;
                n -= npad;
            }
            if (n > 0 && s[n - 1] == eos) n -= 1;                                      // :1081-1082
            // _retrieve_segment: keep everything up to the last *pair* of timestamp tokens, or the whole window
            const double time_offset = (double)seek[i] * 0.02 / 2.0;
            const float off32 = (float)time_offset;
            const bool single_ending = n >= 2 && s[n - 2] < tb && s[n - 1] >= tb;
            int last_pair_end = -1;                                                    // index after the pair's 1st token
            for (int k = 0; k + 1 < n; ++k) if (s[k] >= tb && s[k + 1] >= tb) last_pair_end = k + 1;
            int keep, advance;
            if (last_pair_end >= 0) {
                if (single_ending) { keep = n; advance = a_n[r]; }
                else { keep = last_pair_end + 1; advance = (s[last_pair_end - 1] - tb) * 2; }
            } else { keep = n; advance = a_n[r]; }
            for (int k = 0; k < keep; ++k) {
                out_tok[i].push_back(s[k]);
                out_ts[i].push_back(ts[(size_t)r * (L + 1) + n_prompt + k] + off32);
            }
            seek[i] += advance;
        }
    }
    for (int i = 0; i < B; ++i) {
        const int n = (int)out_tok[i].size();
        if (n > cap) return fail(c, CW_ERR_INVALID, "item %d produced %d tokens, capacity %d", i, n, cap);
        lens[i] = n;
        memcpy(tokens + (size_t)i * cap, out_tok[i].data(), (size_t)n * 4);
        memcpy(token_ts + (size_t)i * cap, out_ts[i].data(), (size_t)n * 4);
    }
    if (n_passes) *n_passes = passes;
    return CW_OK;
}

int32_t cw_align_matrix(cw_ctx* c, const float* attn, int32_t B, int32_t Ha, int32_t N, int32_t M, const int32_t* n_cols,
                        int32_t width, float* mat_out) {
    float *dw = nullptr, *dmean = nullptr, *dstd = nullptr, *dmat = nullptr; int* dn = nullptr;
    size_t nw = (size_t)B * Ha * N * M;
    HIPCHK(c, hipMalloc((void**)&dw, nw * 4)); HIPCHK(c, hipMalloc((void**)&dmean, (size_t)B * Ha * M * 4));
    HIPCHK(c, hipMalloc((void**)&dstd, (size_t)B * Ha * M * 4)); HIPCHK(c, hipMalloc((void**)&dmat, (size_t)B * N * M * 4));
    HIPCHK(c, hipMalloc((void**)&dn, B * 4));
    HIPCHK(c, hipMemcpy(dw, attn, nw * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dn, n_cols, B * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(dmat, 0, (size_t)B * N * M * 4));
    int r = run_alignment(c, dw, B, Ha, N, M, 0, N, dn, width, dmean, dstd, dmat);
    if (r == CW_OK) {
        hipError_t e = hipStreamSynchronize(c->st);
        if (e == hipSuccess) e = hipMemcpy(mat_out, dmat, (size_t)B * N * M * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = fail(c, CW_ERR_HIP, "align_matrix: %s", hipGetErrorString(e));
    }
    hipFree(dw); hipFree(dmean); hipFree(dstd); hipFree(dmat); hipFree(dn);
    return r;
}

int32_t cw_dtw(cw_ctx* c, const float* mat, int32_t N, int32_t M, int32_t* text_idx, int32_t* time_idx, int32_t* path_len) {
    if (N < 1 || N > 512 || M < 1) return fail(c, CW_ERR_INVALID, "dtw: N=%d M=%d unsupported", N, M);
    float* dmat = nullptr; unsigned char* dtr = nullptr; int *dfc = nullptr, *dpt = nullptr, *dpj = nullptr, *dpl = nullptr, *dn = nullptr;
    HIPCHK(c, hipMalloc((void**)&dmat, (size_t)N * M * 4)); HIPCHK(c, hipMalloc((void**)&dtr, (size_t)N * M));
    HIPCHK(c, hipMalloc((void**)&dfc, N * 4)); HIPCHK(c, hipMalloc((void**)&dpt, (size_t)(N + M + 2) * 4));
    HIPCHK(c, hipMalloc((void**)&dpj, (size_t)(N + M + 2) * 4)); HIPCHK(c, hipMalloc((void**)&dpl, 4)); HIPCHK(c, hipMalloc((void**)&dn, 4));
    HIPCHK(c, hipMemcpy(dmat, mat, (size_t)N * M * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dn, &M, 4, hipMemcpyHostToDevice));
    float* dskew = nullptr;
    HIPCHK(c, hipMalloc((void**)&dskew, cw_dtw_skew_floats(1, N, M) * 4));
    int r = cw_launch_dtw(dmat, 1, N, M, dn, dtr, dfc, dpt, dpj, dpl, c->st, c->dtw_block ? nullptr : dskew);
    if (r == CW_OK) {
        std::vector<int> pt(N + M + 2), pj(N + M + 2);
        int n = 0;
        hipError_t e = hipStreamSynchronize(c->st);
        if (e == hipSuccess) e = hipMemcpy(&n, dpl, 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(pt.data(), dpt, pt.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(pj.data(), dpj, pj.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = fail(c, CW_ERR_HIP, "dtw: %s", hipGetErrorString(e));
        else {
            for (int k = 0; k < n; ++k) { text_idx[k] = pt[n - 1 - k]; time_idx[k] = pj[n - 1 - k]; }
            *path_len = n;
        }
    } else {
        fail(c, r, "dtw launch rejected N=%d", N);
    }
    hipFree(dmat); hipFree(dtr); hipFree(dfc); hipFree(dpt); hipFree(dpj); hipFree(dpl); hipFree(dn); hipFree(dskew);
    return r;
}

int32_t cw_adjust_pauses(cw_ctx* c, double* start, double* end, int32_t W, double thr) {
    if (W <= 0) return CW_OK;
    if (W > c->pause_cap) {   // grow the persistent scratch (start/end in, start/end out)
        int cap = 1024;
        while (cap < W) cap *= 2;
        CWCHK(c, dmalloc(c, &c->d_pause, (size_t)4 * cap * 8, false));
        c->pause_cap = cap;
    }
    double* ds = c->d_pause;              // [in W | out W]
    double* de = c->d_pause + 2 * (size_t)c->pause_cap;
    HIPCHK(c, hipMemcpyAsync(ds, start, (size_t)W * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(de, end, (size_t)W * 8, hipMemcpyHostToDevice, c->st));
    CWCHK(c, cw_launch_pauses(ds, de, W, thr, c->st));
    HIPCHK(c, hipMemcpyAsync(start, ds + W, (size_t)W * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(end, de + W, (size_t)W * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// audio ingest
// ------------------------------------------------------------------------------------------------
static int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int64_t cw_resampled_length(int64_t n, int32_t sr_in, int32_t sr_out) {
    if (n < 0 || sr_in <= 0 || sr_out <= 0) return -1;
    const int g = igcd(sr_in, sr_out);
    const int64_t o = sr_in / g, w = sr_out / g;
    return (n * w + o - 1) / o;
}

// torchaudio's _get_sinc_resample_kernel with dtype=None (f64 arithmetic, rounded to f32 at the end): K[p][j],
// p in 0..new-1, j in 0..2*width+orig-1
// false: the rate pair needs an unreasonably large polyphase table (coprime rates in the MHz range, corrupted headers)
static bool resample_taps(int sr_in, int sr_out, std::vector<float>& K, int& orig, int& nw, int& width) {
    const int g = igcd(sr_in, sr_out);
    orig = sr_in / g; nw = sr_out / g;
    const double lpw = 6.0, rolloff = 0.99;
    const double base_freq = (double)(orig < nw ? orig : nw) * rolloff;
    const double w = ceil(lpw * orig / base_freq);
    if (w > 1e6 || ((double)nw * (2.0 * w + orig)) > 536870912.0) return false;   // 512 M taps = 2 GB
    width = (int)w;
    const int n_taps = 2 * width + orig;
    K.assign((size_t)nw * n_taps, 0.f);
    const double scale = base_freq / orig;
    for (int p = 0; p < nw; ++p)
        for (int j = 0; j < n_taps; ++j) {
            double t = (double)(-p) / nw + (double)(j - width) / orig;
            t *= base_freq;
            t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
            const double c = cos(t * M_PI / lpw / 2.0);
            const double window = c * c;
            t *= M_PI;
            const double sinc = t == 0.0 ? 1.0 : sin(t) / t;
            K[(size_t)p * n_taps + j] = (float)(sinc * window * scale);
        }
    return true;
}

int32_t cw_resample_taps(int32_t sr_in, int32_t sr_out, float* taps, int32_t cap, int32_t* orig, int32_t* nw,
                         int32_t* width) {
    if (sr_in <= 0 || sr_out <= 0) return CW_ERR_INVALID;
    std::vector<float> K;
    int o = 0, w = 0, wd = 0;
    if (!resample_taps(sr_in, sr_out, K, o, w, wd)) return CW_ERR_INVALID;
    if (orig) *orig = o;
    if (nw) *nw = w;
    if (width) *width = wd;
    if (taps) {
        if ((size_t)cap < K.size()) return CW_ERR_INVALID;
        memcpy(taps, K.data(), K.size() * 4);
    }
    return CW_OK;
}

int32_t cw_ingest(cw_ctx* c, const void* raw, int32_t fmt, int32_t channels, int64_t n_frames, int32_t sr_in,
                  int32_t sr_out, int32_t normalise, float* out) {
    static const int bytes_of[] = {1, 2, 3, 4, 4, 8};
    if (fmt < 0 || fmt > CW_PCM_F64) return fail(c, CW_ERR_INVALID, "unknown sample format %d", fmt);
    if (channels < 1 || n_frames < 1 || sr_in <= 0 || sr_out <= 0)
        return fail(c, CW_ERR_INVALID, "bad audio geometry (channels=%d frames=%lld %d->%d Hz)", channels, (long long)n_frames, sr_in, sr_out);
    const size_t raw_bytes = (size_t)n_frames * channels * bytes_of[fmt];
    const int64_t n_out = cw_resampled_length(n_frames, sr_in, sr_out);
    void* d_raw = nullptr; float *d_mono = nullptr, *d_out = nullptr, *d_taps = nullptr; double* d_acc = nullptr;
    int rc = CW_OK;
    auto cleanup = [&]() { hipFree(d_raw); hipFree(d_mono); hipFree(d_out); hipFree(d_taps); hipFree(d_acc); };
#define ING(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(c, CW_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    ING(hipMalloc(&d_raw, raw_bytes));
    ING(hipMalloc((void**)&d_mono, (size_t)n_frames * 4));
    ING(hipMemcpyAsync(d_raw, raw, raw_bytes, hipMemcpyHostToDevice, c->st));
    rc = cw_launch_pcm_to_mono(d_raw, fmt, channels, n_frames, d_mono, c->st);
    if (rc == CW_OK && normalise) {
        ING(hipMalloc((void**)&d_acc, 16));
        rc = cw_launch_normalise(d_mono, n_frames, d_acc, c->st);
    }
    const float* d_res = d_mono;
    if (rc == CW_OK && sr_in != sr_out) {                       // F.resample returns its input when the rates agree
        std::vector<float> K, Kt;
        int orig = 0, nw = 0, width = 0;
        if (!resample_taps(sr_in, sr_out, K, orig, nw, width)) {
            cleanup();
            return fail(c, CW_ERR_INVALID, "unsupported sampling-rate pair %d -> %d Hz (polyphase table too large)", sr_in, sr_out);
        }
        const int n_taps = 2 * width + orig;
        Kt.resize(K.size());
        for (int p = 0; p < nw; ++p) for (int j = 0; j < n_taps; ++j) Kt[(size_t)j * nw + p] = K[(size_t)p * n_taps + j];
        ING(hipMalloc((void**)&d_taps, Kt.size() * 4));
        ING(hipMalloc((void**)&d_out, (size_t)n_out * 4));
        ING(hipMemcpyAsync(d_taps, Kt.data(), Kt.size() * 4, hipMemcpyHostToDevice, c->st));
        ING(hipStreamSynchronize(c->st));                       // Kt is a local
        rc = cw_launch_resample(d_mono, n_frames, d_taps, orig, nw, width, n_out, d_out, c->st);
        d_res = d_out;
    }
    if (rc != CW_OK) { cleanup(); return fail(c, rc, "cw_ingest: launch rejected"); }
    ING(hipMemcpyAsync(out, d_res, (size_t)n_out * 4, hipMemcpyDeviceToHost, c->st));
    ING(hipStreamSynchronize(c->st));
#undef ING
    cleanup();
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// kernel-level test hooks
// ------------------------------------------------------------------------------------------------
// e4m3 copies of the encoder qkv / fc1 / fc2 and decoder cross-K/V weights from the resident 16-bit weights (row-wise scales)
static int enc8_quantise(cw_ctx* c) {
    const int D = c->d.d_model, F = c->d.ffn_dim;
    for (auto& L : c->enc) {
        CWCHK(c, KD(c, cw_launch_quant_rows_fp8, L.wqkv, 3 * D, D, L.wqkv8, L.sqkv, c->st));
        CWCHK(c, KD(c, cw_launch_quant_rows_fp8, L.w1, F, D, L.w18, L.s1, c->st));
        CWCHK(c, KD(c, cw_launch_quant_rows_fp8, L.w2, D, F, L.w28, L.s2, c->st));
    }
    for (auto& L : c->dec) CWCHK(c, KD(c, cw_launch_quant_rows_fp8, L.wkv_c, 2 * D, D, L.wkv_c8, L.skv_c, c->st));
    KCHK(c);
    c->enc8_stale = false;
    return CW_OK;
}

int32_t cw_set_option(cw_ctx* c, const char* name, int32_t value) {
    if (!name) return fail(c, CW_ERR_INVALID, "cw_set_option: null option name");
    if (!strcmp(name, "cross_kv_fp8")) {
        if (!value) { c->kv8 = false; return CW_OK; }
        if (!c->bf16) return fail(c, CW_ERR_INVALID, "cross_kv_fp8 needs the bf16 engine (the f32 engine is the parity mode)");
        if (!c->dec[0].ck8) {
            const size_t n = (size_t)c->Bm * c->d.n_heads * CW_N_CTX * 64;
            const size_t nv = (size_t)c->Bm * cw_bf16::cw_kv8_v_bytes(c->d.n_heads, CW_N_CTX);   // V: fragment-major for the fp8 matrix cores
            for (auto& L : c->dec) {
                CWCHK(c, dmalloc(c, &L.ck8, n, false)); CWCHK(c, dmalloc(c, &L.cv8, nv, false));
                CWCHK(c, dmalloc(c, &L.kvs, (size_t)c->Bm * c->d.n_heads * 2 * 4));
            }
        }
        c->kv8 = true;
        c->nb_encoded = 0;                                   // windows must be re-encoded to fill the fp8 cache
        for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // graphs hold the kernel choice
        return CW_OK;
    }
    if (!strcmp(name, "encoder_gemm_fp8")) {
        if (!value) { c->enc8 = false; c->enc8_mask = 0; c->nb_encoded = 0; return CW_OK; }
        if (!c->bf16) return fail(c, CW_ERR_INVALID, "encoder_gemm_fp8 needs a 16-bit engine (the f32 engine is the parity mode)");
        const int D = c->d.d_model, F = c->d.ffn_dim;
        if (D % 256 != 0 || F % 256 != 0 || D > 2048) return fail(c, CW_ERR_INVALID, "encoder_gemm_fp8: d_model / ffn_dim must be multiples of 256 (d_model <= 2048)");
        CWCHK(c, cw_check_weights(c));                       // the e4m3 copies are made from the resident 16-bit weights
        if (!c->h8) {
            const size_t MR = (size_t)c->Bm * CW_N_CTX;
            CWCHK(c, dmalloc(c, &c->h8, MR * D, false)); CWCHK(c, dmalloc(c, &c->mid8, MR * F, false));
            CWCHK(c, dmalloc(c, &c->sa8, MR * 4)); CWCHK(c, dmalloc(c, &c->smid8, MR * 4));
            for (auto& L : c->enc) {
                CWCHK(c, dmalloc(c, &L.wqkv8, (size_t)3 * D * D, false)); CWCHK(c, dmalloc(c, &L.sqkv, (size_t)3 * D * 4));
                CWCHK(c, dmalloc(c, &L.w18, (size_t)F * D, false)); CWCHK(c, dmalloc(c, &L.s1, (size_t)F * 4));
                CWCHK(c, dmalloc(c, &L.w28, (size_t)D * F, false)); CWCHK(c, dmalloc(c, &L.s2, (size_t)D * 4));
            }
            for (auto& L : c->dec) { CWCHK(c, dmalloc(c, &L.wkv_c8, (size_t)2 * D * D, false)); CWCHK(c, dmalloc(c, &L.skv_c, (size_t)2 * D * 4)); }
        }
        CWCHK(c, enc8_quantise(c));
        c->enc8 = true;
        c->enc8_mask = value >= 16 ? ((value - 16) & 15) : 15;   // 1 = every GEMM; 16 + mask = a subset (tools/fp8_sweep.py)
        c->nb_encoded = 0;                                   // windows must be re-encoded
        return CW_OK;
    }
#ifndef CW_EXPERIMENTS
    if ((!strcmp(name, "rows_ln") || !strcmp(name, "skinny")) && value != 0)
        return fail(c, CW_ERR_INVALID, "option %s selects a measured-and-rejected kernel variant that is not in this build (make EXTRA=-DCW_EXPERIMENTS)", name);
#endif
    if (!strcmp(name, "test_epoch_forwards")) {   // test hook: pretend this many decoder forwards have run since the granule epoch last started (epoch_hygiene)
        c->forwards_since_reset = (long long)value;
        return CW_OK;
    }
    if (!strcmp(name, "handoff_fail_pos")) {   // test hook: the in-launch hand-off of qkv_self_kernel gives up at this decoder position (-1: never)
        c->fail_pos = value;
        for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // kernel arguments are baked into the graphs
        return CW_OK;
    }
    if (!strcmp(name, "rows_ln")) {   // 17..64-row decode: 0 = preparation launch in front of every GEMV (A/B, differential tests)
        c->rows_ln_enabled = value != 0;
        for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // graphs hold the kernel choice
        return CW_OK;
    }
    if (!strcmp(name, "skinny")) {    // 17..64-row decode: cw_ctx::skinny_mode (0 = round-3 path, for A/B and differential tests)
        c->skinny_mode = value;
        for (auto& ge : c->step_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }
        return CW_OK;
    }
    return fail(c, CW_ERR_INVALID, "unknown option %s", name);
}

// 1 when the library was built with -DCW_EXPERIMENTS (the measured-and-rejected kernel variants and their A/B switches:
// DESIGN.md "A/B switches"); the differential tests of those variants skip otherwise.
int32_t cw_has_experiments(void) {
#ifdef CW_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

static int g_test_cross_fp8 = 0;   // cw_test_cross_attention through the e4m3 cache (option "cross_test_fp8")
int32_t cw_test_set_option(const char* name, int32_t value) {
    if (!strcmp(name, "cross_test_fp8")) { g_test_cross_fp8 = value; return CW_OK; }
    if (!strcmp(name, "gemm256_min_tiles")) { cw_bf16::cw_gemm_set_256_min_tiles(value); cw_f16::cw_gemm_set_256_min_tiles(value); return CW_OK; }
    if (!strcmp(name, "beam_topk_1block")) { cw_bf16::cw_beam_topk_set_1block(value); cw_f16::cw_beam_topk_set_1block(value); return CW_OK; }
    if (!strcmp(name, "gemm_pp")) { cw_bf16::cw_gemm_set_pp(value); cw_f16::cw_gemm_set_pp(value); return CW_OK; }
    if (!strcmp(name, "gemm_8ph")) { cw_bf16::cw_gemm_set_8ph(value); cw_f16::cw_gemm_set_8ph(value); return CW_OK; }
    if (!strcmp(name, "gemm_gm")) {
#ifndef CW_EXPERIMENTS
        return CW_ERR_INVALID;
#endif
        cw_bf16::cw_gemm_set_gm(value); cw_f16::cw_gemm_set_gm(value); return CW_OK;
    }
    if (!strcmp(name, "gemv_loop")) { cw_bf16::cw_gemv_set_loop(value); cw_f16::cw_gemv_set_loop(value); return CW_OK; }
    if (!strcmp(name, "mt_variant")) { cw_bf16::cw_gemv_set_mt_variant(value); cw_f16::cw_gemv_set_mt_variant(value); return CW_OK; }
    if (!strcmp(name, "comb_rowgroups")) { cw_bf16::cw_gemv_set_comb_rowgroups(value); cw_f16::cw_gemv_set_comb_rowgroups(value); return CW_OK; }
    if (!strcmp(name, "gemm_w128")) { cw_bf16::cw_gemm_set_w128(value); cw_f16::cw_gemm_set_w128(value); return CW_OK; }
    if (!strcmp(name, "cross_valu")) { cw_bf16::cw_cross_set_valu(value); cw_f16::cw_cross_set_valu(value); return CW_OK; }
    if (!strcmp(name, "cross_per_row")) { cw_bf16::cw_cross_set_per_row(value); cw_f16::cw_cross_set_per_row(value); return CW_OK; }
    return CW_ERR_INVALID;
}

int32_t cw_test_gemm(cw_ctx* c, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                     int32_t gelu, float* out) {
    void *dA = nullptr, *dW = nullptr, *dO = nullptr; float* dB = nullptr;
    const size_t e = c->esz;
    HIPCHK(c, hipMalloc(&dA, (size_t)M * K * e)); HIPCHK(c, hipMalloc(&dW, (size_t)N * K * e));
    HIPCHK(c, hipMalloc(&dO, (size_t)M * N * e)); HIPCHK(c, hipMalloc((void**)&dB, (size_t)N * 4));
    CWCHK(c, upload_T(c, dA, 0, A, (size_t)M * K)); CWCHK(c, upload_T(c, dW, 0, W, (size_t)N * K));
    if (bias) HIPCHK(c, hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    AParams ap{dA, K, 0, 0, 0, 0, nullptr, nullptr};
    EpiParams ep = epi0(); ep.out = dO; ep.bias = bias ? dB : nullptr; ep.ldo = N;
    int r = KD(c, cw_launch_gemm, c->bf16, gelu ? EPI_GELU : EPI_STORE, ap, dW, M, N, K, ep, c->st);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_gemm: %s", hipGetErrorString(er)); }
    if (const int reps = cw_sw::cw_switches().test_gemm_reps) {   // kernel A/B timing for the profiles (stderr only)
        hipEvent_t e0, e1;
        if (r == CW_OK && reps > 0 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            hipEventRecord(e0, c->st);
            for (int i = 0; i < reps && r == CW_OK; ++i) r = KD(c, cw_launch_gemm, c->bf16, gelu ? EPI_GELU : EPI_STORE, ap, dW, M, N, K, ep, c->st);
            hipEventRecord(e1, c->st);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / reps;
            fprintf(stderr, "[cw_test_gemm] M=%d N=%d K=%d gelu=%d: %.2f us/launch, %.1f TFLOP/s\n", M, N, K, gelu, us, 2.0 * M * N * K / us * 1e-6);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    if (r == CW_OK) r = download_T(c, dO, 0, out, (size_t)M * N);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dB);
    return r;
}

// e4m3 GEMM of the opt-in encoder mode: A and W are rounded to the engine's 16-bit type, quantised row-wise on the device
// (quant_rows_fp8_kernel) and multiplied by gemm_fp8_pp_kernel; out = T(A W^T + bias) (gelu optional).  N % 256 == 0, K % 128 == 0.
int32_t cw_test_gemm_fp8(cw_ctx* c, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                         int32_t gelu, float* out) {
    if (!c->bf16) return fail(c, CW_ERR_INVALID, "the fp8 GEMM belongs to the 16-bit engines");
    void *dA = nullptr, *dW = nullptr, *dO = nullptr, *dA8 = nullptr, *dW8 = nullptr; float *dB = nullptr, *dsa = nullptr, *dsw = nullptr;
    const size_t e = c->esz;
    HIPCHK(c, hipMalloc(&dA, (size_t)M * K * e)); HIPCHK(c, hipMalloc(&dW, (size_t)N * K * e));
    HIPCHK(c, hipMalloc(&dA8, (size_t)M * K)); HIPCHK(c, hipMalloc(&dW8, (size_t)N * K));
    HIPCHK(c, hipMalloc((void**)&dsa, (size_t)M * 4)); HIPCHK(c, hipMalloc((void**)&dsw, (size_t)N * 4));
    HIPCHK(c, hipMalloc(&dO, (size_t)M * N * e)); HIPCHK(c, hipMalloc((void**)&dB, (size_t)N * 4));
    CWCHK(c, upload_T(c, dA, 0, A, (size_t)M * K)); CWCHK(c, upload_T(c, dW, 0, W, (size_t)N * K));
    if (bias) HIPCHK(c, hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    EpiParams ep = epi0(); ep.out = dO; ep.bias = bias ? dB : nullptr; ep.ldo = N;
    int r = KD(c, cw_launch_quant_rows_fp8, dA, M, K, dA8, dsa, c->st);
    if (r == CW_OK) r = KD(c, cw_launch_quant_rows_fp8, dW, N, K, dW8, dsw, c->st);
    if (r == CW_OK) r = KD(c, cw_launch_gemm_fp8, gelu ? EPI_GELU : EPI_STORE, dA8, K, dW8, M, N, K, dsa, dsw, ep, c->st);
    if (r != CW_OK) fail(c, r, "test_gemm_fp8: launch rejected (M=%d N=%d K=%d)", M, N, K);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_gemm_fp8: %s", hipGetErrorString(er)); }
    if (const int reps = cw_sw::cw_switches().test_gemm_reps) {   // kernel timing for the profiles (stderr only)
        hipEvent_t e0, e1;
        if (r == CW_OK && reps > 0 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            hipEventRecord(e0, c->st);
            for (int i = 0; i < reps && r == CW_OK; ++i) r = KD(c, cw_launch_gemm_fp8, gelu ? EPI_GELU : EPI_STORE, dA8, K, dW8, M, N, K, dsa, dsw, ep, c->st);
            hipEventRecord(e1, c->st);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / reps;
            fprintf(stderr, "[cw_test_gemm_fp8] M=%d N=%d K=%d gelu=%d: %.2f us/launch, %.1f TFLOP/s\n", M, N, K, gelu, us, 2.0 * M * N * K / us * 1e-6);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    if (r == CW_OK) r = download_T(c, dO, 0, out, (size_t)M * N);
    hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dB); hipFree(dA8); hipFree(dW8); hipFree(dsa); hipFree(dsw);
    return r;
}

int32_t cw_test_gemv(cw_ctx* c, int32_t Mb, int32_t N, int32_t K, const float* x, const float* W, const float* bias,
                     const float* ln_g, const float* ln_b, int32_t gelu, float* out) {
    float *dx = nullptr, *dB = nullptr, *dO = nullptr, *dg = nullptr, *db = nullptr, *dxn = nullptr; void* dW = nullptr;
    HIPCHK(c, hipMalloc((void**)&dx, (size_t)Mb * K * 4)); HIPCHK(c, hipMalloc(&dW, (size_t)N * K * c->esz));
    HIPCHK(c, hipMalloc((void**)&dO, (size_t)Mb * N * 4)); HIPCHK(c, hipMalloc((void**)&dB, (size_t)N * 4));
    HIPCHK(c, hipMalloc((void**)&dg, (size_t)K * 4)); HIPCHK(c, hipMalloc((void**)&db, (size_t)K * 4));
    HIPCHK(c, hipMalloc((void**)&dxn, (size_t)Mb * K * 4));
    HIPCHK(c, hipMemcpy(dx, x, (size_t)Mb * K * 4, hipMemcpyHostToDevice));
    CWCHK(c, upload_T(c, dW, 0, W, (size_t)N * K));
    if (bias) HIPCHK(c, hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    if (ln_g) { HIPCHK(c, hipMemcpy(dg, ln_g, (size_t)K * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(db, ln_b, (size_t)K * 4, hipMemcpyHostToDevice)); }
    EpiParams ep = epi0(); ep.outf = dO; ep.bias = bias ? dB : nullptr; ep.ldo = N;
    int r = CW_OK;
    const float* xin = dx;
    if (ln_g && !c->bf16) { r = KD(c, cw_launch_layernorm_f32, dx, dg, db, dxn, Mb, K, c->st); xin = dxn; }
    if (r == CW_OK) r = KD(c, cw_launch_gemv, c->bf16, gelu ? EPI_GELU_F32 : EPI_STORE_F32, xin, Mb, K, dW, N,
                                        (ln_g && c->bf16) ? dg : nullptr, (ln_g && c->bf16) ? db : nullptr, ep, c->st, nullptr, c->d_xfrag);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_gemv: %s", hipGetErrorString(er)); }
    else fail(c, r, "test_gemv: launch rejected (Mb=%d N=%d K=%d)", Mb, N, K);
    if (r == CW_OK) { hipError_t er = hipMemcpy(out, dO, (size_t)Mb * N * 4, hipMemcpyDeviceToHost); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_gemv copy"); }
    hipFree(dx); hipFree(dW); hipFree(dO); hipFree(dB); hipFree(dg); hipFree(db); hipFree(dxn);
    return r;
}

// One skinny-M decoder projection (skinny.hip) on caller-supplied rows, 16-bit engines only.
//   mode 0: out[Mb][N] = n(x) W^T + bias, n = LayerNorm without affine part (folded weights), through K-split planes + finish
//   mode 1: the same through the GELU epilogue (16-bit fragment-major on the device, returned row-major as f32)
//   mode 2: out[Mb][N] (in/out, on the 2^-12 grid) += x16 W^T + bias, x16 = x rounded to the engine's 16-bit type
// nks: k-steps of 32 per block (0: the launcher's own choice).  reps > 0: the launches are also timed over `reps` rounds on
// rotating weight copies larger than the Infinity Cache (cold weights, as in a decoder pass); us[0] = GEMM, us[1] = finish.
int32_t cw_test_skinny(cw_ctx* c, int32_t mode, int32_t Mb, int32_t N, int32_t K, const float* x, const float* W, const float* bias,
                       int32_t nks, int32_t reps, float* out, float* us) {
#ifndef CW_EXPERIMENTS
    return fail(c, CW_ERR_INVALID, "cw_test_skinny: csrc/skinny.hip is an A/B build (make EXTRA=-DCW_EXPERIMENTS)");
#endif
    if (!c->bf16) return fail(c, CW_ERR_INVALID, "cw_test_skinny: 16-bit engines only");
    const bool slice_stats = (mode & 4) != 0;   // LayerNorm statistics from the GEMM's slice records instead of re-reading x
    mode &= 3;
    if (Mb < 1 || Mb > 64 || K % 32 || N % 16 || mode < 0 || mode > 2 || (mode == 1 && N % 32)) return fail(c, CW_ERR_INVALID, "cw_test_skinny: bad shape");
    const int MT = (Mb + 15) / 16;
    if (nks <= 0) nks = KD(c, cw_skinny_pick_nks, N, K, mode == 2 ? 0 : 16);
    if (nks < 1 || (K / 32) % nks) return fail(c, CW_ERR_INVALID, "cw_test_skinny: nks %d does not divide K / 32", nks);
    const int S = (K / 32) / nks;
    const size_t wbytes = (size_t)N * K * 2;
    int ncopy = 1;
    if (reps > 0) { ncopy = (int)((size_t)320 * 1024 * 1024 / wbytes) + 1; if (ncopy > 64) ncopy = 64; }
    DevScope mem;
    float *dx = nullptr, *dB = nullptr, *dO = nullptr, *dws = nullptr, *dP = nullptr, *dst = nullptr; void *dW = nullptr, *dWp = nullptr, *dxf = nullptr, *dfr = nullptr;
    HIPCHK(c, mem.get(&dst, (size_t)64 * 64 * 2 * 4));
    HIPCHK(c, mem.get(&dx, (size_t)Mb * K * 4)); HIPCHK(c, mem.get(&dW, wbytes)); HIPCHK(c, mem.get(&dWp, wbytes * ncopy));
    HIPCHK(c, mem.get(&dO, (size_t)Mb * N * 4)); HIPCHK(c, mem.get(&dB, (size_t)N * 4)); HIPCHK(c, mem.get(&dws, (size_t)N * 4));
    HIPCHK(c, mem.get(&dP, (size_t)S * Mb * N * 4)); HIPCHK(c, mem.get(&dxf, (size_t)MT * 16 * K * 2)); HIPCHK(c, mem.get(&dfr, (size_t)MT * 16 * N * 2));
    HIPCHK(c, hipMemcpy(dx, x, (size_t)Mb * K * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(dB, 0, (size_t)N * 4));
    if (bias) HIPCHK(c, hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    CWCHK(c, upload_T(c, dW, 0, W, (size_t)N * K));
    CWCHK(c, KD(c, cw_launch_fold_rowvec, nullptr, nullptr, 1.0f, nullptr, dW, N, K, nullptr, dws, c->st));
    for (int i = 0; i < ncopy; ++i) CWCHK(c, KD(c, cw_launch_wfrag_pack, dW, N, K, (char*)dWp + wbytes * i, c->st));
    if (mode == 2) {
        std::vector<bf16_t> xf((size_t)MT * 16 * K, 0);
        for (int m = 0; m < Mb; ++m)
            for (int k = 0; k < K; ++k) xf[frag_index(m, k, K)] = c->f16 ? cw_host_f32_to_f16(x[(size_t)m * K + k]) : cw_host_f32_to_bf16(x[(size_t)m * K + k]);
        HIPCHK(c, hipMemcpy(dxf, xf.data(), xf.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dO, out, (size_t)Mb * N * 4, hipMemcpyHostToDevice));
    }
    SkinnyParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.x = dx; sp.xf = dxf; sp.W = dWp; sp.Mb = Mb; sp.K = K; sp.N = N; sp.planes = dP; sp.outf = dO; sp.bias = dB; sp.ldo = N;
    if (reps > 0 && getenv("CW_SK_DBG")) sp.dbg = atoi(getenv("CW_SK_DBG"));   // ablation of the timed launches (-DCW_SK_DEBUG builds; tools/skinny_ablate.py changes it between calls, so not in cw_switches)
    const int sk_dbg = sp.dbg;
    sp.dbg = 0;
    SkinnyFinishParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.planes = dP; fp.S = S; fp.Mb = Mb; fp.N = N; fp.x = dx; fp.K = K; fp.wsum = dws; fp.ep = epi0();
    fp.ep.bias = dB; fp.ep.outf = dO; fp.ep.out = dfr; fp.ep.ldo = N;
    if (slice_stats && mode != 2) {
        if (S > 16) { return fail(c, CW_ERR_INVALID, "cw_test_skinny: slice statistics need S <= 16"); }
        sp.stats = dst; fp.stats = dst;
    }
    const int epi = mode == 1 ? EPI_GELU_FRAG : EPI_STORE_F32;
    int r = KD(c, cw_launch_skinny, mode == 2 ? 1 : 0, sp, nks, c->st);
    if (r == CW_OK && mode != 2) r = KD(c, cw_launch_skinny_finish, epi, fp, c->st);
    if (r != CW_OK) fail(c, r, "cw_test_skinny: launch rejected (mode=%d Mb=%d N=%d K=%d nks=%d)", mode, Mb, N, K, nks);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "cw_test_skinny: %s", hipGetErrorString(er)); }
    if (r == CW_OK) { hipError_t er = hipGetLastError(); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "cw_test_skinny: %s", hipGetErrorString(er)); }
    if (r == CW_OK && mode == 1) {
        std::vector<bf16_t> fr((size_t)MT * 16 * N);
        if (hipMemcpy(fr.data(), dfr, fr.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, CW_ERR_HIP, "cw_test_skinny copy");
        else for (int m = 0; m < Mb; ++m)
            for (int n = 0; n < N; ++n) { const bf16_t v = fr[frag_index(m, n, N)]; out[(size_t)m * N + n] = c->f16 ? cw_host_f16_to_f32(v) : cw_host_bf16_to_f32(v); }
    } else if (r == CW_OK) {
        if (hipMemcpy(out, dO, (size_t)Mb * N * 4, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, CW_ERR_HIP, "cw_test_skinny copy");
    }
    if (r == CW_OK && reps > 0 && us) {
        // `reps` dependent launches captured as one hipGraph (how the decode step runs them) and replayed; the last replay is timed
        sp.dbg = sk_dbg;
        auto time_graph = [&](const std::function<int(int)>& launch, float* us_out) -> int {
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            if (hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal) != hipSuccess) return CW_ERR_HIP;
            int rr = CW_OK;
            for (int i = 0; i < reps && rr == CW_OK; ++i) rr = launch(i);
            hipError_t e = hipStreamEndCapture(c->st, &g);
            if (rr == CW_OK && e != hipSuccess) rr = CW_ERR_HIP;
            if (rr == CW_OK && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) rr = CW_ERR_HIP;
            if (g) hipGraphDestroy(g);
            if (rr != CW_OK) return rr;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e30f;
            for (int k = 0; k < 4; ++k) {
                hipEventRecord(e0, c->st);
                hipGraphLaunch(ge, c->st);
                hipEventRecord(e1, c->st);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (k > 0 && ms < best) best = ms;
            }
            hipEventDestroy(e0); hipEventDestroy(e1); hipGraphExecDestroy(ge);
            *us_out = 1e3f * best / reps;
            return CW_OK;
        };
        r = time_graph([&](int i) { sp.W = (char*)dWp + wbytes * (i % ncopy); return KD(c, cw_launch_skinny, mode == 2 ? 1 : 0, sp, nks, c->st); }, &us[0]);
        if (r == CW_OK && mode != 2) r = time_graph([&](int) { return KD(c, cw_launch_skinny_finish, epi, fp, c->st); }, &us[1]);
        if (r == CW_OK && mode == 2) {                       // no finish launch: the floor of an empty launch instead (negative)
            r = time_graph([&](int) { KD(c, cw_launch_skinny_empty, c->st); return (int)CW_OK; }, &us[1]);
            us[1] = -us[1];
        }
        if (r != CW_OK) fail(c, r, "cw_test_skinny: timing graph failed");
    }
    return r;
}

int32_t cw_test_attention(cw_ctx* c, int32_t B, int32_t H, int32_t S, const float* q, const float* k, const float* v, float* out) {
    const int S_pad = (S + 63) & ~63;
    const size_t e = c->esz, nh = (size_t)B * H * S_pad * 64;
    DevScope mem;
    void *dq = nullptr, *dk = nullptr, *dv = nullptr, *dout = nullptr;
    HIPCHK(c, mem.get(&dq, nh * e)); HIPCHK(c, mem.get(&dk, nh * e)); HIPCHK(c, mem.get(&dv, nh * e));
    HIPCHK(c, mem.get(&dout, (size_t)B * S * H * 64 * e));
    HIPCHK(c, hipMemset(dq, 0, nh * e)); HIPCHK(c, hipMemset(dk, 0, nh * e)); HIPCHK(c, hipMemset(dv, 0, nh * e));
    for (int bh = 0; bh < B * H; ++bh) {   // inputs are [B][H][S][64]
        CWCHK(c, upload_T(c, dq, (size_t)bh * S_pad * 64, q + (size_t)bh * S * 64, (size_t)S * 64));
        CWCHK(c, upload_T(c, dk, (size_t)bh * S_pad * 64, k + (size_t)bh * S * 64, (size_t)S * 64));
        CWCHK(c, upload_T(c, dv, (size_t)bh * S_pad * 64, v + (size_t)bh * S * 64, (size_t)S * 64));
    }
    int r = KD(c, cw_launch_attn_encoder, c->bf16, dq, dk, dv, dout, B, H, S, S_pad, c->st);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_attention: %s", hipGetErrorString(er)); }
    if (const int reps = cw_sw::cw_switches().test_attn_reps) {   // kernel A/B timing for the profiles (stderr only)
        hipEvent_t e0, e1;
        if (r == CW_OK && reps > 0 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            hipEventRecord(e0, c->st);
            for (int i = 0; i < reps && r == CW_OK; ++i) r = KD(c, cw_launch_attn_encoder, c->bf16, dq, dk, dv, dout, B, H, S, S_pad, c->st);
            hipEventRecord(e1, c->st);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / reps, fl = 4.0 * B * H * (double)S * S * 64;
            fprintf(stderr, "[cw_test_attention] B=%d H=%d S=%d: %.2f us/launch, %.1f TFLOP/s\n", B, H, S, us, fl / us * 1e-6);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    if (r == CW_OK) r = download_T(c, dout, 0, out, (size_t)B * S * H * 64);
    return r;
}

// One launch of the key-split cross-attention decode kernel on caller-supplied rows: q [B][H*64] (already scaled), k / v
// [B / kv_div][H][S][64].  part_o [ATT_NS][B][H*64] and part_ml [B][H][ATT_NS][2] come back raw (the consumer's combine is the
// test's); head `align_head` is captured as alignment slot 0: align [B][S] un-normalised + align_ml [B][ATT_NS][2].
int32_t cw_test_cross_attention(cw_ctx* c, int32_t B, int32_t H, int32_t S, int32_t kv_div, const float* q, const float* k,
                                const float* v, int32_t align_head, float* part_o, float* part_ml, float* align, float* align_ml) {
    if (B < 1 || H < 1 || S < 1 || kv_div < 1 || B % kv_div || align_head < 0 || align_head >= H) return fail(c, CW_ERR_INVALID, "test_cross_attention: shape");
    const int Bk = B / kv_div, D = H * 64;
    const size_t nkv = (size_t)Bk * H * S * 64;
    float *dq = nullptr, *dpo = nullptr, *dml = nullptr, *dal = nullptr, *daml = nullptr; void *dk = nullptr, *dv = nullptr;
    int *dslot = nullptr, *dpos = nullptr;
    DevScope mem;
    HIPCHK(c, mem.get(&dq, (size_t)B * D * 4)); HIPCHK(c, mem.get(&dk, nkv * c->esz)); HIPCHK(c, mem.get(&dv, nkv * c->esz));
    HIPCHK(c, mem.get(&dpo, (size_t)ATT_NS * B * D * 4)); HIPCHK(c, mem.get(&dml, (size_t)B * H * ATT_NS * 2 * 4));
    HIPCHK(c, mem.get(&dal, (size_t)B * S * 4)); HIPCHK(c, mem.get(&daml, (size_t)B * ATT_NS * 2 * 4));
    HIPCHK(c, mem.get(&dslot, (size_t)H * 4)); HIPCHK(c, mem.get(&dpos, (size_t)B * 4));
    std::vector<int> slot(H, -1); slot[align_head] = 0;
    HIPCHK(c, hipMemcpy(dslot, slot.data(), (size_t)H * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(dpos, 0, (size_t)B * 4)); HIPCHK(c, hipMemset(dal, 0, (size_t)B * S * 4));
    HIPCHK(c, hipMemcpy(dq, q, (size_t)B * D * 4, hipMemcpyHostToDevice));
    CWCHK(c, upload_T(c, dk, 0, k, nkv)); CWCHK(c, upload_T(c, dv, 0, v, nkv));
    CrossSplitParams p{};
    p.q = dq; p.K = dk; p.V = dv; p.n_keys = S; p.part_o = dpo; p.part_ml = dml; p.align_out = dal; p.align_ml = daml;
    p.align_slot = dslot; p.pos = dpos; p.n_align = 1; p.align_rows = 1; p.B = B; p.H = H; p.kv_div = kv_div;
    // option "cross_test_fp8": the rows go through the e4m3 cache instead (quantiser + the fp8 cross-attention kernel)
    void *dk8 = nullptr, *dv8 = nullptr; float* dkvs = nullptr;
    const bool fp8 = g_test_cross_fp8 && c->bf16;
    if (fp8) {
        HIPCHK(c, mem.get(&dk8, nkv)); HIPCHK(c, mem.get(&dv8, (size_t)Bk * cw_bf16::cw_kv8_v_bytes(H, S)));
        HIPCHK(c, mem.get(&dkvs, (size_t)Bk * H * 2 * 4));
        int rq = KD(c, cw_launch_kv_quant_fp8, dk, dv, dk8, dv8, dkvs, Bk, H, S, c->st);
        if (rq != CW_OK) { return fail(c, rq, "test_cross_attention: quantiser rejected S=%d", S); }
        p.K = dk8; p.V = dv8; p.kv_scale = dkvs;
    }
    auto launch = [&]() { return fp8 ? KD(c, cw_launch_attn_cross_split_fp8, p, c->st) : KD(c, cw_launch_attn_cross_split, c->bf16, p, c->st); };
    int r = launch();
    if (r != CW_OK) fail(c, r, "test_cross_attention: launch rejected (B=%d H=%d S=%d kv_div=%d)", B, H, S, kv_div);
    if (r == CW_OK) { hipError_t er = hipStreamSynchronize(c->st); if (er != hipSuccess) r = fail(c, CW_ERR_HIP, "test_cross_attention: %s", hipGetErrorString(er)); }
    if (const int reps = cw_sw::cw_switches().test_attn_reps) {   // kernel A/B timing for the profiles (stderr only)
        hipEvent_t e0, e1;
        if (r == CW_OK && reps > 0 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            hipEventRecord(e0, c->st);
            for (int i = 0; i < reps && r == CW_OK; ++i) r = launch();
            hipEventRecord(e1, c->st);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            fprintf(stderr, "[cw_test_cross_attention] B=%d H=%d S=%d kv_div=%d: %.2f us/launch\n", B, H, S, kv_div, 1e3 * ms / reps);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    if (r == CW_OK && (hipMemcpy(part_o, dpo, (size_t)ATT_NS * B * D * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(part_ml, dml, (size_t)B * H * ATT_NS * 2 * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(align, dal, (size_t)B * S * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(align_ml, daml, (size_t)B * ATT_NS * 2 * 4, hipMemcpyDeviceToHost) != hipSuccess))
        r = fail(c, CW_ERR_HIP, "test_cross_attention copy");
    return r;
}

// One call of the fused logits-processor + argmax kernel (sample_kernel) on caller-supplied rows: logits [nb][V],
// ids [nb][t] = prompt + generated so far (the kernel's grammar state is rebuilt from it), choice_out [nb] = the
// token the kernel picks for index t.  Differential test against TF/generation/logits_process.py:203-260, 1816-2047.
int32_t cw_test_sample(cw_ctx* c, int32_t nb, const float* logits, const int32_t* ids, int32_t t, int32_t n_prompt,
                       int32_t min_new_tokens, int32_t max_length, int32_t* choice_out) {
    const int V = c->d.vocab_size, TGT = c->d.max_target_positions;
    if (!c->gen_set) return fail(c, CW_ERR_STATE, "cw_set_generation not called");
    if (nb < 1 || nb > c->Bm || t < n_prompt || t < 1 || t >= TGT || n_prompt < 1) return fail(c, CW_ERR_INVALID, "test_sample: bad args");
    const int tb = c->gen.no_timestamps_token_id + 1;
    std::vector<int> hid((size_t)nb * TGT, c->gen.pad_token_id), last(nb, -1), posv(64, t - 1);
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < t; ++k) {
            const int tok = ids[(size_t)b * t + k];
            if (tok < 0 || tok >= V) return fail(c, CW_ERR_INVALID, "test_sample: token %d out of range", tok);
            hid[(size_t)b * TGT + k] = tok;
            if (k >= n_prompt && tok >= tb) last[b] = tok;
        }
    HIPCHK(c, hipMemcpy(c->d_ids, hid.data(), hid.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_last_ts, last.data(), nb * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_pos, posv.data(), 64 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemset(c->d_finished, 0, nb * 4));
    HIPCHK(c, hipMemset(c->d_argmax, 0xff, (size_t)nb * TGT * 4));
    const int cfg[4] = {n_prompt, min_new_tokens, max_length, 0};
    HIPCHK(c, hipMemcpy(c->d_cfg, cfg, sizeof(cfg), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy2D(c->dlogits, (size_t)c->Vpad * 4, logits, (size_t)V * 4, (size_t)V * 4, nb, hipMemcpyHostToDevice));
    CWCHK(c, launch_sample(c, nb, false));
    HIPCHK(c, hipStreamSynchronize(c->st));
    std::vector<int> am((size_t)nb * TGT);
    HIPCHK(c, hipMemcpy(am.data(), c->d_argmax, am.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < nb; ++b) choice_out[b] = am[(size_t)b * TGT + t];
    return CW_OK;
}

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
int32_t cw_stage_times(cw_ctx* c, float* ms, int32_t* calls, int32_t reset) {
    for (int i = 0; i < CW_N_STAGES; ++i) {
        if (ms) ms[i] = c->stage_ms[i];
        if (calls) calls[i] = c->stage_calls[i];
        if (reset) { c->stage_ms[i] = 0.f; c->stage_calls[i] = 0; }
    }
    return CW_OK;
}

int32_t cw_time_kernel(cw_ctx* c, int32_t which, int32_t nb, int32_t iters, float* avg_ms, double* algo_bytes) {
    if (iters > 0) CWCHK(c, epoch_hygiene(c, (long long)iters + 8));
    if (which == 100 || which == 101) {
        // launch-boundary floor: a captured graph of `iters` dependent near-empty kernels (100), or the same
        // launched eagerly (101); avg_ms = time per kernel
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (which == 100) {
            HIPCHK(c, hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < iters; ++i) KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st);
            HIPCHK(c, hipStreamEndCapture(c->st, &g));
            HIPCHK(c, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIPCHK(c, hipGraphLaunch(ge, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
        }
        HIPCHK(c, hipEventRecord(c->ev0, c->st));
        if (which == 100) { for (int r = 0; r < 5; ++r) HIPCHK(c, hipGraphLaunch(ge, c->st)); }
        else for (int r = 0; r < 5; ++r) for (int i = 0; i < iters; ++i) KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st);
        HIPCHK(c, hipEventRecord(c->ev1, c->st));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *avg_ms = ms / (5.0f * iters);
        *algo_bytes = 0;
        if (ge) hipGraphExecDestroy(ge);
        if (g) hipGraphDestroy(g);
        return CW_OK;
    }
    const int D = c->d.d_model, H = c->d.n_heads, F = c->d.ffn_dim;
    if (nb < 1 || nb > c->Bm || iters < 1) return fail(c, CW_ERR_INVALID, "time_kernel: bad args");
    const int TGT = c->d.max_target_positions, V = c->d.vocab_size;
    // launches cycle through the decoder layers, as the real step does: every launch streams its weights / K,V cache
    // from HBM (one layer's 61 MB cross cache alone would otherwise sit in the 256 MB Infinity Cache and flatter the
    // kernel: 11.6 us "hot" vs 14.1 us inside the step)
    int launch_no = 0;
    const bool hot = which >= 200;                              // 200 + k: kernel k on ONE layer's operands (L2 / Infinity-Cache resident)
    if (hot) which -= 200;
    auto launch = [&]() -> int {
        LayerW& L = c->dec[hot ? 0 : (launch_no++) % c->d.dec_layers];
        switch (which) {
            case 0: {   // fc1: LN + GEMV + GELU
                EpiParams ep = epi0(); ep.outf = c->dmid; ep.bias = L.b1; ep.ldo = F;
                return gemv_ln(c, EPI_GELU_F32, c->dx, nb, D, L.w1, F, L.ln2_g, c->ln_folded ? nullptr : L.ln2_b, ep);
            }
            case 1: {   // cross-attention (the variant the decode step launches: with the fused stage in front it finishes the query)
                if (c->bf16) {
                    CrossSplitParams p{c->dq, L.ck, L.cv, CW_N_CTX, c->d_part_o, c->d_part_ml, nullptr, c->d_align_ml, nullptr,
                                       c->d_pos, 0, 0, nb, H};
                    if (c->fuse6_ready && c->fuse6_enabled && c->ln_folded && nb <= 16 && (!c->kv8 || KD(c, cw_cross8_is_mfma, CW_N_CTX))) {
                        p.kv_div = 1; p.qa = c->d_qa; p.qb = c->d_qb; p.qw = L.q_wsum; p.qbias = L.bq_c;
                        p.pstats = c->d_pstats; p.n_pstats = D / 16;
                        if (c->kv8) { p.K = L.ck8; p.V = L.cv8; p.kv_scale = L.kvs; return KD(c, cw_launch_attn_cross_split_fp8, p, c->st); }
                        return KD(c, cw_launch_attn_cross_split, true, p, c->st);
                    }
                    if (c->kv8) { p.K = L.ck8; p.V = L.cv8; p.kv_scale = L.kvs; return KD(c, cw_launch_attn_cross_split_fp8, p, c->st); }
                    return KD(c, cw_launch_attn_cross_split, true, p, c->st);
                }
                DecAttnParams p = dec_attn(c->dq, L.ck, L.cv, CW_N_CTX, CW_N_CTX, nullptr, c->dattn, nb, H);
                return KD(c, cw_launch_attn_decode, c->bf16, p, c->st);
            }
            case 2: {   // self-attention out projection (in-place residual, K split)
                EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.bo; ep.ldo = D;
                return gemv_ln(c, EPI_RESID_F32, c->dattn, nb, D, L.wo, D, nullptr, nullptr, ep);
            }
            case 3: {   // LN + qkv + cache append
                EpiParams ep = epi0(); ep.outf = c->dq; ep.out1 = L.sk; ep.out2 = L.sv; ep.bias = L.bqkv;
                ep.H = H; ep.S_pad = TGT; ep.d_model = D; ep.row_pos = c->d_pos;
                return gemv_ln(c, EPI_QKV_CACHE, c->dx, nb, D, L.wqkv, 3 * D, L.ln1_g, c->ln_folded ? nullptr : L.ln1_b, ep);
            }
            case 4: {   // LN + cross q
                EpiParams ep = epi0(); ep.outf = c->dq; ep.bias = L.bq_c; ep.ldo = D;
                return gemv_ln(c, EPI_STORE_F32, c->dx, nb, D, L.wq_c, D, L.lnc_g, c->ln_folded ? nullptr : L.lnc_b, ep);
            }
            case 5: {   // fc2
                EpiParams ep = epi0(); ep.outf = c->dx; ep.resid = c->dx; ep.bias = L.b2; ep.ldo = D;
                return gemv_ln(c, EPI_RESID_F32, c->dmid, nb, F, L.w2, D, nullptr, nullptr, ep);
            }
            case 6: {   // self-attention at the positions in d_pos
                DecAttnParams p = dec_attn(c->dq, L.sk, L.sv, TGT, 0, c->d_pos, c->dattn, nb, H);
                return KD(c, cw_launch_attn_decode, c->bf16, p, c->st);
            }
            case 7: {   // logits
                EpiParams ep = epi0(); ep.outf = c->dlogits; ep.ldo = c->Vpad;
                return gemv_ln(c, EPI_STORE_F32, c->dx, nb, D, (c->bf16 && c->wpacked) ? c->embed_pk : c->embed, V, c->dec_ln_g, c->dec_ln_b, ep);
            }
            case 8:     // near-empty kernel: launch/boundary floor
                return KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st);
            case 9: {   // persistent stage A (declayer.hip) behind a near-empty launch that bumps the granule epoch (as the step's sampler does):
                        // without it the polls would find the previous launch's granules
                if (!c->bf16 || !c->fuse6_ready || !c->wpacked || nb > 8) return CW_ERR_INVALID;
                int r = KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st, c->d_epoch);
                if (r != CW_OK) return r;
                return KD(c, cw_launch_dec_layer, dec_layer_params(c, (launch_no - 1) % c->d.dec_layers, nb, c->dx, c->dx1), c->n_cu, c->st);
            }
            default: return CW_ERR_INVALID;
        }
    };
    if (which == 6 || which == 3) { CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st)); }
    for (int i = 0; i < 3; ++i) CWCHK(c, launch());
    HIPCHK(c, hipEventRecord(c->ev0, c->st));
    for (int i = 0; i < iters; ++i) CWCHK(c, launch());
    HIPCHK(c, hipEventRecord(c->ev1, c->st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / iters;
    const double e = (double)c->esz;
    switch (which) {
        case 0: *algo_bytes = (double)F * D * e + (double)nb * D * 4 + (double)nb * F * 4; break;
        case 1: *algo_bytes = 2.0 * nb * H * CW_N_CTX * 64 * (c->kv8 ? 1.0 : e) + 2.0 * nb * D * 4; break;
        case 9: *algo_bytes = 2.0 * nb * H * CW_N_CTX * 64 * e + 3.0 * D * D * e + 6.0 * nb * D * 4; break;
        case 2: case 4: *algo_bytes = (double)D * D * e + 2.0 * nb * D * 4; break;
        case 3: *algo_bytes = 3.0 * D * D * e + 4.0 * nb * D * 4; break;
        case 5: *algo_bytes = (double)F * D * e + (double)nb * (D + F) * 4; break;
        case 6: *algo_bytes = 2.0 * nb * H * 65 * 64 * e + 2.0 * nb * D * 4; break;
        case 7: *algo_bytes = (double)V * D * e + (double)nb * (D + V) * 4; break;
        default: *algo_bytes = 0; break;
    }
    return CW_OK;
}

// kernel launches behind stage `stage` of the layer as last enumerated by cw_time_decode_stage (17..64 rows: 2 where a preparation launch precedes the GEMV)
int32_t cw_decode_stage_launches(cw_ctx* c, int32_t stage) {
    return (stage >= 0 && stage < c->stage_count && stage < CW_MAX_DEC_STAGES) ? c->stage_launches[stage] : 0;
}
const char* cw_decode_stage_name(int32_t kind) { return (kind >= 0 && kind < DST_N) ? kDecStageName[kind] : "?"; }

// One launch of the decoder layer AS decode_step ISSUES IT at nb rows (same code path, same arguments), `iters` times back to back,
// cycling through the layers so that every launch streams its operands from HBM.  stage = index of the launch inside the layer
// (0 .. *n_stages - 1; the count depends on the rows, the dtype and the switches); stage == -1: the whole layer (all its launches).
int32_t cw_time_decode_stage(cw_ctx* c, int32_t nb, int32_t stage, int32_t iters, float* avg_ms, double* algo_bytes, int32_t* kind,
                             int32_t* n_stages) {
    if (nb < 1 || nb > c->Bm || iters < 1 || stage < -1 || stage >= CW_MAX_DEC_STAGES) return fail(c, CW_ERR_INVALID, "time_decode_stage: bad args");
    if (c->beam_K > 0) return fail(c, CW_ERR_STATE, "time_decode_stage: greedy rows only");
    // the launches run for real at decoder position 64: they append to the self-attention cache and write alignment rows there
    if (c->d.max_target_positions <= 65) return fail(c, CW_ERR_INVALID, "time_decode_stage: needs max_target_positions > 65 (has %d)", c->d.max_target_positions);
    CWCHK(c, cw_check_weights(c));
    CWCHK(c, epoch_hygiene(c, (long long)iters + 8));
    if (nb > c->nb_encoded) return fail(c, CW_ERR_STATE, "time_decode_stage: nb=%d but %d windows encoded (the cross-attention reads their K/V)", nb, c->nb_encoded);
    const int D = c->d.d_model, H = c->d.n_heads, F = c->d.ffn_dim, NL = c->d.dec_layers;
    // a hand-off that gave up while timing (shared GPU) must not send the next cw_decode to the fallback path: drain and clear
    struct Restore { cw_ctx* c; ~Restore() { c->stage_sel = -1; c->layer_sel = -1; (void)hipStreamSynchronize(c->st); (void)hipMemset(c->d_err, 0, 4); } } restore{c};
    c->hist_short = false;                                       // position 64: 65 keys
    CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st, c->d_epoch));
    // which launches does a layer have at this row count?  (nothing is launched: no stage has this index)
    c->stage_sel = CW_MAX_DEC_STAGES; c->layer_sel = 0;
    CWCHK(c, decode_step(c, nb, false));
    const int ns = c->stage_count;
    if (n_stages) *n_stages = ns;
    if (stage >= ns) return fail(c, CW_ERR_INVALID, "time_decode_stage: the layer has %d launches at %d rows", ns, nb);
    c->stage_sel = stage;
    auto one = [&](int i) -> int {
        c->layer_sel = i % NL;
        // granule tags are (epoch, layer): a second pass over the layers must not find the first pass's granules valid
        if (c->layer_sel == 0) CWCHK(c, KD(c, cw_launch_set_pos, c->d_pos, 64, nb, c->st, c->d_epoch));
        return decode_step(c, nb, false);
    };
    for (int i = 0; i < 3; ++i) CWCHK(c, one(i));
    HIPCHK(c, hipEventRecord(c->ev0, c->st));
    for (int i = 0; i < iters; ++i) CWCHK(c, one(3 + i));
    HIPCHK(c, hipEventRecord(c->ev1, c->st));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / iters;
    // algorithmic bytes: the weights of the projection(s), the cache rows, the activation rows in and out (f32 unless noted)
    const double e = (double)c->esz, act = (double)nb * D * 4.0;
    auto bytes_of = [&](int k) -> double {
        const double self_kv = 2.0 * nb * H * 65 * 64 * e;                                        // 65 cached positions (d_pos = 64)
        const double cross_kv = 2.0 * nb * H * CW_N_CTX * 64 * (c->kv8 ? 1.0 : e);
        switch (k) {
            case DST_QKV: return 3.0 * D * D * e + 4.0 * act;
            case DST_SELF_ATTN: return self_kv + 2.0 * act;
            case DST_QKV_SELF: return 3.0 * D * D * e + self_kv + 2.0 * act;
            case DST_O_PROJ: case DST_CROSS_Q: return (double)D * D * e + 2.0 * act;
            case DST_STACK: return 3.0 * D * D * e + 5.0 * act;                                   // x, a in; qa, qb, x1 out
            case DST_CROSS_ATTN: return cross_kv + 2.0 * act;
            case DST_STACK_CROSS: return 3.0 * D * D * e + cross_kv + 4.0 * act;
            case DST_CROSS_O: return (double)D * D * e + 2.0 * act;
            case DST_FC1: return (double)F * D * e + act + (double)nb * F * e;
            case DST_FC2: return (double)F * D * e + (double)nb * F * e + 2.0 * act;
            case DST_MLP_PAIR: case DST_MLP_CHAIN: return 2.0 * F * D * e + 3.0 * act;
            default: return 0.0;
        }
    };
    if (stage >= 0) {
        if (kind) *kind = c->stage_kind[stage];
        *algo_bytes = bytes_of(c->stage_kind[stage]);
    } else {
        if (kind) *kind = -1;
        double t = 0.0;
        for (int i = 0; i < ns && i < CW_MAX_DEC_STAGES; ++i) t += bytes_of(c->stage_kind[i]);
        *algo_bytes = t;
    }
    return CW_OK;
}

}  // extern "C"
