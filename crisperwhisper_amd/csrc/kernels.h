// Internal launcher interface between the kernel translation units and the engine.
#pragma once
#include "switches.h"
#include "common.h"

// A-operand addressing for the tile GEMMs: plain row-major or implicit conv1d(k=3, pad=1) gather.
struct AParams {
    const void* A;
    int lda;
    int amode;             // 0 = plain [M][lda]; 1 = conv gather over time-major [rows][C_in]
    int T_out, C_in, stride;
    const int* row_off;    // [batch] first input row of each item's window
    const int* row_valid;  // [batch] valid input rows in the window (beyond -> zeros)
};

struct CombineParams;
// ln_g != null, ln_b == null: plain normalisation (x - mean) * rstd -- the affine part lives in W / bias (fold_layernorm)

// elementwise.hip

struct SampleParams {
    const float* logits;       // [B][ldv], rows 16-byte aligned
    int V, B, ldv;
    const unsigned char* mask; // [V] bit0 always suppressed, bit1 suppressed at begin
    int eos, pad, timestamp_begin, max_initial_timestamp_index;  // max_initial < 0: none
    const int* cfg;            // device [4]: n_prompt (begin index), min_new_tokens, max_length, use_forced
    int* pos;                  // device [B]: position of the last decoder input; token pos+1 is being chosen
    int ids_stride;            // tokens per row in `ids`
    int* ids;                  // [B][ids_stride]  (in/out)
    const int* forced;         // [B][ids_stride] or null; value >= 0 forces that token at index t
    int* argmax_trace;         // [B][ids_stride] or null: the un-forced choice
    int* last_ts_tok;          // [B] last timestamp token generated (or -1)
    int* finished;             // [B]
    int* n_unfinished;         // [1] recomputed every call
    // fused embedding of the chosen token for the next decoder step
    const void* embed;         // [V][d] T
    const float* pos_embed;    // [max_target][d] f32
    float* x_out;              // [B][d] f32
    int d;
    int embed_bf16;
    const void* partials;      // [B][SAMPLE_NS] slice records (32 bytes each), written by stage 1 of the sampler
    // optional (null: off): running sum / count of log_softmax(processed scores)[chosen token] per row, for the average
    // log-probability of generation_whisper.py:1958-1974 (logprob_threshold)
    float* lp_sum;             // [B]
    int* lp_cnt;               // [B]
    unsigned int* epoch;       // optional: device counter of decoder forwards, bumped by block 0 (declayer.hip tags its granules with it)
};
// beam search (elementwise.hip): per row the n_cand best processed log-probabilities of the next token
// (log_softmax of the raw logits, then the same processors as the greedy path) ...
// ... and the re-ordering of the per-row state after the host picked (parent row, token) for every row
struct BeamAdvanceParams {
    int* ids; int* ids_tmp; int ids_stride;      // [rows][stride] token history
    int* anc; int* anc_tmp; int cap;             // [rows][cap] cache-row ancestry of the self-attention keys
    const int* parent; const int* token;         // [rows]
    int* pos;                                    // [rows] position of the last decoder input
    const void* embed; const float* pos_embed; float* x_out; int d; int embed_bf16;
    int rows;
};

// attention.hip
struct DecAttnParams {
    const float* q;        // [B][H*64] f32, already scaled
    const void* K;         // [B][H][cap][64] T
    const void* V;         // [B][H][cap][64] T
    int cap;               // rows allocated per (b,h)
    int n_keys;            // keys to attend over when `pos` is null
    const int* pos;        // device [B]: decoder input position; self-attention uses n_keys = pos[b] + 1
    float* out;            // [B][H*64] f32
    unsigned short* out_frag;  // non-null: write bf16 in MFMA fragment-major order instead (frag_index, K = H*64)
    float* align_out;      // [B][n_align][align_rows][S] or null
    const int* align_slot; // [H] slot index of each head in this layer (or -1), device
    int n_align, align_rows;            // alignment row written = pos[b]
    int B, H;
    int kv_div;            // > 1: query rows b share cache row b / kv_div (beams of one audio item; cross-attention)
    const int* anc;        // non-null: [B][cap] cache row holding key k of query row b (beam-search self-attention)
    int short_hist;        // host hint: every row's history is <= 64 keys (self-attention requests 64 rows per (row, head) up front, not 128)
};

// Cross-attention decode split over the 1500 keys (flash-decoding): ATT_NS blocks per (batch, head) write
// un-normalised partial outputs + (max, sum); the consumer GEMV combines them while loading its activations,
// alignment rows are normalised once per generate call by cw_launch_align_normalize.
#ifndef ATT_NS
#define ATT_NS 6
#endif
#ifndef CROSS_THREADS
#define CROSS_THREADS 512
#endif
struct CrossSplitParams {
    const float* q;        // [B][H*64]
    const void* K;         // [B][H][1500][64]
    const void* V;
    int n_keys;            // 1500
    float* part_o;         // [ATT_NS][B][H*64] un-normalised sum_k exp(s_k - m) v_k
    float* part_ml;        // [B][H][ATT_NS][2]  (m, l)
    float* align_out;      // [B][n_align][align_rows][n_keys] (un-normalised exp) or null
    float* align_ml;       // [B][n_align][align_rows][ATT_NS][2]
    const int* align_slot; // [H]
    const int* pos;        // [B] alignment row to write
    int n_align, align_rows, B, H;
    const float* kv_scale; // fp8 cache only: [B][H][2] dequantisation scales of K and V (null: K/V hold T)
    int kv_div;            // > 1: rows b share the K/V of audio item b / kv_div (beam search)
    // fused out-projection / query stage (decfuse.hip): q is finished here from the two partial projections and the statistics of
    // the residual row,  q = rstd(xs) * (qa + qb - mean(xs) * qw) + qbias;  all null: `q` holds the finished query
    const float* xstat;    // [B][H*64] residual rows the LayerNorm statistics are taken of (null: plain q), or
    const float* pstats;   // [n_pstats][16][2] per-block (sum, sum of squares) of those rows, left by the producing GEMV
    int n_pstats;
    const float* qa;       // [B][H*64] W'q x + W'q bo
    const float* qb;       // [B][H*64] (W'q Wo) a
    const float* qw;       // [H*64]    W'q 1
    const float* qbias;    // [H*64]    b'q
    float* a_out;          // [B][H*64] non-null: one block per (row, head) over all keys writes the finished output (no splits)
    void* a_frag;          // non-null: the same launch shape, output as 16-bit MFMA fragment-major rows (frag_index, K = H*64) for the
                           // out-projection of the 17..64-row path; q finished from `xstat` + q_planes planes at qa, or plain `q`
    int q_planes;          // a_frag + xstat: qa holds this many K-split planes (skinny.hip), q_plane_stride floats apart; qb may be null
    int q_plane_stride;
};
// opt-in fp8 (OCP e4m3) cross-attention cache: quantise one layer's bf16 K/V [B][H][S][64] with a scale per (b, h, K|V)
struct CombineParams {     // activations of a GEMV = combination of ATT_NS attention partials
    const float* part_ml;  // null: plain activations
    int H;                 // heads
    int plane;             // elements per partial plane (B * K)
    // 33..64 rows, column-owning out-projection behind this combine (gemv_mt_kernel OWN): the preparation launch also leaves the
    // row's mean one stage earlier, from the per-tile LayerNorm partial sums of gemv_stack_kernel ([group of 16 rows][tile][16][2])
    const float* pstats;   // null: not wanted
    int n_pstats;
    float* cvec_out;       // [rows]
};

// decfuse.hip: one launch over a row-stacked weight matrix W [sum n_tiles * 16][K]; up to 3 segments
struct StackSeg {
    const float* x;        // [Mb][K] f32 input rows of this segment (rounded to 16 bit, no LayerNorm)
    const float* bias;     // [n_tiles*16] or null
    const float* wsum;     // epi 0 / 2, non-null: [n_tiles*16] row sums of this segment's weights -- the rows are rounded as x - mean(x)
                           // (the operand a LayerNorm consumer is sensitive to) and mean(x) * wsum[n] is added back in f32
    const float* resid;    // epi 1: [Mb][n_tiles*16]
    float* out;            // [Mb][n_tiles*16]
    float* out2;           // epi 1: optional second copy of the result
    float* pstats;         // epi 1: optional [blocks of the segment][16][2] per-block (sum, sum of squares) of every output row
    int tile0, n_tiles;    // rows [tile0*16, (tile0+n_tiles)*16) of W
    int nt;                // 16-column tiles per block of this segment (<= the launch's NT; 0 = NT)
    int block0;            // filled in by the launcher
    int epi;               // 0: out = acc + bias;  1: out = resid + grid(acc + bias)   (2^-12 residual grid)
                           // 2: out += acc + bias (f32 atomics; two segments summing into a zeroed buffer: order-independent)
};
struct StackParams {
    const void* W;
    int wpk;               // W is fragment-major (gemm.hip: wfrag_pack_kernel) instead of row-major
    int K, Mb, nseg;
    StackSeg seg[3];
    float* zero;           // optional: the launch also clears zero_n4 float4 (a buffer a LATER launch accumulates into)
    int zero_n4;
};
// decfuse.hip: fc2 that finishes fc1 on load, mid = gelu(rstd (u - mean w1sum) + b1), (mean, rstd) = LayerNorm statistics of
// the residual rows `xstat` (an untouched copy: the launch's own atomics are already modifying the stream x)
struct Fc2xParams {
    const float* xstat;    // [Mb][D]
    const float* u;        // [Mb][F] W'1 x2 summands: W'1 x1 + W'1 bo_c + (W'1 Wo_c) a_c
    const float* w1sum;    // [F] W'1 1
    const float* b1;       // [F] b'1
    const void* W2;        // [D][F]
    const float* b2;       // [D]
    float* x;              // [Mb][D] residual stream, accumulated in place
    int Mb, D, F;
    int wpk;               // W2 is fragment-major
};

// decfuse.hip: fc1 -> GELU -> fc2 in ONE launch.  The F / 32 blocks form F / D independent groups of D / 32 blocks (a group =
// the fc1 columns of one d_model-wide K slice of fc2 = the blocks that then share that slice); inside a group the blocks meet
// at a reusable arrival barrier between the two GEMVs, and fc2's weights are requested before the barrier.
struct MlpPairParams {
    float* x;              // [Mb][D] residual stream: LayerNorm input of fc1, accumulated into by fc2 (f32 atomics, 2^-12 grid)
    const void* W1;        // [F][D] 16-bit, LayerNorm gamma folded in
    const float* b1;       // [F]
    const void* W2;        // [D][F]
    const float* b2;       // [D]
    void* mid;             // [8][F] 16-bit scratch (GELU output), handed over inside the launch
    unsigned int* bar;     // [2 * F / D]: arrival counter and generation of every group (zero-initialised, reusable)
    int* err;              // set to 1 when a block gave up waiting at the barrier (never expected; results are then invalid)
    int Mb, D, F;
    int wpk;               // W1 / W2 are fragment-major
    int fence;             // 1: agent-scope acquire fence behind the group barrier (round-3 behaviour, A/B: CW_MLP_PAIR_FENCE=1)
};

// decfuse.hip: decode GEMV for 17..64 rows (beam search: items x hypotheses) without the per-GEMV preparation launch.
// Activations are 16-bit MFMA fragment-major rows (common.h: frag_index).  A *producer* (residual epilogue, whole K per block,
// no atomics) leaves three things: the f32 residual stream, its 16-bit fragment-major copy (NOT normalised) and per-block
// partial sums (sum, sum of squares) of every row over the block's 16 columns.  A *consumer* multiplies the un-normalised copy by
// LayerNorm-folded weights W' and applies the LayerNorm through its linearity,  out = rstd (W' x - mean W'1) + b',  with the
// statistics summed from the producer's partials.
struct RowsParams {
    const void* xf;          // fragment-major 16-bit activations [ceil(Mb/16)][K/32][64][8]
    const void* W;           // [N][K] 16-bit weights, row-major or fragment-major (wpk)
    int Mb, K, N, wpk;
    EpiParams ep;            // consumer: any GEMV epilogue; producer: outf / resid / bias / ldo of the residual stream
    const float* ln_pstats;  // consumer: [ln_nblk][64][2] partial (sum, sum of squares) of the rows of x; null = plain GEMV
    int ln_nblk;
    const float* ln_wsum;    // consumer: [N] row sums of W'
    int lo_off;              // != 0: the residual copy is carried as two 16-bit halves, x = hi + lo, the low halves lo_off
                             // 16-byte fragments behind the high ones (consumer: reads xf; producer: writes xf_out)
    void* xf_out;            // producer: fragment-major 16-bit copy of the new residual rows (K' = N)
    float* pstats_out;       // producer: [N/16][64][2]
};

// skinny.hip: decoder linear layers for 17..64 rows -- 64-column x Kb blocks for ALL rows, activations staged once per block in
// LDS, K split over grid.y.  mode 0: f32 residual rows (LayerNorm by linearity, finished by skinny_finish_kernel) -> partial
// planes; mode 1: 16-bit fragment-major rows -> f32 atomics into the residual stream (2^-12 grid).
struct SkinnyParams {
    const float* x;        // mode 0: [Mb][K] f32 rows
    const void* xf;        // mode 1: 16-bit fragment-major rows [ceil(Mb/16)][K/32][64][8] (common.h: frag_index)
    const void* W;         // [N][K] 16-bit, fragment-major (gemm.hip: wfrag_pack_kernel)
    int Mb, K, N;
    int Kb;                // filled in by the launcher: K columns per block
    float* planes;         // mode 0: [S][Mb][N] f32 partial products, S = K / Kb
    float* stats;          // mode 0, optional: [S][Mb][2] slice statistics of the rows (mean, sum of squares about it) for the consumer's LayerNorm
    float* outf;           // mode 1: [Mb][ldo] f32, accumulated in place
    const float* bias;     // mode 1: [N] or null (added by slice 0)
    int ldo;
    int dbg;               // -DCW_SK_DEBUG builds only: ablation switches of tools/skinny_bench.py
};
struct SkinnyFinishParams {
    const float* planes;   // [S][Mb][N]
    int S, Mb, N;
    const float* x;        // [Mb][K] rows whose LayerNorm statistics apply (null: none)
    const float* stats;    // non-null: [S][Mb][2] slice statistics left by the GEMM -- used instead of re-reading x (K = row length)
    int K;
    const float* wsum;     // [N] row sums of the folded 16-bit weights (W' 1); null with x == null
    EpiParams ep;          // destination + folded bias; EPI_STORE_F32 / EPI_QKV_CACHE / EPI_GELU_FRAG
};

// declayer.hip: persistent decoder-layer kernel (one 1024-thread workgroup per CU), rows <= 8.  Stage A = the fused
// out-projection / cross-query stage (StackParams with NT = 1, fragment-major weights) + the key-split cross-attention
// (CrossSplitParams, FUSED) in ONE launch: the cross-attention K/V rows are requested at kernel entry and stream while the
// GEMV tile runs; results cross CUs as 8-byte {tag, value} granules.
struct DecLayerParams {
    const void* Ws;            // [3 D][D] 16-bit fragment-major [W'q_c ; W'q_c Wo ; Wo]
    const float* x;            // [Mb][D] residual rows entering the stage
    const float* a;            // [Mb][D] self-attention output
    const float* qa_bias;      // [D] W'q_c bo
    const float* q_wsum;       // [D] W'q_c 1 for the centred rounding of x (null: x is rounded as it is)
    const float* bo;           // [D]
    float* x1;                 // [Mb][D] out: x + grid(Wo a + bo)
    const void* K;             // [Mb][H][n_keys][64] cross-attention cache
    const void* V;
    int n_keys;
    float* part_o;             // as CrossSplitParams
    float* part_ml;
    float* align_out;
    float* align_ml;
    const int* align_slot;
    const int* pos;
    int n_align, align_rows;
    const float* qw;           // [D] W'q_c 1
    const float* qbias;        // [D] b'q_c
    unsigned long long* gq;    // [2][16][D] granules: qa, qb
    unsigned long long* gps;   // [D / 16][16][2] granules: per-tile (sum, sum of squares) of the rows of x1
    const unsigned int* epoch; // device counter, bumped once per decoder forward (set_pos_kernel / sample_kernel)
    int layer;
    int* err;                  // set to 1 when a poll gave up (results are then invalid; the engine reports it)
    int Mb, D, H;
    int kv_wait;               // A/B: 0 = K/V requested at kernel entry, 1 / 2 = after the chain's rows are in LDS / after its MFMAs
};

// declayer.hip: LayerNorm + q/k/v projection + self-attention of rows <= 8 in ONE launch (gemv2_bf16_kernel<EPI_QKV_CACHE> tiles,
// then attn_decode_kernel per (row, head) in the first rows x heads blocks; the tiles reach the attention as granules)
struct QkvSelfParams {
    const float* x;            // [Mb][D] residual rows (LayerNorm gamma / beta folded into W / bias)
    const void* W;             // [3 D][D] 16-bit fragment-major
    const float* bias;         // [3 D]
    void* sk;                  // self-attention cache [Mb][H][cap][64]: row pos[b] is appended, rows 0..pos[b] are attended over
    void* sv;
    int cap;
    const int* pos;            // [Mb] device
    float* out;                // [Mb][D] attention output
    float* q_plain;            // optional: the query also as plain f32 rows [Mb][D] (debugging: attn_decode_kernel can run behind this launch)
    int no_attn;               // debugging: tiles only
    unsigned long long* gq;    // [16][D] granules: query
    unsigned long long* gkv;   // [2][16][D / 2] granules: this step's key / value rows, two 16-bit values each
    const unsigned int* epoch;
    int layer;
    int* err;                  // 0, or 1 + the decoder position of the first forward in which a wait gave up
    int Mb, D, H;
    int fail_pos;              // test hook: the item of layer 0 gives up at this position (-1: never)
};

// declayer.hip: LayerNorm + fc1 + GELU and fc2 + residual of rows <= 8 in ONE launch (fc1 blocks publish a flag each, fc2 blocks
// request their weights at entry and wait for the flags)
struct MlpChainParams {
    const float* x;            // [Mb][D] residual rows entering the MLP (LayerNorm gamma / beta folded into W1 / b1)
    const void* W1;            // [F][D] 16-bit fragment-major
    const float* b1;           // [F]
    const void* W2;            // [D][F] 16-bit fragment-major
    const float* b2;           // [D]
    float* xio;                // [Mb][D] the residual rows again: fc2 accumulates x += grid(W2 mid + b2) in place
    void* mid;                 // [Mb][F] 16-bit gelu(fc1) rows (written write-through, read by sc1 loads)
    unsigned long long* flags; // [F / 32] {tag, 1}: fc1 block j has drained its stores
    const unsigned int* epoch;
    int layer;
    int* err;
    int Mb, D, F;
    int Kb2;                   // set by the launcher: K slice of an fc2 block
    int delay;                 // A/B: the fc2 blocks sleep this many x 64 clocks before they request their weights
};

// mel.hip
struct MelTables {
    const double* cos_t;  // [400]
    const double* sin_t;  // [400]
    const double* window; // [400]
    const float* filters; // [201][n_mels]
    // f64 matrix-core kernel (mel.hip: mel_mfma_kernel)
    const int* fb_lo;     // [n_mels] first / last FFT bin with a non-zero filter weight
    const int* fb_hi;
    int dbg;              // -DCW_SK_DEBUG builds: ablation switches
};


// ---- launchers of the dtype-dependent translation units (gemm / attention / elementwise / mel .hip), compiled twice:
// namespace cw_bf16 (bfloat16 build) and cw_f16 (-DCW_F16, IEEE binary16); `bool bf16` = "16-bit engine" in both, false
// selects the f32 parity kernels (identical in the two builds).
#define CW_DTYPE_KERNEL_DECLS \
    int cw_launch_gemm(bool bf16, int epi, const AParams& ap, const void* W, int M, int N, int K, const EpiParams& ep, hipStream_t st); \
    void cw_gemm_set_256_min_tiles(int n); \
    void cw_gemm_set_pp(int on); \
    void cw_gemm_set_8ph(int on); \
    void cw_gemm_set_gm(int gm); \
    void cw_gemm_set_w128(int on); \
    void cw_gemv_set_comb_rowgroups(int on); \
    void cw_gemv_set_mt_variant(int v); \
    void cw_gemv_set_loop(int on); \
    int cw_launch_gemm_w128(int epi, const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const EpiParams& ep, int tm2, int tn2, hipStream_t st); \
    void cw_cross_set_valu(int on); \
    void cw_cross_set_per_row(int on); \
    int cw_launch_layernorm_fp8(const float* x, const float* g, const float* b, void* out8, float* scale, int rows, int d, hipStream_t st); \
    int cw_launch_quant_rows_fp8(const void* x, int rows, int K, void* out8, float* scale, hipStream_t st); \
    int cw_launch_gemm_fp8(int epi, const void* A8, int lda, const void* W8, int M, int N, int K, const float* sa, const float* sw, const EpiParams& ep, hipStream_t st); \
    int cw_launch_fold_layernorm(const float* Wf, int N, int K, const float* g, const float* beta, float scale, void* w_out, float* bias, hipStream_t st); \
    int cw_launch_gemv(bool bf16, int epi, const float* x, int Mb, int K, const void* W, int N, const float* ln_g, const float* ln_b, const EpiParams& ep, hipStream_t st, const CombineParams* comb = nullptr, void* scratch = nullptr, bool wpacked = false); \
    int cw_gemv_own_nt(int N); \
    int cw_launch_gemv_own(const void* xf, int Mb, int K, const void* W, int N, const EpiParams& ep, const float* cvec, void* xf_out, float* stats_out, hipStream_t st, bool wpacked); \
    int cw_launch_gemv_lna(const void* xf, int Mb, int K, const void* W, int N, const EpiParams& ep, const float* stats_in, int n_stats, const float* wsum, hipStream_t st, bool wpacked); \
    size_t cw_wfrag_elems(int N, int K); \
    int cw_launch_wfrag_pack(const void* src, int N, int K, void* dst, hipStream_t st); \
    int cw_launch_fold_product(const float* A, const float* s, float scale, const float* B, int N, int J, int K, void* C16, hipStream_t st); \
    int cw_launch_fold_rowvec(const float* A, const float* s, float scale, const float* v, const void* W16, int N, int J, float* c_out, float* w_out, hipStream_t st); \
    int cw_launch_gemv_stack(const StackParams& p, int nt, hipStream_t st); \
    int cw_launch_dec_layer(const DecLayerParams& p, int n_cu, hipStream_t st); \
    bool cw_mlp_chain_ok(int Mb, int D, int F); \
    int cw_launch_mlp_chain(const MlpChainParams& p, hipStream_t st); \
    int cw_launch_qkv_self(const QkvSelfParams& p, hipStream_t st); \
    int cw_qkv_self_blocks_per_cu(int D, int cap); \
    size_t cw_dec_layer_lds(int D); \
    int cw_launch_gemv_fc2x(const Fc2xParams& p, hipStream_t st); \
    int cw_launch_mlp_pair(const MlpPairParams& p, hipStream_t st); \
    int cw_launch_gemv_rows(int epi, bool produce, const RowsParams& p, hipStream_t st); \
    int cw_launch_rows_combine(const float* part_o, int Mb, int K, const CombineParams& cb, void* xf, hipStream_t st); \
    int cw_launch_rows_prep(const float* x, int Mb, int K, void* xf, float* pstats, int lo_off, hipStream_t st); \
    int cw_skinny_pick_nks(int N, int K, int s_max); \
    int cw_launch_skinny(int mode, const SkinnyParams& p, int nks, hipStream_t st); \
    int cw_launch_skinny_finish(int epi, const SkinnyFinishParams& p, hipStream_t st); \
    void cw_launch_skinny_empty(hipStream_t st); \
    int cw_launch_layernorm(bool bf16_out, const float* x, const float* g, const float* b, void* out, int rows, int d, hipStream_t st); \
    int cw_launch_layernorm_f32(const float* x, const float* g, const float* b, float* out, int rows, int d, hipStream_t st); \
    int cw_launch_sample(const SampleParams& p, hipStream_t st); \
    int cw_launch_beam_topk(const SampleParams& p, int n_cand, float* cand_val, int* cand_id, float* scratch, hipStream_t st); \
    size_t cw_beam_topk_scratch_floats(int rows); \
    void cw_beam_topk_set_1block(int on); \
    int cw_launch_beam_advance(const BeamAdvanceParams& p, hipStream_t st); \
    int cw_launch_align_gather(const float* align, const int* row_of_pos, int n_items, int n_align, int align_rows, int L, int n_keys, float* out, hipStream_t st); \
    int cw_launch_set_pos(int* pos, int value, int B, hipStream_t st, unsigned int* epoch = nullptr); \
    int cw_launch_embed(const int* ids, int ids_stride, int t, const void* embed, int embed_bf16, const float* pos_embed, float* x_out, int B, int d, hipStream_t st); \
    int cw_launch_attn_encoder(bool bf16, const void* Q, const void* K, const void* V, void* out, int B, int H, int S, int S_pad, hipStream_t st); \
    int cw_launch_attn_decode(bool bf16, const DecAttnParams& p, hipStream_t st); \
    int cw_launch_attn_cross_split(bool bf16, const CrossSplitParams& p, hipStream_t st); \
    int cw_launch_kv_quant_fp8(const void* K, const void* V, void* K8, void* V8, float* kv_scale, int B, int H, int S, hipStream_t st); \
    bool cw_cross8_is_mfma(int n_keys); \
    int cw_launch_attn_cross_split_fp8(const CrossSplitParams& p, hipStream_t st); \
    size_t cw_kv8_v_bytes(int H, int S); \
    int cw_launch_align_normalize(float* align, const float* align_ml, int B, int n_align, int align_rows, int L, int n_keys, hipStream_t st); \
    int cw_launch_mel(const MelTables& t, const float* pcm, int B, int n_mels, float* logspec_tm, unsigned int* gmax, hipStream_t st); \
    int cw_launch_mel_finish(const float* logspec_tm, const unsigned int* gmax, int B, int n_mels, void* feats_tm, int feats_bf16, float* feats_hf, hipStream_t st);
namespace cw_bf16 { CW_DTYPE_KERNEL_DECLS }
namespace cw_f16 { CW_DTYPE_KERNEL_DECLS }

// ingest.hip
int cw_launch_pcm_to_mono(const void* raw, int fmt, int channels, long long n_frames, float* out, hipStream_t st);
int cw_launch_normalise(float* x, long long n, double* acc2, hipStream_t st);
int cw_launch_resample(const float* x, long long n_in, const float* taps_t, int orig, int nw, int width,
                       long long n_out, float* out, hipStream_t st);

// align.hip
int cw_launch_align_stats(const float* w, int B, int Ha, int rows_cap, int S, int row0, int N, const int* n_cols,
                          float* mean, float* stdv, hipStream_t st);
int cw_launch_align_filter(const float* w, int B, int Ha, int rows_cap, int S, int row0, int N, const int* n_cols,
                           const float* mean, const float* stdv, int width, float* mat, hipStream_t st);
// skew: workspace of cw_dtw_skew_floats(B, N, S) floats (null: round-1 block kernel)
size_t cw_dtw_skew_floats(int B, int N, int S);
int cw_launch_dtw(const float* mat, int B, int N, int S, const int* n_cols, unsigned char* trace, int* first_col,
                  int* path_text, int* path_time, int* path_len, hipStream_t st, float* skew = nullptr);
int cw_launch_pauses(double* start, double* end, int W, double thr, hipStream_t st);
