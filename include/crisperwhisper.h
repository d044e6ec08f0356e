/*
 * crisperwhisper.h -- C ABI of the MI355X-native CrisperWhisper inference-and-alignment path.
 *
 * The reference (nyrahealth/CrisperWhisper @ 2024-10-22) has no FFI layer: its hot path is reached
 * through the HuggingFace pipeline object protocol (REF/transcribe.py:21-33).  This header declares
 * the device-side replacement for each internal seam of that protocol; the
 * Python shim in crisperwhisper_amd/ binds it with ctypes and re-implements the
 * pipeline call surface on top.  "TF/" = site-packages/transformers 5.15.0, "REF/" = the reference.
 *
 * Conventions: every function returns 0 on success or a negative errno-style code (cw_last_error()
 * gives the message); no exceptions cross the ABI; all pointer arguments are caller-owned HOST
 * buffers unless the name ends in _dev; the context owns all device memory and one HIP stream; one
 * context per device per process; a context is not re-entrant (external locking).
 */
#ifndef CRISPERWHISPER_H
#define CRISPERWHISPER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cw_ctx cw_ctx;

#define CW_DTYPE_F32 0   /* parity mode: f32 weights, activations and arithmetic                       */
#define CW_DTYPE_BF16 1  /* performance mode: bf16 weights/activations, f32 accumulation + residuals   */
#define CW_DTYPE_F16 2   /* same engine in IEEE binary16, the reference's GPU dtype (REF/transcribe.py:10,  */
                         /* REF/app.py:111): same MFMA rate, 3 more significand bits than bfloat16           */

#define CW_N_SAMPLES 480000  /* 30 s @ 16 kHz  (TF/models/whisper/feature_extraction_whisper.py:88-93) */
#define CW_N_FRAMES 3000     /* mel frames per window                                                  */
#define CW_N_CTX 1500        /* encoder frames (max_source_positions)                                  */

/* Model geometry: the fields of WhisperConfig the path reads (TF/models/whisper/configuration_whisper.py). */
typedef struct {
    int32_t d_model, n_heads, ffn_dim, enc_layers, dec_layers, n_mels, vocab_size;
    int32_t max_target_positions;   /* 448                                                             */
    int32_t median_filter_width;    /* config.median_filter_width, read at generation_whisper.py:346   */
    int32_t dtype;                  /* CW_DTYPE_*                                                      */
    int32_t max_batch;              /* chunks in flight (pipeline batch_size, REF/transcribe.py:27)    */
    int32_t n_align;                /* generation_config.alignment_heads                               */
    const int32_t* align_layers;    /* [n_align]                                                       */
    const int32_t* align_heads;     /* [n_align]                                                       */
} cw_model_desc;

/* Generation settings: what WhisperGenerationMixin.generate derives from generation_config
 * (TF/models/whisper/generation_whisper.py:650-722, 1774-1812).                                       */
typedef struct {
    int32_t eos_token_id, pad_token_id;
    int32_t no_timestamps_token_id;          /* timestamp_begin = this + 1                              */
    int32_t max_initial_timestamp_index;     /* < 0: unset                                              */
    const int32_t* suppress_tokens; int32_t n_suppress;
    const int32_t* begin_suppress_tokens; int32_t n_begin_suppress;
} cw_gen_cfg;

int32_t cw_abi_version(void);
cw_ctx* cw_create(const cw_model_desc* desc, int32_t device);
void cw_destroy(cw_ctx* ctx);
const char* cw_last_error(cw_ctx* ctx);          /* ctx may be NULL after a failed cw_create           */
int32_t cw_sync(cw_ctx* ctx);

/* Weights: one call per tensor of WhisperForConditionalGeneration.state_dict() (HF names, f32 host data,
 * replaces AutoModelForSpeechSeq2Seq.from_pretrained + model.to(device), REF/transcribe.py:14-17).
 * The context fuses q/k/v projections, folds the 1/8 query scale (modeling_whisper.py:309) and re-lays
 * the conv kernels for implicit GEMM.  proj_out.weight is tied to embed_tokens (:965) and ignored.     */
int32_t cw_load_tensor(cw_ctx* ctx, const char* hf_name, const float* data, const int64_t* shape, int32_t ndim);
/* 0 when every tensor of the geometry has been received, else CW_ERR_STATE with the missing names in cw_last_error
 * (also enforced by the first cw_encode: device buffers start zero-filled, a partial checkpoint must not run).     */
int32_t cw_check_weights(cw_ctx* ctx);
int32_t cw_set_generation(cw_ctx* ctx, const cw_gen_cfg* cfg);
/* Context options (before cw_encode).  "cross_kv_fp8" = 1: the cross-attention K/V cache is additionally stored in OCP
 * e4m3 with one scale per (chunk, head, K|V) and the decode step streams that copy (half the bytes of the dominant
 * stream; 16-bit engines only; an accuracy-gated performance mode, BASELINE configs[3], not the parity path).  The copy is
 * read by v_mfma_f32_16x16x32_fp8_fp8 (K row-major, V in the fragment order of the B operand; csrc/attention.hip).
 * "encoder_gemm_fp8" = 1 (after the weights are loaded; 16-bit engines): e4m3 copies of the encoder's qkv / fc1 / fc2 and the
 * decoder's cross-K/V weights are made with one scale per output row, the LayerNorms in front of those GEMMs emit e4m3 rows with
 * one scale per row, and the GEMMs run on v_mfma_scale_f32_16x16x128_f8f6f4 -- the fp8 MFMA half of BASELINE configs[3];
 * accuracy-gated like the cache option, 0 switches back.                                                              */
int32_t cw_set_option(cw_ctx* ctx, const char* name, int32_t value);

/* ---- audio ingest (the step in front of seam 1) -------------------------------------------------
 * cw_ingest: interleaved little-endian sample frames as they sit in a RIFF/WAVE data chunk -> mono f32 at sr_out, on
 * device: integer formats scaled to [-1, 1), channels averaged (what `ffmpeg -ac 1 -ar 16000 -f f32le` of
 * TF/pipelines/audio_utils.py:9-45 delivers), optional (y - mean) / std / 8 of REF/app.py:85-93, then
 * torchaudio.functional.resample with its defaults (sinc_interp_hann, lowpass_filter_width = 6, rolloff = 0.99:
 * TF/pipelines/automatic_speech_recognition.py:398-412, REF/app.py:94-95).  out holds
 * cw_resampled_length(n_frames, sr_in, sr_out) = ceil(n_frames * sr_out / sr_in) floats.
 * cw_resample_taps exposes the tap table [new][2*width + orig] (f64 arithmetic rounded to f32) for differential tests. */
#define CW_PCM_U8 0
#define CW_PCM_S16 1
#define CW_PCM_S24 2
#define CW_PCM_S32 3
#define CW_PCM_F32 4
#define CW_PCM_F64 5
int64_t cw_resampled_length(int64_t n_frames, int32_t sr_in, int32_t sr_out);
int32_t cw_ingest(cw_ctx* ctx, const void* raw, int32_t fmt, int32_t channels, int64_t n_frames, int32_t sr_in,
                  int32_t sr_out, int32_t normalise, float* out);
int32_t cw_resample_taps(int32_t sr_in, int32_t sr_out, float* taps, int32_t cap, int32_t* orig, int32_t* nw,
                         int32_t* width);

/* FLAC container (host, no GPU): the part of `ffmpeg_read` (TF/pipelines/audio_utils.py:9-45) that turns a .flac file into
 * integer samples -- RFC 9639 decoder with CRC-8 / CRC-16 / STREAMINFO-MD5 verification (csrc/flac.cpp).  cw_flac_decode writes
 * interleaved samples left-justified to 32 bits ([frames][channels] int32: exactly what cw_ingest(CW_PCM_S32) scales by
 * 2^-31, i.e. ffmpeg's s16 / s32 -> f32 conversion); pcm_s32 == NULL only reports the frame count.  cap_frames > 0 also
 * bounds the decoding: the frame loop stops with an error as soon as more frames than that were produced (a decompression
 * bomb never materialises); with pcm_s32 == NULL it is the most the caller would accept, 0 = unbounded.  Errors: negative
 * code, text in cw_flac_last_error().                                                                                 */
int32_t cw_flac_info(const uint8_t* data, int64_t n_bytes, int32_t* sample_rate, int32_t* channels,
                     int32_t* bits_per_sample, int64_t* total_frames);
int32_t cw_flac_decode(const uint8_t* data, int64_t n_bytes, int32_t* pcm_s32, int64_t cap_frames, int64_t* n_frames);
const char* cw_flac_last_error(void);

/* ---- seam 1: feature extractor (WhisperFeatureExtractor.__call__, feature_extraction_whisper.py:193-346)
 * pcm: [B][n_samples[b]] packed back to back (each <= CW_N_SAMPLES; zero-padded to 30 s on device).
 * feats_out (nullable): [B][n_mels][3000] f32, HF layout.  n_frames_out (nullable): attention_mask.sum(-1).
 * Features stay resident in the context as items 0..B-1 for cw_encode.                                 */
int32_t cw_mel(cw_ctx* ctx, const float* pcm, int32_t B, const int32_t* n_samples, float* feats_out,
               int32_t* n_frames_out);
/* Same, split so a benchmark can keep the PCM resident in HBM and time only device work.               */
int32_t cw_upload_pcm(cw_ctx* ctx, const float* pcm, int32_t B, const int32_t* n_samples);
int32_t cw_mel_resident(cw_ctx* ctx, int32_t B);
/* Test hook: install externally computed features ([B][n_mels][3000] f32) as items 0..B-1.             */
int32_t cw_set_features(cw_ctx* ctx, const float* feats, int32_t B);

/* ---- seam 2: model.generate, split into its device stages ------------------------------------------
 * cw_encode: WhisperEncoder.forward (modeling_whisper.py:590-646) on the 3000-frame windows
 *   features[item[i]][:, seek[i] : seek[i] + n_frames[i]] zero-padded to 3000 (generation_whisper.py:1831-1852),
 *   followed by the cross-attention K/V projection of all decoder layers (:322-335).                   */
int32_t cw_encode(cw_ctx* ctx, int32_t nb, const int32_t* item, const int32_t* seek, const int32_t* n_frames);
int32_t cw_get_encoder_output(cw_ctx* ctx, float* out /* [nb][1500][d_model] */, int32_t nb);

/* cw_decode: one greedy GenerationMixin.generate call (TF/generation/utils.py:2783-2973) over the nb
 * encoded windows: decoder forward with KV caches, logits processors (logits_process.py:203-260,
 * 1816-2047), argmax, eos/pad bookkeeping, until every row is finished.  Alignment-head cross-attention
 * rows are retained on device for cw_token_timestamps.
 *   prompt [nb][n_prompt]; max_new_tokens < 0: bounded by max_length; forced (nullable) [nb][max_target]:
 *   entries >= 0 teacher-force that sequence position (the un-forced choice still goes to argmax_out).
 *   sequences [nb][max_target] (prompt included), lengths [nb] = prompt + generated,
 *   argmax_out (nullable) [nb][max_target].                                                            */
int32_t cw_decode(cw_ctx* ctx, int32_t nb, const int32_t* prompt, int32_t n_prompt, int32_t max_length,
                  int32_t min_new_tokens, const int32_t* forced, int32_t* sequences, int32_t* lengths,
                  int32_t* argmax_out);
/* Deterministic half of generate_with_fallback (generation_whisper.py:970-1116, _need_fallback :1243-1287) at temperature 0:
 * with both thresholds set, a window whose average token log-probability (log_softmax of the processed scores at the
 * generated tokens, eos included, :1958-1974) is below logprob_threshold AND whose no-speech probability
 * (WhisperNoSpeechDetection, logits_process.py:2050-2112: softmax of the raw logits at the <|startoftranscript|> position, token
 * no_timestamps_token_id - 1) is above no_speech_threshold is skipped: seek advances by the window, no segment (:879-881).
 * NaN = unset.  cw_transcribe applies them; stage-wise callers use cw_no_speech_probs before cw_decode and
 * cw_get_avg_logprobs after it (token scores are tracked while a logprob threshold is set).  The stochastic half --
 * re-decoding at higher temperatures -- is not implemented.                                                              */
int32_t cw_set_thresholds(cw_ctx* ctx, float logprob_threshold, float no_speech_threshold);
int32_t cw_no_speech_probs(cw_ctx* ctx, int32_t nb, int32_t sot_token, float* out /* [nb] */);
int32_t cw_get_avg_logprobs(cw_ctx* ctx, float* out /* [nb] */, int32_t nb);
int32_t cw_get_logits(cw_ctx* ctx, float* out /* [nb][vocab] */, int32_t nb);       /* last sampled step */
int32_t cw_set_logits_capture(cw_ctx* ctx, float* host_buf, int32_t max_steps);    /* [steps][nb][vocab] */
int32_t cw_get_alignment(cw_ctx* ctx, float* out /* [nb][n_align][L][1500] */, int32_t nb, int32_t L);

/* cw_transcribe: the whole of WhisperGenerationMixin.generate(return_timestamps=True, return_token_timestamps=True)
 * for the B feature items resident after cw_mel -- init tokens incl. language detection (:1455-1608, :1610-1673,
 * reusing the first encoder pass), the seek loop with batch shrinking (:785-903), eos/pad stripping (:1060-1082),
 * _retrieve_segment (:1977-2074) -- on top of cw_encode / cw_decode / cw_token_timestamps.  Greedy, no fallback.
 * Output per item: the concatenated segment tokens and their absolute token timestamps (what the pipeline hands to
 * _decode_asr, TF/pipelines/automatic_speech_recognition.py:529-540): tokens/token_ts [B][cap], lens [B].          */
typedef struct {
    int32_t sot_token;               /* decoder_start_token_id (<|startoftranscript|>)                             */
    int32_t language_token;          /* e.g. id of <|en|>; -1: detect per item                                   */
    int32_t task_token;              /* id of <|transcribe|> / <|translate|>; -1: none (only valid with detection) */
    int32_t max_new_tokens;          /* -1: bounded by max_length                                                  */
    int32_t min_new_tokens;          /* 0: none                                                                    */
    int32_t max_length;              /* generation_config.max_length (448)                                         */
    const int32_t* lang_ids; int32_t n_lang_ids;   /* generation_config.lang_to_id values (for detection)         */
} cw_transcribe_cfg;
int32_t cw_transcribe(cw_ctx* ctx, int32_t B, const int32_t* num_frames, const cw_transcribe_cfg* cfg,
                      int32_t* tokens, float* token_ts, int32_t* lens, int32_t cap, int32_t* n_passes);

/* ---- beam search (the transformers 5.x ASR pipeline defaults to num_beams = 5,
 * TF/pipelines/automatic_speech_recognition.py:160-163): device half of GenerationMixin._beam_search
 * (TF/generation/utils.py:3208-3520) over the encoded windows 0..n_items-1, items x beams decoder rows (row = item *
 * num_beams + beam; the context must have been created with max_batch >= items x beams).
 *   cw_beam_begin    prompt [n_items][n_prompt] replicated over the beams, prompt positions forwarded.
 *   cw_beam_step     one decoder forward for every row + log_softmax + the logits processors (:3402-3403); per row the
 *                    n_cand (<= 64) best processed log-probabilities and their tokens, best first (-inf / -1 padded).
 *   cw_beam_advance  the caller chose, for every row, the row it descends from (same item) and its next token
 *                    (:3125-3170, :3480-3486): token history, self-attention cache ancestry and decoder input follow.
 *   cw_beam_finish   row_of_pos [n_items][L]: for every returned sequence and decoder input position, the row whose
 *                    forward pass produced it (HF's unrolled `beam_indices`, generation_whisper.py:262-303): the
 *                    alignment-head rows are gathered accordingly; cw_token_timestamps(n_items, L, ...) follows.   */
int32_t cw_beam_begin(cw_ctx* ctx, int32_t n_items, int32_t num_beams, const int32_t* prompt, int32_t n_prompt,
                      int32_t max_length, int32_t min_new_tokens);
int32_t cw_beam_step(cw_ctx* ctx, int32_t n_cand, float* cand_logprob, int32_t* cand_token);
int32_t cw_beam_advance(cw_ctx* ctx, const int32_t* parent, const int32_t* token);
int32_t cw_beam_finish(cw_ctx* ctx, int32_t n_items, int32_t L, const int32_t* row_of_pos);

/* Host half of the same search (running / finished hypotheses, length penalty, early-stopping heuristic:
 * TF/generation/utils.py:3147, 3173-3245, 3009-3053; float32 like HF), host-only C++: no cw_ctx, no GPU.  Per decoder step:
 *   cw_beam_step -> cw_beam_host_step (candidates in, parent / token out; 1 = go on, 0 = search over) -> cw_beam_advance.
 * cw_beam_host_result: best hypothesis per item -- sequences [n_items][max_length] (pad / eos filled), beam_indices
 * [n_items][max_length - n_prompt] (flat row of every generated position, -1 behind the end), its score.               */
typedef struct cw_beam_host cw_beam_host;
cw_beam_host* cw_beam_host_new(int32_t n_items, int32_t num_beams, int32_t n_prompt, int32_t max_length, int32_t vocab_size,
                               int32_t eos_token_id, int32_t pad_token_id, double length_penalty, int32_t early_stopping,
                               const int32_t* prompt /* [n_items][n_prompt] */);
int32_t cw_beam_host_step(cw_beam_host* s, const float* cand_logprob, const int32_t* cand_token /* [rows][2 * num_beams] */,
                          int32_t* parent, int32_t* token /* [rows] */);
int32_t cw_beam_host_result(const cw_beam_host* s, int64_t* sequences, int32_t* beam_indices, float* score);
void cw_beam_host_free(cw_beam_host* s);

/* cw_token_timestamps: _extract_token_timestamps (generation_whisper.py:241-381) on the retained rows:
 * crop to num_frames[b]//2 encoder frames, drop the n_prompt prompt rows, z-score over tokens, median
 * filter, head mean, DTW, jump times.  L = rows retained = max(lengths) - 1.  ts_out [nb][L+1] seconds. */
int32_t cw_token_timestamps(cw_ctx* ctx, int32_t nb, int32_t L, int32_t n_prompt, const int32_t* num_frames,
                            float* ts_out);

/* ---- stand-alone differential-test entry points for the alignment kernels ---------------------------- */
/* attn [B][Ha][N][M] -> mat [B][N][M] (z-score, median(width), head mean); n_cols[b] <= M columns used.  */
int32_t cw_align_matrix(cw_ctx* ctx, const float* attn, int32_t B, int32_t Ha, int32_t N, int32_t M,
                        const int32_t* n_cols, int32_t width, float* mat_out);
/* _dynamic_time_warping(-mat) (generation_whisper.py:64-115): text_idx/time_idx [N+M] forward order.    */
int32_t cw_dtw(cw_ctx* ctx, const float* mat, int32_t N, int32_t M, int32_t* text_idx, int32_t* time_idx,
               int32_t* path_len);
/* ---- seam 4: adjust_pauses_for_hf_pipeline_output (REF/utils.py:1-29) on word start/end arrays, in place */
int32_t cw_adjust_pauses(cw_ctx* ctx, double* start, double* end, int32_t W, double split_threshold);

/* ---- seam 3: tokenizer._decode_asr(..., return_timestamps="word") (TF/models/whisper/tokenization_whisper.py
 * :901-1406), host-only (no GPU): chunk-seam merge, word grouping, punctuation merge, 0.01 s rounding.
 * Vocabulary: byte-level BPE table.  blob/offsets[n_tokens+1]: raw bytes of each text token; kind[i]: 0 text,
 * 1 special (<|...|>), 2 other (timestamp tokens); lang_class[i] for specials: -1 not a language tag, 0 language
 * written with spaces, 1 language without spaces (zh/ja/th/lo/my/yue: split on unicode points, :1299-1304).      */
typedef struct cw_vocab cw_vocab;
typedef struct cw_collator cw_collator;
cw_vocab* cw_vocab_create(int32_t n_tokens, const uint8_t* blob, const int64_t* offsets, const int8_t* kind,
                          const int8_t* lang_class, int32_t eos, int32_t timestamp_begin, int32_t startofprev,
                          int32_t sot, int32_t default_lang_class);
void cw_vocab_destroy(cw_vocab* v);
cw_collator* cw_collate_begin(const cw_vocab* v, double time_precision);
/* mode 0 (default): word chunks (return_timestamps="word"); mode 1: one chunk per timestamp-delimited segment
 * (return_timestamps=True, :1060-1075): token_ts is ignored, a missing start/end comes back as NaN                   */
int32_t cw_collate_set_mode(cw_collator* c, int32_t mode);
/* one pipeline output (chunk) in audio order: tokens [n_tokens], token_ts [n_ts] seconds, stride in seconds     */
int32_t cw_collate_feed(cw_collator* c, const int64_t* tokens, int32_t n_tokens, const float* token_ts, int32_t n_ts,
                        int32_t has_stride, double chunk_len, double stride_left, double stride_right);
/* flushes leftovers; returns sizes: words, utf-8 bytes of the full text / of all word texts, warned = 1 when
 * Whisper did not predict an ending timestamp (:1112-1116)                                                       */
int32_t cw_collate_finish(cw_collator* c, int32_t* n_words, int64_t* text_bytes, int64_t* words_bytes, int32_t* warned);
int32_t cw_collate_get(cw_collator* c, uint8_t* text, double* starts, double* ends, int64_t* word_offsets /* [n+1] */,
                       uint8_t* words_blob);
void cw_collate_free(cw_collator* c);

/* ---- kernel-level hooks used by the parity tests (host f32 in/out, run in the context's dtype) -------- */
/* process-wide tuning knobs for the tests: "gemm256_min_tiles" = tile count from which the 256x256 GEMM is used */
/* 1 when the library carries the measured-and-rejected kernel variants (built with make EXTRA=-DCW_EXPERIMENTS). */
int32_t cw_has_experiments(void);
int32_t cw_test_set_option(const char* name, int32_t value);
int32_t cw_test_gemm(cw_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* A, const float* W,
                     const float* bias, int32_t gelu, float* out);
/* e4m3 x e4m3 GEMM of the opt-in fp8 encoder mode (row-wise scales, v_mfma_scale_f32_16x16x128_f8f6f4): out = T(A W^T + bias) */
int32_t cw_test_gemm_fp8(cw_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* A, const float* W, const float* bias,
                         int32_t gelu, float* out);
int32_t cw_test_gemv(cw_ctx* ctx, int32_t Mb, int32_t N, int32_t K, const float* x, const float* W,
                     const float* bias, const float* ln_g, const float* ln_b, int32_t gelu, float* out);
/* One skinny-M decoder projection (17..64 rows; csrc/skinny.hip) on caller-supplied rows, 16-bit engines.  mode 0: LayerNorm
 * (no affine part) + projection through K-split planes and the finish launch; 1: the same through GELU; 2: residual rows
 * out += x16 W^T + bias by grid atomics.  nks = k-steps of 32 per block (0 = default), reps > 0 also times the launches on cold
 * weights: us[0] GEMM, us[1] finish (microseconds per launch). */
int32_t cw_test_skinny(cw_ctx* ctx, int32_t mode, int32_t Mb, int32_t N, int32_t K, const float* x, const float* W,
                       const float* bias, int32_t nks, int32_t reps, float* out, float* us);
int32_t cw_test_attention(cw_ctx* ctx, int32_t B, int32_t H, int32_t S, const float* q, const float* k,
                          const float* v, float* out /* [B][S][H*64] */);
/* One launch of the key-split cross-attention decode kernel (CW_ATT_SPLITS = 6 key splits): q [B][H*64] pre-scaled, k / v
 * [B / kv_div][H][S][64] (kv_div rows share one K/V: the hypotheses of an audio item under beam search).  Raw outputs:
 * part_o [6][B][H*64], part_ml [B][H][6][2] = (max, sum) per split; head `align_head` captured as the only alignment head:
 * align [B][S] = exp(s - max of its split), align_ml [B][6][2].                                                     */
#define CW_ATT_SPLITS 6
int32_t cw_test_cross_attention(cw_ctx* ctx, int32_t B, int32_t H, int32_t S, int32_t kv_div, const float* q, const float* k,
                                const float* v, int32_t align_head, float* part_o, float* part_ml, float* align,
                                float* align_ml);
/* One launch of the fused logits processors + greedy choice (MinNewTokensLength, SuppressTokensAtBegin, SuppressTokens,
 * WhisperTimeStamp: TF/generation/logits_process.py:203-260, 1816-2047; argmax TF/generation/utils.py:2925) on
 * caller-supplied rows: logits [nb][vocab], ids [nb][t] = prompt + tokens generated so far; choice_out [nb] = token for
 * sequence index t.  Uses the lists installed by cw_set_generation.                                               */
int32_t cw_test_sample(cw_ctx* ctx, int32_t nb, const float* logits, const int32_t* ids, int32_t t, int32_t n_prompt,
                       int32_t min_new_tokens, int32_t max_length, int32_t* choice_out);

/* ---- measurement -------------------------------------------------------------------------------------- */
#define CW_STAGE_MEL 0
#define CW_STAGE_ENCODER 1
#define CW_STAGE_CROSS_KV 2
#define CW_STAGE_DECODE 3
#define CW_STAGE_TIMESTAMPS 4
#define CW_N_STAGES 5
/* Accumulated HIP-event time per stage (ms) and number of timed invocations since the last reset.        */
int32_t cw_stage_times(cw_ctx* ctx, float* ms /* [CW_N_STAGES] */, int32_t* calls /* [CW_N_STAGES] */, int32_t reset);
/* Times `iters` back-to-back launches of one decode-step kernel on the context's stream with HIP events:
 * which = 0 decode GEMV (fc1 of decoder layer 0, LN fused), 1 cross-attention decode (layer 0).
 * Returns average ms per launch and the algorithmic bytes one launch must move.                          */
int32_t cw_time_kernel(cw_ctx* ctx, int32_t which, int32_t nb, int32_t iters, float* avg_ms, double* algo_bytes);
/* Times ONE launch of the decoder layer exactly as the decode step issues it for `nb` greedy rows (the step's own launch code with
 * the step's arguments; the layers are cycled so that every launch streams its operands from HBM).  stage = index of the launch
 * inside the layer, 0 .. *n_stages - 1 (the count depends on rows, dtype and cache mode); stage = -1 times the whole layer.
 * Returns average ms per launch, the algorithmic bytes it must move, and its kind (cw_decode_stage_name).  Measurement aid of
 * bench.py's roofline block: the reference has no counterpart (it times the pipeline call, /root/reference/transcribe.py:33). */
int32_t cw_time_decode_stage(cw_ctx* ctx, int32_t nb, int32_t stage, int32_t iters, float* avg_ms, double* algo_bytes,
                             int32_t* kind, int32_t* n_stages);
const char* cw_decode_stage_name(int32_t kind);
/* Kernel launches behind stage `stage` of the layer last enumerated by cw_time_decode_stage (2 at 17..64 rows where a preparation
 * launch precedes the GEMV); 0 for an unknown stage. */
int32_t cw_decode_stage_launches(cw_ctx* ctx, int32_t stage);
/* Number of calls this context repeated on the launch-per-stage decoder kernels because blocks of one launch waited for each other
 * in vain (only when the GPU is shared with other work; 0 in normal operation).  After the first one the context stays on those
 * kernels; results are identical either way. */
int32_t cw_handoff_fallbacks(cw_ctx* ctx);
/* ... of which cw_decode did not start over: the decoder kernels record the position of the first forward that ran on a missed
 * hand-off, the host sees it with the one-step lag of its "rows still running" read, rebuilds the sampler's per-row state from the
 * token ids (/root/reference has no counterpart; the state is that of transformers' WhisperTimeStampLogitsProcessor +
 * stopping criteria, generation_whisper.py / logits_process.py:1933-2048) and resumes at that position.  The whole call is
 * repeated instead while a logprob threshold is set (its running sums cannot be rebuilt from ids). */
int32_t cw_handoff_resumes(cw_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
