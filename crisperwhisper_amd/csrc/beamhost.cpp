// Host half of beam search: running / finished hypotheses of GenerationMixin._beam_search
// (TF/generation/utils.py:3208-3520; candidate union :3147 / :3436, running beams :3173-3190, finished hypotheses :3192-3245,
// early-stopping heuristic :3009-3053), in float32 like HF.  One call per decoder step between cw_beam_step (per-row best
// candidates from the device) and cw_beam_advance (cache re-ordering on the device): the 45 small numpy calls this replaces took
// 0.23 ms per step next to a 2.4 ms decoder forward.  Host-only C++ (no HIP): runs on any box, the CPU tests hold it bit-equal to
// the numpy statement of the same steps (crisperwhisper_amd/generation.py: beam_search(..., native_host=False)) on random
// candidate streams.  Built without floating-point contraction: every addition / division below is one IEEE binary32 operation,
// in numpy's order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>
#include <vector>

#include "../../include/crisperwhisper.h"

// the library's errno-style codes (common.h; that header is HIP-only)
#define CW_OK 0
#define CW_ERR_INVALID (-22)
#define CW_ERR_STATE (-1)

struct cw_beam_host {
    int B, K, n_prompt, max_length, V, eos, keep, cur_len;
    double length_penalty;                         // a Python float on the reference side: the exponent stays binary64
    bool early_stopping;
    int gen_cap;                                   // max_length - n_prompt
    std::vector<int64_t> running_seq, sequences;  // [B][K][max_length]
    std::vector<int32_t> running_bi, beam_indices;  // [B][K][gen_cap]
    std::vector<float> running_scores, beam_scores;  // [B][K]
    std::vector<uint8_t> finished;                 // [B][K]
    std::vector<uint8_t> unsat;                    // [B]
};

extern "C" {

cw_beam_host* cw_beam_host_new(int32_t n_items, int32_t num_beams, int32_t n_prompt, int32_t max_length, int32_t vocab_size,
                               int32_t eos_token_id, int32_t pad_token_id, double length_penalty, int32_t early_stopping,
                               const int32_t* prompt) {
    if (n_items < 1 || num_beams < 1 || n_prompt < 1 || max_length <= n_prompt || vocab_size < 1 || !prompt) return nullptr;
    cw_beam_host* s = new (std::nothrow) cw_beam_host();
    if (!s) return nullptr;
    s->B = n_items; s->K = num_beams; s->n_prompt = n_prompt; s->max_length = max_length; s->V = vocab_size;
    s->eos = eos_token_id; s->keep = 2 * num_beams; s->cur_len = n_prompt;   // beams_to_keep = max(2, 1 + n_eos_tokens) * K (:3280)
    s->length_penalty = length_penalty; s->early_stopping = early_stopping != 0;
    s->gen_cap = max_length - n_prompt;
    const int64_t fill = pad_token_id ? pad_token_id : eos_token_id;          // output_fill_value (:3323)
    const size_t BK = (size_t)n_items * num_beams;
    s->running_seq.assign(BK * max_length, fill);
    for (int b = 0; b < n_items; ++b)
        for (int k = 0; k < num_beams; ++k)
            for (int t = 0; t < n_prompt; ++t) s->running_seq[((size_t)b * num_beams + k) * max_length + t] = prompt[(size_t)b * n_prompt + t];
    s->sequences = s->running_seq;
    s->running_bi.assign(BK * s->gen_cap, -1);
    s->beam_indices = s->running_bi;
    s->running_scores.assign(BK, -1.0e9f);
    for (int b = 0; b < n_items; ++b) s->running_scores[(size_t)b * num_beams] = 0.0f;
    s->beam_scores.assign(BK, -1.0e9f);
    s->finished.assign(BK, 0);
    s->unsat.assign(n_items, 1);
    return s;
}

void cw_beam_host_free(cw_beam_host* s) { delete s; }

// indices of the k largest values, largest first, ties towards the lower index (torch.topk on CPU = stable argsort of -x)
static void topk_desc(const float* v, int n, int k, int* out, std::vector<int>& idx) {
    idx.resize(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return v[a] > v[b]; });
    for (int i = 0; i < k; ++i) out[i] = idx[i];
}

// cand_logprob / cand_token: [n_items * num_beams][2 * num_beams] from cw_beam_step (best first; -inf / -1 padded).
// parent / token: [n_items * num_beams] for cw_beam_advance.  Returns 1 = go on, 0 = the search is over, < 0 = error.
int32_t cw_beam_host_step(cw_beam_host* s, const float* cand_logprob, const int32_t* cand_token, int32_t* parent,
                          int32_t* token) {
    if (!s || !cand_logprob || !cand_token || !parent || !token) return CW_ERR_INVALID;
    if (s->cur_len >= s->max_length) return CW_ERR_STATE;
    const int B = s->B, K = s->K, keep = s->keep, ML = s->max_length, GC = s->gen_cap, n = K * keep;
    // the candidates index the vocabulary and order the hypotheses: a token past the vocabulary would address a beam that does
    // not exist, a NaN has no place in the order -- refuse before any state changes
    for (size_t i = 0; i < (size_t)B * n; ++i)
        if (cand_token[i] >= s->V || cand_logprob[i] != cand_logprob[i]) return CW_ERR_INVALID;
    const int cur = s->cur_len, gpos = cur - s->n_prompt;
    const float NEG = -1.0e9f, NINF = -std::numeric_limits<float>::infinity();
    std::vector<float> acc(n), topk_lp(keep), run_lp(keep), lp2(keep), m_scores(K + keep);
    std::vector<int64_t> flat(n), topk_seq((size_t)keep * ML), new_seq((size_t)K * ML);
    std::vector<int32_t> topk_bi((size_t)keep * GC), new_bi((size_t)K * GC);
    std::vector<int> order(n), nxt(K), sel(K), tmp;
    std::vector<uint8_t> hits(keep), did_top(keep), new_fin(K);
    std::vector<float> new_scores(K);
    bool all_hits = true;
    const float denom = (float)std::pow((double)(cur + 1 - s->n_prompt), s->length_penalty);
    for (int b = 0; b < B; ++b) {
        float* rs = &s->running_scores[(size_t)b * K];
        float* bs = &s->beam_scores[(size_t)b * K];
        uint8_t* fin = &s->finished[(size_t)b * K];
        int64_t* rseq = &s->running_seq[(size_t)b * K * ML];
        int64_t* fseq = &s->sequences[(size_t)b * K * ML];
        int32_t* rbi = &s->running_bi[(size_t)b * K * GC];
        int32_t* fbi = &s->beam_indices[(size_t)b * K * GC];
        // candidate union of the item's beams in (value desc, flattened vocabulary index asc) order: torch.topk over [K * V] (:3147)
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < keep; ++j) {
                const int i = k * keep + j;
                const int32_t tk = cand_token[((size_t)b * K + k) * keep + j];
                const float a = cand_logprob[((size_t)b * K + k) * keep + j] + rs[k];                 // :3436
                acc[i] = tk >= 0 ? a : NINF;
                flat[i] = (int64_t)k * s->V + (tk > 0 ? tk : 0);
            }
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            if (acc[x] != acc[y]) return acc[x] > acc[y];
            return flat[x] < flat[y];
        });
        bool all_fin_before = true;
        for (int k = 0; k < K; ++k) all_fin_before = all_fin_before && fin[k];
        const bool full = all_fin_before && s->early_stopping;
        for (int j = 0; j < keep; ++j) {
            const int i = order[j];
            topk_lp[j] = acc[i];
            const int beam = (int)(flat[i] / s->V);
            const int64_t id = flat[i] % s->V;
            std::memcpy(&topk_seq[(size_t)j * ML], rseq + (size_t)beam * ML, sizeof(int64_t) * ML);
            topk_seq[(size_t)j * ML + cur] = id;
            std::memcpy(&topk_bi[(size_t)j * GC], rbi + (size_t)beam * GC, sizeof(int32_t) * GC);
            topk_bi[(size_t)j * GC + gpos] = beam + b * K;
            hits[j] = (id == s->eos) || (cur + 1 >= ML);                                              // :3456-3462
            all_hits = all_hits && hits[j];
            run_lp[j] = hits[j] ? topk_lp[j] + NEG : topk_lp[j];                                      // :3173-3190
            did_top[j] = hits[j] && j < K;
            float v = topk_lp[j] / denom;                                                             // :3192-3245
            if (full) v = v + NEG;
            if (!s->unsat[b]) v = v + NEG;
            if (!did_top[j]) v = v + NEG;
            lp2[j] = v;
        }
        // finished hypotheses first (they read the old `sequences`), then the running beams of the next iteration
        for (int k = 0; k < K; ++k) m_scores[k] = bs[k];
        for (int j = 0; j < keep; ++j) m_scores[K + j] = lp2[j];
        topk_desc(m_scores.data(), K + keep, K, sel.data(), tmp);
        for (int k = 0; k < K; ++k) {
            const int m = sel[k];
            const int64_t* src = m < K ? fseq + (size_t)m * ML : &topk_seq[(size_t)(m - K) * ML];
            const int32_t* sbi = m < K ? fbi + (size_t)m * GC : &topk_bi[(size_t)(m - K) * GC];
            std::memcpy(&new_seq[(size_t)k * ML], src, sizeof(int64_t) * ML);
            std::memcpy(&new_bi[(size_t)k * GC], sbi, sizeof(int32_t) * GC);
            new_scores[k] = m_scores[m];
            new_fin[k] = m < K ? fin[m] : did_top[m - K];
        }
        std::memcpy(fseq, new_seq.data(), sizeof(int64_t) * K * ML);
        std::memcpy(fbi, new_bi.data(), sizeof(int32_t) * K * GC);
        for (int k = 0; k < K; ++k) { bs[k] = new_scores[k]; fin[k] = new_fin[k]; }
        topk_desc(run_lp.data(), keep, K, nxt.data(), tmp);
        for (int k = 0; k < K; ++k) {
            std::memcpy(rseq + (size_t)k * ML, &topk_seq[(size_t)nxt[k] * ML], sizeof(int64_t) * ML);
            std::memcpy(rbi + (size_t)k * GC, &topk_bi[(size_t)nxt[k] * GC], sizeof(int32_t) * GC);
            new_scores[k] = run_lp[nxt[k]];
        }
        for (int k = 0; k < K; ++k) {
            rs[k] = new_scores[k];
            parent[(size_t)b * K + k] = rbi[(size_t)k * GC + gpos];
            token[(size_t)b * K + k] = (int32_t)rseq[(size_t)k * ML + cur];
        }
    }
    s->cur_len = cur + 1;
    // stopping condition of the search as a whole (:3009-3053, early_stopping False / True; "never" is not offered)
    const float denom2 = (float)std::pow((double)(s->cur_len - s->n_prompt), s->length_penalty);
    bool any_unsat = false, all_fin = true;
    for (int b = 0; b < B; ++b) {
        const float* bs = &s->beam_scores[(size_t)b * K];
        const uint8_t* fin = &s->finished[(size_t)b * K];
        const float best_possible = s->running_scores[(size_t)b * K] / denom2;
        float mn = bs[0];
        for (int k = 1; k < K; ++k) mn = std::min(mn, bs[k]);
        bool improve = false;
        for (int k = 0; k < K; ++k) {
            const float worst = fin[k] ? mn : NEG;
            improve = improve || (best_possible > worst);
            all_fin = all_fin && fin[k];
        }
        s->unsat[b] = s->unsat[b] && improve;
        any_unsat = any_unsat || s->unsat[b];
    }
    const bool go_on = any_unsat && !(all_fin && s->early_stopping) && !all_hits;
    return go_on ? 1 : 0;
}

// best hypothesis of every item: sequences [n_items][max_length] (filled with the pad / eos id behind the end), beam_indices
// [n_items][max_length - n_prompt] (-1 behind the end), score [n_items]
int32_t cw_beam_host_result(const cw_beam_host* s, int64_t* sequences, int32_t* beam_indices, float* score) {
    if (!s || !sequences || !beam_indices || !score) return CW_ERR_INVALID;
    for (int b = 0; b < s->B; ++b) {
        std::memcpy(sequences + (size_t)b * s->max_length, &s->sequences[(size_t)b * s->K * s->max_length], sizeof(int64_t) * s->max_length);
        std::memcpy(beam_indices + (size_t)b * s->gen_cap, &s->beam_indices[(size_t)b * s->K * s->gen_cap], sizeof(int32_t) * s->gen_cap);
        score[b] = s->beam_scores[(size_t)b * s->K];
    }
    return CW_OK;
}

}  // extern "C"
