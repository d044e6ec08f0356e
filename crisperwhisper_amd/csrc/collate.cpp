// Host-side word collation (no GPU): same observable behaviour as
// WhisperTokenizer._decode_asr(..., return_timestamps="word") -- TF/models/whisper/tokenization_whisper.py:901-1150 --
// and its helpers (:1153-1406): stride-aware time offsets, timestamps deferred inside strides, longest-common
// token-subsequence seam merge constrained by timestamp order, unicode/space word grouping, punctuation merging,
// 0.01 s rounding.  Python-str semantics (code-point indexing, str.strip(), substring `in`, round(x, 2)) are
// reproduced explicitly; byte-level BPE text = concatenated token bytes decoded as UTF-8 with one U+FFFD per
// maximal ill-formed subsequence (what both CPython's errors="replace" and tokenizers' ByteLevel decoder do).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/crisperwhisper.h"

typedef std::u32string U32;

namespace {

const char32_t REPL = 0xFFFD;

// UTF-8 -> code points, Unicode 15 table 3-7 well-formedness, maximal-subpart replacement.
void utf8_decode_replace(const std::string& s, U32& out) {
    out.clear();
    const unsigned char* p = (const unsigned char*)s.data();
    size_t n = s.size(), i = 0;
    while (i < n) {
        unsigned char b0 = p[i];
        if (b0 < 0x80) { out.push_back(b0); ++i; continue; }
        int need = 0; unsigned char lo = 0x80, hi = 0xBF; char32_t cp = 0;
        if (b0 >= 0xC2 && b0 <= 0xDF) { need = 1; cp = b0 & 0x1F; }
        else if (b0 == 0xE0) { need = 2; lo = 0xA0; cp = b0 & 0x0F; }
        else if ((b0 >= 0xE1 && b0 <= 0xEC) || b0 == 0xEE || b0 == 0xEF) { need = 2; cp = b0 & 0x0F; }
        else if (b0 == 0xED) { need = 2; hi = 0x9F; cp = b0 & 0x0F; }
        else if (b0 == 0xF0) { need = 3; lo = 0x90; cp = b0 & 0x07; }
        else if (b0 >= 0xF1 && b0 <= 0xF3) { need = 3; cp = b0 & 0x07; }
        else if (b0 == 0xF4) { need = 3; hi = 0x8F; cp = b0 & 0x07; }
        else { out.push_back(REPL); ++i; continue; }          // invalid start byte
        size_t j = i + 1;
        bool ok = true;
        for (int k = 0; k < need; ++k, ++j) {
            if (j >= n) { ok = false; break; }                  // unexpected end of data
            unsigned char b = p[j];
            unsigned char l = (k == 0) ? lo : 0x80, h = (k == 0) ? hi : 0xBF;
            if (b < l || b > h) { ok = false; break; }          // invalid continuation byte
            cp = (cp << 6) | (b & 0x3F);
        }
        if (ok) { out.push_back(cp); i = j; }
        else { out.push_back(REPL); i = j; }                    // consumes the valid prefix (>= 1 byte)
    }
}

void utf8_encode(const U32& s, std::string& out) {
    out.clear();
    for (char32_t c : s) {
        if (c < 0x80) out.push_back((char)c);
        else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
        else if (c < 0x10000) { out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
        else { out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
    }
}

bool py_isspace(char32_t c) {   // str.isspace() / _PyUnicode_IsWhitespace
    return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
           (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
U32 py_strip(const U32& s) {
    size_t a = 0, b = s.size();
    while (a < b && py_isspace(s[a])) ++a;
    while (b > a && py_isspace(s[b - 1])) --b;
    return s.substr(a, b - a);
}
bool py_in(const U32& needle, const U32& hay) { return needle.empty() || hay.find(needle) != U32::npos; }

double py_round2(double x) {    // round(x, 2): correctly rounded decimal, like float.__round__
    if (!isfinite(x)) return x;
    char buf[64];
    snprintf(buf, sizeof(buf), "%.2f", x);
    return strtod(buf, nullptr);
}

const U32 PUNCT = U"!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~";
const U32 PREPEND = U"\"'\u201c\u00a1\u00bf([{-";
const U32 APPEND = U"\"'.\u3002,\uff0c!\uff01?\uff1f:\uff1a\u201d)]}\u3001";

typedef std::pair<double, double> Span;

}  // namespace

struct cw_vocab {
    std::vector<std::string> bytes;      // raw bytes per token ("" for non-text)
    std::vector<signed char> kind;       // 0 text, 1 special, 2 other (timestamps / holes)
    std::vector<signed char> lang_class; // for specials: -1 not a language, 0 space-separated, 1 no-space language
    int eos, tb, startofprev, sot;
    int default_lang_class;              // 0 / 1

    void text(const std::vector<int>& ids, U32& out) const {
        std::string b;
        for (int id : ids)
            if (id >= 0 && id < (int)bytes.size() && kind[id] == 0) b += bytes[id];
        utf8_decode_replace(b, out);
    }
};

struct Word { U32 text; double start, end; };

struct cw_collator {
    const cw_vocab* v;
    double tp;
    int segment_size;
    std::vector<Word> words;
    U32 full_text;
    int lang_class;                      // -1 unknown
    double time_offset;
    std::vector<std::vector<int>> pending;
    std::vector<std::vector<Span>> pending_ts;
    bool has_open; double open_start;
    bool skip;
    bool warned;
    int mode;                            // 0: word chunks (return_timestamps="word"), 1: one chunk per timestamp-delimited segment (True)
    std::string err;

    void merge_overlapping(std::vector<int>& toks, std::vector<Span>& ts) {
        std::vector<int> left = pending[0];
        std::vector<Span> left_ts = pending_ts[0];
        toks.clear(); ts.clear();
        for (size_t si = 1; si < pending.size(); ++si) {
            const std::vector<int>& right = pending[si];
            const std::vector<Span>& right_ts = pending_ts[si];
            const int nl = (int)left.size(), nr = (int)right.size();
            double best = 0.0;
            int wl0 = nl, wl1 = nl, wr0 = 0, wr1 = 0;
            for (int span = 1; span < nl + nr; ++span) {
                const int l0 = nl - span > 0 ? nl - span : 0, l1 = nl < nl + nr - span ? nl : nl + nr - span;
                const int r0 = span - nl > 0 ? span - nl : 0, r1 = nr < span ? nr : span;
                int hits = 0;
                for (int k = 0; k < l1 - l0; ++k)
                    if (left[l0 + k] == right[r0 + k] && (mode != 0 || left_ts[l0 + k] <= right_ts[r0 + k])) ++hits;
                const double score = (double)hits / (double)span + (double)span / 10000.0;
                if (hits > 1 && score > best) { best = score; wl0 = l0; wl1 = l1; wr0 = r0; wr1 = r1; }
            }
            const int lmid = (wl1 + wl0) / 2, rmid = (wr1 + wr0) / 2;
            toks.insert(toks.end(), left.begin(), left.begin() + lmid);
            ts.insert(ts.end(), left_ts.begin(), left_ts.begin() + lmid);
            left.assign(right.begin() + rmid, right.end());
            left_ts.assign(right_ts.begin() + rmid, right_ts.end());
        }
        toks.insert(toks.end(), left.begin(), left.end());
        ts.insert(ts.end(), left_ts.begin(), left_ts.end());
    }

    // end_time: the closing timestamp of the segment (NaN: Whisper predicted none), used by the segment mode only
    void flush(double end_time) {
        std::vector<int> toks; std::vector<Span> ts;
        merge_overlapping(toks, ts);
        U32 whole;
        v->text(toks, whole);
        full_text += whole;
        if (mode == 1) {                                         // :1060-1075 without the word collation
            Word w; w.text = whole; w.start = has_open ? open_start : NAN; w.end = end_time;
            words.push_back(w);
            pending.clear(); pending_ts.clear();
            has_open = false;
            return;
        }
        // unicode pieces (:1315-1344)
        std::vector<U32> pieces; std::vector<std::vector<int>> pidx;
        {
            std::vector<int> cur, curi; size_t consumed = 0; U32 s;
            for (int k = 0; k < (int)toks.size(); ++k) {
                cur.push_back(toks[k]); curi.push_back(k);
                v->text(cur, s);
                size_t p = s.find(REPL);
                if (p == U32::npos || consumed + p >= whole.size() || whole[consumed + p] == REPL) {
                    pieces.push_back(s); pidx.push_back(curi);
                    consumed += s.size();
                    cur.clear(); curi.clear();
                }
            }
        }
        const int lc = lang_class >= 0 ? lang_class : v->default_lang_class;
        std::vector<U32> ws; std::vector<std::vector<int>> wi;
        if (lc == 1) { ws = pieces; wi = pidx; }
        else {                                                   // :1347-1368
            for (size_t k = 0; k < pieces.size(); ++k) {
                const U32& s = pieces[k];
                bool starts = toks[pidx[k][0]] >= v->eos || (!s.empty() && s[0] == U' ') || py_in(py_strip(s), PUNCT) || ws.empty();
                if (starts) { ws.push_back(s); wi.push_back(pidx[k]); }
                else { ws.back() += s; wi.back().insert(wi.back().end(), pidx[k].begin(), pidx[k].end()); }
            }
        }
        // punctuation merging (:1371-1406)
        {
            int j = (int)ws.size() - 1;
            for (int i = (int)ws.size() - 2; i >= 0; --i) {
                if (!ws[i].empty() && ws[i][0] == U' ' && py_in(py_strip(ws[i]), PREPEND)) {
                    ws[j] = ws[i] + ws[j];
                    std::vector<int> t = wi[i]; t.insert(t.end(), wi[j].begin(), wi[j].end()); wi[j] = t;
                    ws[i].clear(); wi[i].clear();
                } else j = i;
            }
            int i = 0;
            for (int jj = 1; jj < (int)ws.size(); ++jj) {
                const bool ends_space = !ws[i].empty() && ws[i].back() == U' ';
                if (!ends_space && py_in(ws[jj], APPEND)) {
                    ws[i] += ws[jj];
                    wi[i].insert(wi[i].end(), wi[jj].begin(), wi[jj].end());
                    ws[jj].clear(); wi[jj].clear();
                } else i = jj;
            }
        }
        for (size_t k = 0; k < ws.size(); ++k) {
            if (ws[k].empty()) continue;
            Word w; w.text = ws[k]; w.start = ts[wi[k].front()].first; w.end = ts[wi[k].back()].second;
            words.push_back(w);
        }
        pending.clear(); pending_ts.clear();
        has_open = false;
    }
};

extern "C" {

cw_vocab* cw_vocab_create(int32_t n_tokens, const uint8_t* blob, const int64_t* offsets, const int8_t* kind,
                          const int8_t* lang_class, int32_t eos, int32_t timestamp_begin, int32_t startofprev,
                          int32_t sot, int32_t default_lang_class) {
    if (n_tokens <= 0 || !blob || !offsets || !kind || !lang_class) return nullptr;
    cw_vocab* v = new cw_vocab();
    v->bytes.resize(n_tokens); v->kind.resize(n_tokens); v->lang_class.resize(n_tokens);
    for (int i = 0; i < n_tokens; ++i) {
        v->bytes[i].assign((const char*)blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
        v->kind[i] = kind[i]; v->lang_class[i] = lang_class[i];
    }
    v->eos = eos; v->tb = timestamp_begin; v->startofprev = startofprev; v->sot = sot;
    v->default_lang_class = default_lang_class;
    return v;
}
void cw_vocab_destroy(cw_vocab* v) { delete v; }

cw_collator* cw_collate_begin(const cw_vocab* v, double time_precision) {
    if (!v) return nullptr;
    cw_collator* c = new cw_collator();
    c->v = v; c->tp = time_precision; c->segment_size = 1500; c->lang_class = -1; c->time_offset = 0.0;
    c->has_open = false; c->open_start = 0.0; c->skip = false; c->warned = false; c->mode = 0;
    return c;
}
int32_t cw_collate_set_mode(cw_collator* c, int32_t mode) {
    if (!c || mode < 0 || mode > 1) return -22;
    c->mode = mode;
    return 0;
}
void cw_collate_free(cw_collator* c) { delete c; }

int32_t cw_collate_feed(cw_collator* c, const int64_t* tokens, int32_t n_tokens, const float* token_ts, int32_t n_ts,
                        int32_t has_stride, double chunk_len, double stride_left, double stride_right) {
    const cw_vocab* v = c->v;
    const int tb = v->tb; const double tp = c->tp;
    std::vector<int> ids(tokens, tokens + n_tokens);
    if (!ids.empty() && v->startofprev >= 0 && ids[0] == v->startofprev) {       // _strip_prompt (:725-743)
        size_t k = 0;
        while (k < ids.size() && ids[k] != v->sot) ++k;
        if (k < ids.size()) ids.erase(ids.begin(), ids.begin() + k); else ids.clear();
    }
    bool has_deferred = false; int deferred_from = 0;
    double first_ts = (double)tb;
    if (has_stride) {
        c->time_offset -= stride_left;
        const double right_start = chunk_len - stride_right;
        if (stride_left != 0.0) first_ts = stride_left / tp + tb;
        if (stride_right != 0.0) {
            for (int k = (int)ids.size() - 1; k >= 0; --k) {
                const int t = ids[k];
                if (t >= tb) {
                    if (has_deferred && (double)(t - tb) * tp < right_start) break;
                    deferred_from = t; has_deferred = true;
                }
            }
        }
    }
    std::vector<int> cur; std::vector<Span> cur_ts;
    double cur_max = 0.0, prev_len = 0.0, penult = 0.0;
    for (int i = 0; i < (int)ids.size(); ++i) {
        const int t = ids[i];
        const int kind = (t >= 0 && t < (int)v->kind.size()) ? v->kind[t] : 2;
        if (kind == 1) {
            if (v->lang_class[t] >= 0) c->lang_class = v->lang_class[t];
        } else if (t >= tb) {
            const double stamp = (double)(t - tb) * tp;
            if (stamp < cur_max) {
                const bool single_end = i >= 2 && !(ids[i - 1] >= tb && ids[i - 2] >= tb);
                if (single_end) prev_len += tp * c->segment_size;
                else { cur_max = penult; prev_len += penult; }
            }
            penult = cur_max;
            cur_max = stamp;
            const double when = py_round2((double)(t - tb) * tp + c->time_offset + prev_len);
            if (has_deferred && deferred_from != 0 && t >= deferred_from) c->skip = true;
            else if (c->skip || (!c->pending.empty() && (double)t < first_ts)) c->skip = false;
            else if (!c->has_open) { c->has_open = true; c->open_start = when; }
            else if (when != c->open_start) {
                c->pending.push_back(cur); c->pending_ts.push_back(cur_ts);
                c->flush(when);
                cur.clear(); cur_ts.clear();
            }
        } else {
            cur.push_back(t);
            if (c->mode == 1) { cur_ts.push_back(Span(0.0, 0.0)); continue; }
            if (i >= n_ts) { c->err = "token_timestamps shorter than tokens"; return -22; }
            const double begin = (i == 0) ? py_round2(0.0 + c->time_offset) : py_round2((double)token_ts[i - 1] + c->time_offset);
            cur_ts.push_back(Span(begin, py_round2((double)token_ts[i] + c->time_offset)));
        }
    }
    if (has_stride) c->time_offset += chunk_len - stride_right;
    if (!cur.empty()) { c->pending.push_back(cur); c->pending_ts.push_back(cur_ts); }
    else {
        bool any = false;
        for (auto& p : c->pending) if (!p.empty()) any = true;
        if (!any) { c->pending.clear(); c->pending_ts.clear(); c->has_open = false; }
    }
    return 0;
}

int32_t cw_collate_finish(cw_collator* c, int32_t* n_words, int64_t* text_bytes, int64_t* words_bytes, int32_t* warned) {
    if (!c->pending.empty()) { c->warned = true; c->flush(NAN); }
    std::string s;
    utf8_encode(c->full_text, s);
    int64_t wb = 0;
    for (auto& w : c->words) { std::string t; utf8_encode(w.text, t); wb += (int64_t)t.size(); }
    *n_words = (int32_t)c->words.size(); *text_bytes = (int64_t)s.size(); *words_bytes = wb; *warned = c->warned ? 1 : 0;
    return 0;
}

int32_t cw_collate_get(cw_collator* c, uint8_t* text, double* starts, double* ends, int64_t* word_offsets, uint8_t* words_blob) {
    std::string s;
    utf8_encode(c->full_text, s);
    memcpy(text, s.data(), s.size());
    int64_t off = 0;
    for (size_t k = 0; k < c->words.size(); ++k) {
        std::string t; utf8_encode(c->words[k].text, t);
        word_offsets[k] = off;
        memcpy(words_blob + off, t.data(), t.size());
        off += (int64_t)t.size();
        starts[k] = c->words[k].start; ends[k] = c->words[k].end;
    }
    word_offsets[c->words.size()] = off;
    return 0;
}

}  // extern "C"
