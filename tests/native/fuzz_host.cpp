// Sanitizer fuzz driver for the host-only half of the C ABI (csrc/collate.cpp, csrc/flac.cpp): built by
// tests/test_native_sanitizers.py with g++ -fsanitize=address,undefined and run on (a) mutated copies of valid FLAC
// streams given on the command line, (b) random byte strings, (c) random token / timestamp / stride streams through the
// word collator in both modes.  Exit code 0 = no sanitizer report and every call returned (errors are fine, crashes are not).
// usage: fuzz_host <iterations> <seed> [valid.flac ...]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/crisperwhisper.h"

static uint64_t g_state = 1;
static uint32_t rnd() {   // splitmix64 -> 32 bits
    uint64_t z = (g_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}
static uint32_t below(uint32_t n) { return n ? rnd() % n : 0; }

static std::vector<uint8_t> read_file(const char* path) {
    std::vector<uint8_t> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n > 0 ? (size_t)n : 0);
    if (n > 0 && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

static long g_flac_ok = 0, g_flac_err = 0;
static void try_flac(const std::vector<uint8_t>& d) {
    int32_t sr = 0, ch = 0, bps = 0;
    int64_t total = 0;
    const int32_t r = cw_flac_info(d.data(), (int64_t)d.size(), &sr, &ch, &bps, &total);
    // capacity: what the header announces (bounded), or a small fixed buffer when it lies / says "unknown"
    int64_t cap = (r == 0 && total > 0 && total < (1 << 20)) ? total : 4096;
    if (ch < 1 || ch > 8) ch = 8;
    std::vector<int32_t> pcm((size_t)cap * (size_t)ch);
    int64_t got = 0;
    const int32_t r2 = cw_flac_decode(d.data(), (int64_t)d.size(), pcm.data(), cap, &got);
    if (r2 == 0) { ++g_flac_ok; if (got < 0 || got > cap) { fprintf(stderr, "flac: %lld frames into a %lld-frame buffer\n", (long long)got, (long long)cap); exit(3); } }
    else { ++g_flac_err; (void)cw_flac_last_error(); }
}

static void fuzz_flac(const std::vector<std::vector<uint8_t>>& seeds, int iters) {
    for (const auto& s : seeds) try_flac(s);   // the valid streams themselves must decode
    if (g_flac_ok != (long)seeds.size()) { fprintf(stderr, "a seed stream did not decode\n"); exit(4); }
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> d;
        const uint32_t kind = below(10);
        if (seeds.empty() || kind == 0) {                       // pure noise, sometimes with the magic in front
            d.resize(below(600));
            for (auto& b : d) b = (uint8_t)rnd();
            if (d.size() >= 4 && below(2)) memcpy(d.data(), "fLaC", 4);
        } else {
            d = seeds[below((uint32_t)seeds.size())];
            const uint32_t m = below(6);
            if (m == 0 && !d.empty()) d.resize(below((uint32_t)d.size()));                    // truncation
            else if (m == 1) { for (int k = 0, n = 1 + (int)below(8); k < n && !d.empty(); ++k) d[below((uint32_t)d.size())] ^= (uint8_t)(1u << below(8)); }
            else if (m == 2) { for (int k = 0, n = 1 + (int)below(16); k < n && !d.empty(); ++k) d[below((uint32_t)d.size())] = (uint8_t)rnd(); }
            else if (m == 3 && d.size() > 8) { const uint32_t a = below((uint32_t)d.size()), b = below((uint32_t)d.size()), n = below(64);   // splice
                                                for (uint32_t k = 0; k < n && a + k < d.size() && b + k < d.size(); ++k) d[a + k] = d[b + k]; }
            else if (m == 4 && d.size() > 42) { for (int k = 0; k < 4; ++k) d[4 + below(38)] = (uint8_t)rnd(); }                           // STREAMINFO lies
            else if (!d.empty()) { const uint32_t a = below((uint32_t)d.size()); d.insert(d.begin() + a, (size_t)below(32), (uint8_t)rnd()); }
        }
        try_flac(d);
    }
}

// ---- collator ------------------------------------------------------------------------------------------------
struct Vocab {
    cw_vocab* v = nullptr;
    int n = 0, eos = 0, tb = 0, sot = 0, prev = 0;
};
static Vocab make_vocab() {
    // 256 single-byte tokens, a few multi-byte pieces (valid and ill-formed UTF-8), specials, 1501 timestamp tokens
    std::vector<std::string> toks;
    std::vector<int8_t> kind, lang;
    for (int b = 0; b < 256; ++b) { toks.push_back(std::string(1, (char)b)); kind.push_back(0); lang.push_back(-1); }
    const char* extra[] = {" the", " a", "ing", ",", ".", " \xc3\xa9t\xc3\xa9", "\xe4\xb8\xad", "\xe6\x96", "\x87", " ", "!?", "'s", "\xf0\x9f\x98", "\x80"};
    for (const char* e : extra) { toks.push_back(e); kind.push_back(0); lang.push_back(-1); }
    Vocab V;
    V.eos = (int)toks.size(); toks.push_back("<|endoftext|>"); kind.push_back(1); lang.push_back(-1);
    V.sot = (int)toks.size(); toks.push_back("<|startoftranscript|>"); kind.push_back(1); lang.push_back(-1);
    toks.push_back("<|en|>"); kind.push_back(1); lang.push_back(0);
    toks.push_back("<|zh|>"); kind.push_back(1); lang.push_back(1);
    toks.push_back("<|transcribe|>"); kind.push_back(1); lang.push_back(-1);
    V.prev = (int)toks.size(); toks.push_back("<|startofprev|>"); kind.push_back(1); lang.push_back(-1);
    toks.push_back("<|notimestamps|>"); kind.push_back(1); lang.push_back(-1);
    V.tb = (int)toks.size();
    for (int i = 0; i <= 1500; ++i) { char buf[32]; snprintf(buf, sizeof buf, "<|%.2f|>", i * 0.02); toks.push_back(buf); kind.push_back(2); lang.push_back(-1); }
    std::vector<uint8_t> blob;
    std::vector<int64_t> off(1, 0);
    for (const auto& t : toks) { blob.insert(blob.end(), t.begin(), t.end()); off.push_back((int64_t)blob.size()); }
    V.n = (int)toks.size();
    V.v = cw_vocab_create(V.n, blob.data(), off.data(), kind.data(), lang.data(), V.eos, V.tb, V.prev, V.sot, 0);
    return V;
}

static void fuzz_collate(const Vocab& V, int iters) {
    for (int it = 0; it < iters; ++it) {
        cw_collator* c = cw_collate_begin(V.v, below(8) ? 0.02 : 0.0);
        if (!c) { fprintf(stderr, "cw_collate_begin failed\n"); exit(5); }
        cw_collate_set_mode(c, (int32_t)below(3));                       // 2 is invalid on purpose
        const int chunks = 1 + (int)below(5);
        for (int k = 0; k < chunks; ++k) {
            const int n = (int)below(120);
            std::vector<int64_t> tok((size_t)n);
            std::vector<float> ts((size_t)n);
            float t = 0.f;
            for (int i = 0; i < n; ++i) {
                const uint32_t r = below(100);
                if (r < 60) tok[i] = (int64_t)below(256 + 14);
                else if (r < 85) tok[i] = V.tb + (int64_t)below(1501);
                else if (r < 95) tok[i] = V.eos + (int64_t)below(7);
                else tok[i] = (int64_t)below(4) == 0 ? -1 - (int64_t)below(10) : (int64_t)V.n + (int64_t)below(1000);   // out of range
                t += (below(10) ? 0.02f : -0.1f) * (float)below(12);
                ts[i] = below(50) ? t : (below(2) ? NAN : INFINITY);
            }
            const int n_ts = below(6) ? n : (int)below((uint32_t)n + 2);   // sometimes a length mismatch
            ts.resize((size_t)(n_ts > n ? n_ts : n), 0.f);
            const double len = below(4) ? 30.0 : (double)below(40) - 5.0;
            cw_collate_feed(c, tok.data(), n, ts.data(), n_ts, (int32_t)below(2), len, (double)below(8), (double)below(8));
        }
        int32_t n_words = 0, warned = 0;
        int64_t tb = 0, wb = 0;
        if (cw_collate_finish(c, &n_words, &tb, &wb, &warned) == 0 && n_words >= 0 && tb >= 0 && wb >= 0) {
            std::vector<uint8_t> text((size_t)tb + 1), words((size_t)wb + 1);
            std::vector<double> st((size_t)n_words + 1), en((size_t)n_words + 1);
            std::vector<int64_t> wo((size_t)n_words + 2);
            cw_collate_get(c, text.data(), st.data(), en.data(), wo.data(), words.data());
        }
        cw_collate_free(c);
    }
}

// host half of beam search (csrc/beamhost.cpp): random geometries and candidate streams, now and then a token past the
// vocabulary or a NaN (must be refused with the state untouched), results read back after every outcome
static long g_beam_runs = 0, g_beam_rejects = 0;
static void fuzz_beam_host(int iters) {
    for (int it = 0; it < iters; ++it) {
        const int B = 1 + (int)below(4), K = 1 + (int)below(6), n_prompt = 1 + (int)below(4), V = 2 + (int)below(200);
        const int max_length = n_prompt + 1 + (int)below(20), keep = 2 * K, rows = B * K;
        const int eos = (int)below((uint32_t)V), pad = below(3) ? (int)below((uint32_t)V) : 0;
        std::vector<int32_t> prompt((size_t)B * n_prompt);
        for (auto& p : prompt) p = (int32_t)below((uint32_t)V);
        if (below(20) == 0) {   // invalid geometry must give no object
            if (cw_beam_host_new(B, K, n_prompt, n_prompt, V, eos, pad, 1.0, 0, prompt.data())) { fprintf(stderr, "bad geometry accepted\n"); exit(7); }
            continue;
        }
        cw_beam_host* s = cw_beam_host_new(B, K, n_prompt, max_length, V, eos, pad, below(2) ? 1.0 : 0.1 * (double)below(30), (int32_t)below(2), prompt.data());
        if (!s) { fprintf(stderr, "cw_beam_host_new failed\n"); exit(7); }
        std::vector<float> val((size_t)rows * keep);
        std::vector<int32_t> tok((size_t)rows * keep), parent((size_t)rows), token((size_t)rows);
        ++g_beam_runs;
        for (int step = 0; step < max_length + 2; ++step) {     // two calls past the end: CW_ERR_STATE, not a write past max_length
            bool hostile = false;
            for (int r = 0; r < rows; ++r) {
                float v = -(float)below(8) * 0.25f;
                const int npad = below(6) ? 0 : (int)below((uint32_t)keep + 1);
                for (int j = 0; j < keep; ++j) {
                    v -= (float)below(4) * 0.25f;
                    const size_t i = (size_t)r * keep + j;
                    val[i] = v; tok[i] = below(8) == 0 ? eos : (int32_t)below((uint32_t)V);
                    if (j >= keep - npad) { val[i] = -INFINITY; tok[i] = -1; }
                    if (below(2000) == 0) { val[i] = NAN; hostile = true; }
                    if (below(2000) == 0) { tok[i] = V + (int32_t)below(1000); hostile = true; }
                }
            }
            const int32_t rc = cw_beam_host_step(s, val.data(), tok.data(), parent.data(), token.data());
            if (hostile) { if (rc >= 0) { fprintf(stderr, "hostile candidates accepted (%d)\n", rc); exit(7); } ++g_beam_rejects; continue; }
            if (rc < 0 && step < max_length - n_prompt) { fprintf(stderr, "cw_beam_host_step failed early (%d)\n", rc); exit(7); }
            if (rc == 1)
                for (int r = 0; r < rows; ++r)
                    if (parent[r] / K != r / K || token[r] < 0 || token[r] >= V) { fprintf(stderr, "parent / token out of range\n"); exit(7); }
            if (rc == 0 && below(2)) break;                      // sometimes keep calling after the search is over
        }
        std::vector<int64_t> seq((size_t)B * max_length);
        std::vector<int32_t> bi((size_t)B * (max_length - n_prompt));
        std::vector<float> score((size_t)B);
        if (cw_beam_host_result(s, seq.data(), bi.data(), score.data()) != 0) { fprintf(stderr, "cw_beam_host_result failed\n"); exit(7); }
        cw_beam_host_free(s);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    g_state = argc > 2 ? (uint64_t)atoll(argv[2]) : 1;
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; ++i) { auto v = read_file(argv[i]); if (!v.empty()) seeds.push_back(v); }
    fuzz_flac(seeds, iters);
    Vocab V = make_vocab();
    if (!V.v) { fprintf(stderr, "cw_vocab_create failed\n"); return 6; }
    fuzz_collate(V, iters / 4 + 1);
    cw_vocab_destroy(V.v);
    fuzz_beam_host(iters / 4 + 1);
    printf("fuzz_host: %ld FLAC streams decoded, %ld rejected, %d collator runs, %ld beam searches (%ld hostile steps refused), no sanitizer report\n",
           g_flac_ok, g_flac_err, iters / 4 + 1, g_beam_runs, g_beam_rejects);
    return 0;
}
