// Native FLAC decoder (host C++, no dependency): the step in front of the hot path for the container SURVEY.md 8(f).1 names
// beside WAV.  Replaces, for .flac input, the `ffmpeg -i file -ac 1 -ar 16000 -f f32le` subprocess of
// TF/pipelines/audio_utils.py:9-45 up to the decoded integer samples; mono mixdown, int -> float scaling and resampling
// stay on the device (cw_ingest, csrc/ingest.hip).
//
// Written from the FLAC format specification (RFC 9639): stream marker, metadata blocks (STREAMINFO read, all others
// skipped), frame header incl. UTF-8 coded frame / sample number, CRC-8 and CRC-16, the four subframe types (CONSTANT,
// VERBATIM, FIXED order 0..4, LPC order 1..32), wasted bits, Rice / Rice2 residual partitions with escape codes, the
// three stereo decorrelation modes, 4..32 bits per sample, and the MD5 signature of the decoded samples in STREAMINFO
// (checked when it is set: a decoder that disagrees with the encoder must fail loudly, not hand over wrong audio).
// No reference decoder exists in this image (no libFLAC / ffmpeg): tests/test_audio_ingest.py exercises every code path
// against an independent bit-level FLAC *writer* (tests/flac_writer.py) -- "parity unpinned" against real encoders.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/crisperwhisper.h"

namespace {

thread_local char g_flac_err[256] = "";
int flac_fail(const char* msg) {
    snprintf(g_flac_err, sizeof(g_flac_err), "%s", msg);
    return -22;
}

struct BitReader {
    const uint8_t* p; size_t n; size_t pos = 0;      // byte position
    uint64_t acc = 0; int nbits = 0;                 // bit accumulator (MSB first)
    bool bad = false;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    inline void fill(int need) {
        while (nbits < need) {
            if (pos >= n) { bad = true; acc <<= 8; nbits += 8; continue; }
            acc = (acc << 8) | p[pos++]; nbits += 8;
        }
    }
    inline uint32_t u(int bits) {                    // 0..32 bits
        if (bits == 0) return 0;
        fill(bits);
        const uint32_t v = (uint32_t)((acc >> (nbits - bits)) & ((bits == 32) ? 0xffffffffull : ((1ull << bits) - 1)));
        nbits -= bits;
        return v;
    }
    inline int64_t s(int bits) {                     // signed, 1..33 bits
        if (bits <= 32) {
            const uint32_t v = u(bits);
            const int sh = 64 - bits;
            return ((int64_t)((uint64_t)v << sh)) >> sh;
        }
        const uint64_t hi = u(bits - 32), lo = u(32);
        const uint64_t v = (hi << 32) | lo;
        const int sh = 64 - bits;
        return ((int64_t)(v << sh)) >> sh;
    }
    inline uint32_t unary() {                        // number of 0 bits before the next 1
        uint32_t c = 0;
        for (;;) {
            fill(1);
            if (bad) return c;
            // scan the bits we have
            while (nbits > 0) {
                if ((acc >> (nbits - 1)) & 1) { --nbits; return c; }
                --nbits; ++c;
            }
        }
    }
    inline void align() { nbits -= nbits % 8; }
    inline size_t byte_pos() const { return pos - (size_t)(nbits / 8); }
};

uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
    }
    return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint16_t)d[i] << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x8005) : (uint16_t)(c << 1);
    }
    return c;
}

// ---- MD5 (RFC 1321), for the STREAMINFO signature ----------------------------------------------------------------
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0; uint8_t buf[64]; int fill = 0;
    static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
            0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
            0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
            0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
            0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
            0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B;
            B = B + rol(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t k = (64 - (size_t)fill) < n ? (64 - (size_t)fill) : n;
            memcpy(buf + fill, p, k); fill += (int)k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t pad = 0x80; update(&pad, 1);
        const uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (8 * i));
        update(lb, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
    }
};

struct StreamInfo {
    int min_block = 0, max_block = 0, sample_rate = 0, channels = 0, bps = 0;
    int64_t total = 0;
    uint8_t md5[16] = {0};
    size_t audio_off = 0;
};

int parse_metadata(const uint8_t* d, size_t n, StreamInfo& si) {
    if (n < 4 + 4 + 34 || memcmp(d, "fLaC", 4) != 0) return flac_fail("not a FLAC stream (missing fLaC marker)");
    size_t pos = 4;
    bool have = false;
    for (;;) {
        if (pos + 4 > n) return flac_fail("truncated FLAC metadata");
        const bool last = d[pos] & 0x80;
        const int type = d[pos] & 0x7f;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > n) return flac_fail("truncated FLAC metadata block");
        if (type == 0) {
            if (len < 34) return flac_fail("short STREAMINFO block");
            const uint8_t* s = d + pos;
            si.min_block = (s[0] << 8) | s[1];
            si.max_block = (s[2] << 8) | s[3];
            si.sample_rate = ((int)s[10] << 12) | ((int)s[11] << 4) | (s[12] >> 4);
            si.channels = ((s[12] >> 1) & 7) + 1;
            si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            memcpy(si.md5, s + 18, 16);
            have = true;
        } else if (type == 127) {
            return flac_fail("invalid FLAC metadata block type 127");
        }
        pos += len;
        if (last) break;
    }
    if (!have) return flac_fail("FLAC stream without STREAMINFO");
    if (si.sample_rate <= 0 || si.bps < 4 || si.bps > 32) return flac_fail("unsupported FLAC STREAMINFO (sample rate / bits per sample)");
    si.audio_off = pos;
    return 0;
}

// residual of one subframe: `n` = blocksize, `order` = predictor order; residuals written to res[order..n-1]
int read_residual(BitReader& br, int n, int order, std::vector<int64_t>& res) {
    const int method = (int)br.u(2);
    if (method > 1) return flac_fail("reserved FLAC residual coding method");
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.u(4);
    const int parts = 1 << porder;
    if ((n >> porder) << porder != n && porder > 0) return flac_fail("FLAC partition order does not divide the block size");
    int idx = order;
    for (int pI = 0; pI < parts; ++pI) {
        int cnt = (n >> porder) - (pI == 0 ? order : 0);
        if (porder == 0) cnt = n - order;
        if (cnt < 0) return flac_fail("FLAC partition smaller than the predictor order");
        const int param = (int)br.u(pbits);
        if (param == esc) {
            const int raw = (int)br.u(5);
            for (int k = 0; k < cnt; ++k) res[idx++] = raw ? br.s(raw) : 0;
        } else {
            for (int k = 0; k < cnt; ++k) {
                const uint64_t q = br.unary();
                const uint64_t v = (q << param) | (param ? br.u(param) : 0);
                if (v >> 33) return flac_fail("FLAC residual outside 32 bits (corrupt stream)");   // RFC 9639 9.2.7.3: |residual| < 2^31
                res[idx++] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
            }
        }
        if (br.bad) return flac_fail("truncated FLAC residual");
    }
    return 0;
}

int read_subframe(BitReader& br, int n, int bps, std::vector<int64_t>& out) {
    if (br.u(1)) return flac_fail("FLAC subframe padding bit set");
    const int type = (int)br.u(6);
    int wasted = 0;
    if (br.u(1)) wasted = (int)br.unary() + 1;
    if (wasted >= bps) return flac_fail("FLAC wasted bits exceed the sample size");
    bps -= wasted;
    const int64_t lim = (int64_t)1 << bps;                   // one bit of slack over the legal range; bounds the predictor sums below
    out.assign(n, 0);
    if (type == 0) {                                         // CONSTANT
        const int64_t v = br.s(bps);
        for (int i = 0; i < n; ++i) out[i] = v;
    } else if (type == 1) {                                  // VERBATIM
        for (int i = 0; i < n; ++i) out[i] = br.s(bps);
    } else if (type >= 8 && type <= 12) {                    // FIXED, order type - 8
        const int order = type - 8;
        if (order > n) return flac_fail("FLAC fixed predictor order exceeds the block size");
        for (int i = 0; i < order; ++i) out[i] = br.s(bps);
        int r = read_residual(br, n, order, out);
        if (r) return r;
        for (int i = order; i < n; ++i) {
            int64_t p = 0;
            switch (order) {
                case 1: p = out[i - 1]; break;
                case 2: p = 2 * out[i - 1] - out[i - 2]; break;
                case 3: p = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
                case 4: p = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
                default: break;
            }
            out[i] += p;
            if (out[i] >= lim || out[i] < -lim) return flac_fail("FLAC sample outside its bit depth (corrupt stream)");
        }
    } else if (type >= 32) {                                 // LPC, order (type & 31) + 1
        const int order = (type & 31) + 1;
        if (order > n) return flac_fail("FLAC LPC order exceeds the block size");
        for (int i = 0; i < order; ++i) out[i] = br.s(bps);
        const int prec = (int)br.u(4) + 1;
        if (prec == 16) return flac_fail("invalid FLAC LPC precision");
        const int shift = (int)br.s(5);
        if (shift < 0) return flac_fail("negative FLAC LPC shift");
        int64_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.s(prec);
        int r = read_residual(br, n, order, out);
        if (r) return r;
        for (int i = order; i < n; ++i) {
            int64_t acc = 0;
            for (int k = 0; k < order; ++k) acc += coef[k] * out[i - 1 - k];   // |coef| < 2^15, |out| < 2^33, order <= 32: < 2^53
            out[i] += acc >> shift;
            if (out[i] >= lim || out[i] < -lim) return flac_fail("FLAC sample outside its bit depth (corrupt stream)");
        }
    } else {
        return flac_fail("reserved FLAC subframe type");
    }
    if (br.bad) return flac_fail("truncated FLAC subframe");
    if (wasted) for (int i = 0; i < n; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
    return 0;
}

// one frame starting at d[off]; appends blocksize x channels samples; returns bytes consumed or < 0
int64_t read_frame(const uint8_t* d, size_t n, size_t off, const StreamInfo& si, std::vector<std::vector<int64_t>>& ch_out,
                   int& blocksize_out, int& bps_out) {
    BitReader br(d + off, n - off);
    if (br.u(14) != 0x3ffe) return flac_fail("lost FLAC frame sync");
    if (br.u(1)) return flac_fail("reserved bit set in FLAC frame header");
    br.u(1);                                                 // blocking strategy: only affects the meaning of the coded number
    const int bs_code = (int)br.u(4), sr_code = (int)br.u(4), ch_code = (int)br.u(4), ss_code = (int)br.u(3);
    if (br.u(1)) return flac_fail("reserved bit set in FLAC frame header");
    {   // UTF-8 style coded frame / sample number (1..7 bytes)
        const uint32_t b0 = br.u(8);
        int extra = 0;
        if (b0 & 0x80) {
            if ((b0 & 0xe0) == 0xc0) extra = 1; else if ((b0 & 0xf0) == 0xe0) extra = 2; else if ((b0 & 0xf8) == 0xf0) extra = 3;
            else if ((b0 & 0xfc) == 0xf8) extra = 4; else if ((b0 & 0xfe) == 0xfc) extra = 5; else if (b0 == 0xfe) extra = 6;
            else return flac_fail("invalid coded number in FLAC frame header");
        }
        for (int i = 0; i < extra; ++i) if ((br.u(8) & 0xc0) != 0x80) return flac_fail("invalid coded number in FLAC frame header");
    }
    int bs;
    if (bs_code == 0) return flac_fail("reserved FLAC block size code");
    else if (bs_code == 1) bs = 192;
    else if (bs_code <= 5) bs = 576 << (bs_code - 2);
    else if (bs_code == 6) bs = (int)br.u(8) + 1;
    else if (bs_code == 7) bs = (int)br.u(16) + 1;
    else bs = 256 << (bs_code - 8);
    if (sr_code == 12) br.u(8); else if (sr_code == 13 || sr_code == 14) br.u(16); else if (sr_code == 15) return flac_fail("invalid FLAC sample rate code");
    static const int ss_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_tab[ss_code];
    if (bps < 0) return flac_fail("reserved FLAC sample size code");
    if (bps == 0) bps = si.bps;
    const size_t hdr_len = br.byte_pos();
    const uint32_t c8 = br.u(8);
    if (br.bad) return flac_fail("truncated FLAC frame header");
    if (crc8(d + off, hdr_len) != c8) return flac_fail("FLAC frame header CRC-8 mismatch");
    int nch;
    if (ch_code <= 7) nch = ch_code + 1; else if (ch_code <= 10) nch = 2; else return flac_fail("reserved FLAC channel assignment");
    if (nch != si.channels) return flac_fail("FLAC frame channel count differs from STREAMINFO");
    std::vector<std::vector<int64_t>> sub(nch);
    for (int c = 0; c < nch; ++c) {
        int b = bps;
        if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) b += 1;   // the side channel
        if (b > 33) return flac_fail("FLAC side channel wider than 33 bits");
        int r = read_subframe(br, bs, b, sub[c]);
        if (r) return r;
    }
    br.align();
    const size_t body_len = br.byte_pos();
    const uint32_t c16 = br.u(16);
    if (br.bad) return flac_fail("truncated FLAC frame");
    if (crc16(d + off, body_len) != c16) return flac_fail("FLAC frame CRC-16 mismatch");
    if (ch_code == 8) { for (int i = 0; i < bs; ++i) sub[1][i] = sub[0][i] - sub[1][i]; }                 // left, side
    else if (ch_code == 9) { for (int i = 0; i < bs; ++i) sub[0][i] = sub[1][i] + sub[0][i]; }            // side, right
    else if (ch_code == 10) {                                                                             // mid, side
        for (int i = 0; i < bs; ++i) {
            const int64_t side = sub[1][i];
            const int64_t mid = (int64_t)(((uint64_t)sub[0][i] << 1) | (uint64_t)(side & 1));
            sub[0][i] = (mid + side) >> 1;
            sub[1][i] = (mid - side) >> 1;
        }
    }
    for (int c = 0; c < nch; ++c) ch_out[c].insert(ch_out[c].end(), sub[c].begin(), sub[c].end());
    blocksize_out = bs; bps_out = bps;
    return (int64_t)(body_len + 2);
}

// max_frames > 0: give up INSIDE the frame loop once more than that many sample frames are decoded -- a few hundred bytes of
// CONSTANT subframes expand without bound, and the caller's size check would only run after everything was materialised
int decode_all(const uint8_t* d, size_t n, StreamInfo& si, std::vector<std::vector<int64_t>>& ch, int64_t max_frames) {
    int r = parse_metadata(d, n, si);
    if (r) return r;
    if (max_frames > 0 && si.total > max_frames) return flac_fail("FLAC stream declares more sample frames than the caller accepts");
    ch.assign(si.channels, {});
    size_t off = si.audio_off;
    Md5 md5;
    const int bytes_ps = (si.bps + 7) / 8;
    std::vector<uint8_t> raw;
    while (off + 2 <= n) {
        if (si.total > 0 && (int64_t)ch[0].size() >= si.total) break;
        const size_t before = ch[0].size();
        int bs = 0, bps = 0;
        const int64_t used = read_frame(d, n, off, si, ch, bs, bps);
        if (used < 0) return (int)used;
        if (bps != si.bps) return flac_fail("FLAC frame sample size differs from STREAMINFO");
        off += (size_t)used;
        if (max_frames > 0 && (int64_t)ch[0].size() > max_frames) return flac_fail("FLAC stream decodes to more sample frames than the caller accepts");
        raw.resize((size_t)bs * si.channels * bytes_ps);     // MD5 is over interleaved little-endian samples
        size_t w = 0;
        for (int i = 0; i < bs; ++i)
            for (int c = 0; c < si.channels; ++c) {
                const int64_t v = ch[c][before + i];
                for (int bI = 0; bI < bytes_ps; ++bI) raw[w++] = (uint8_t)((uint64_t)v >> (8 * bI));
            }
        md5.update(raw.data(), raw.size());
    }
    if (ch[0].empty()) return flac_fail("FLAC stream without audio frames");
    if (si.total > 0 && (int64_t)ch[0].size() != si.total) return flac_fail("FLAC stream ends before the sample count of STREAMINFO");
    bool has_md5 = false;
    for (int i = 0; i < 16; ++i) has_md5 = has_md5 || si.md5[i] != 0;
    if (has_md5) {
        uint8_t got[16];
        md5.final(got);
        if (memcmp(got, si.md5, 16) != 0) return flac_fail("decoded audio does not match the MD5 signature in STREAMINFO");
    }
    return 0;
}

}  // namespace

extern "C" {

const char* cw_flac_last_error(void) { return g_flac_err; }

int32_t cw_flac_info(const uint8_t* data, int64_t n, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                     int64_t* total_frames) {
    if (!data || n < 0) return flac_fail("null FLAC buffer");
    StreamInfo si;
    const int r = parse_metadata(data, (size_t)n, si);
    if (r) return r;
    if (sample_rate) *sample_rate = si.sample_rate;
    if (channels) *channels = si.channels;
    if (bits_per_sample) *bits_per_sample = si.bps;
    if (total_frames) *total_frames = si.total;               // 0: unknown (streamed encoder)
    return 0;
}

int32_t cw_flac_decode(const uint8_t* data, int64_t n, int32_t* pcm_s32, int64_t cap_frames, int64_t* n_frames) {
    if (!data || n < 0 || !n_frames) return flac_fail("null FLAC buffer");
    StreamInfo si;
    std::vector<std::vector<int64_t>> ch;
    // cap_frames bounds the decoding itself (size query: the most the caller would accept; 0 = no bound)
    const int r = decode_all(data, (size_t)n, si, ch, cap_frames > 0 ? cap_frames : 0);
    if (r) return r;
    const int64_t frames = (int64_t)ch[0].size();
    *n_frames = frames;
    if (!pcm_s32) return 0;                                   // size query
    if (cap_frames < frames) return flac_fail("output buffer too small for the decoded FLAC stream");
    const int up = 32 - si.bps;                               // left-justify: cw_ingest's S32 path scales by 2^-31
    for (int64_t i = 0; i < frames; ++i)
        for (int c = 0; c < si.channels; ++c) pcm_s32[i * si.channels + c] = (int32_t)((uint32_t)(uint64_t)ch[c][i] << up);
    return 0;
}

}  // extern "C"
