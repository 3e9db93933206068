// piz.cpp -- decoder for OpenEXR's PIZ compression (value LUT + 2-D Haar-like wavelet + Huffman),
// the default of many renderers' .exr output and one of the schemes tinyexr's LoadEXR (the reference's
// loader, src/main.cpp:155) accepts.
//
// Written from the published description of the format (OpenEXR "PIZ" scheme: ImfPizCompressor /
// ImfHuf / ImfWav).  STATUS: no third-party PIZ file exists in this environment, so the decoder is
// verified only against an independent encoder written for the tests (tests/test_codecs.py) from the
// same description -- bit-stream layout, canonical code assignment, 14/16-bit wavelet modes, odd
// sizes, run-length symbol.  Treat real-file compatibility as unconfirmed until checked outside.
#include "image_io.hpp"

#include <cstring>

namespace mid {
namespace codec {

namespace {

constexpr int kSymbolBits = 16, kFastBits = 14;
constexpr int kSymbolCount = (1 << kSymbolBits) + 1, kFastEntries = 1 << kFastBits, kFastMask = kFastEntries - 1;
constexpr int kShortZeroRunCode = 59, kLongZeroRunCode = 63, kLongZeroRunBase = 2 + kLongZeroRunCode - kShortZeroRunCode;
constexpr int kBitmapBytes = 8192;

struct FastEntry {                  // one slot of the kFastBits-bit prefix table
    int length = 0;                 // code no longer than the prefix: its length (0 = none)
    int symbol = 0;                 // that code's symbol; for longer codes: how many share this prefix
    std::vector<int> candidates;    // longer codes sharing this prefix: their symbols
};

struct BitReader {
    const uint8_t *in, *end;
    uint64_t c = 0;
    int lc = 0;
    bool ok = true;
    uint64_t get(int n)
    {
        while (lc < n) {
            if (in >= end) { ok = false; return 0; }
            c = (c << 8) | *in++;
            lc += 8;
        }
        lc -= n;
        return (c >> lc) & ((1ull << n) - 1);
    }
};

inline int code_length(uint64_t code) { return (int)(code & 63); }
inline uint64_t code_bits(uint64_t code) { return code >> 6; }

// Code lengths -> canonical codes (packed as length | code << 6).
void assign_canonical_codes(std::vector<uint64_t> &codes)
{
    uint64_t n[59] = {0};
    for (int i = 0; i < kSymbolCount; ++i) n[codes[i]] += 1;
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) {
        const uint64_t nc = (c + n[i]) >> 1;
        n[i] = c;
        c = nc;
    }
    for (int i = 0; i < kSymbolCount; ++i) {
        const int l = (int)codes[i];
        if (l > 0) codes[i] = (uint64_t)l | (n[l]++ << 6);
    }
}

bool read_length_table(const uint8_t *&p, const uint8_t *end, int im, int iM, std::vector<uint64_t> &codes, std::string &err)
{
    std::fill(codes.begin(), codes.end(), 0);
    BitReader br{p, end};
    for (; im <= iM; ++im) {
        const uint64_t l = br.get(6);
        if (!br.ok) { err = "exr/piz: truncated Huffman table"; return false; }
        codes[im] = l;
        int zero_run = 0;
        if (l == (uint64_t)kLongZeroRunCode) zero_run = (int)br.get(8) + kLongZeroRunBase;
        else if (l >= (uint64_t)kShortZeroRunCode) zero_run = (int)l - kShortZeroRunCode + 2;
        if (!br.ok) { err = "exr/piz: truncated Huffman table"; return false; }
        if (zero_run) {
            if (im + zero_run > iM + 1) { err = "exr/piz: Huffman table overruns"; return false; }
            while (zero_run--) codes[im++] = 0;
            --im;
        }
    }
    p = br.in;
    assign_canonical_codes(codes);
    return true;
}

bool build_fast_table(const std::vector<uint64_t> &codes, int im, int iM, std::vector<FastEntry> &dec, std::string &err)
{
    for (; im <= iM; ++im) {
        const uint64_t c = code_bits(codes[im]);
        const int l = code_length(codes[im]);
        if (l == 0) continue;
        if (c >> l) { err = "exr/piz: invalid Huffman code"; return false; }
        if (l > kFastBits) {
            FastEntry &pl = dec[(size_t)(c >> (l - kFastBits))];
            if (pl.length) { err = "exr/piz: invalid Huffman table"; return false; }
            pl.symbol++;
            pl.candidates.push_back(im);
        } else {
            const size_t base = (size_t)(c << (kFastBits - l));
            for (size_t i = 0; i < ((size_t)1 << (kFastBits - l)); ++i) {
                FastEntry &pl = dec[base + i];
                if (pl.length || !pl.candidates.empty()) { err = "exr/piz: invalid Huffman table"; return false; }
                pl.length = l;
                pl.symbol = im;
            }
        }
    }
    return true;
}

bool decode_symbols(const std::vector<uint64_t> &codes, const std::vector<FastEntry> &dec, const uint8_t *in, const uint8_t *file_end,
                int ni, int rlc, size_t no, uint16_t *out, std::string &err)
{
    uint64_t c = 0;
    int lc = 0;
    uint16_t *outb = out, *oe = out + no;
    const uint8_t *ie = in + (ni + 7) / 8;
    if (ie > file_end) { err = "exr/piz: truncated Huffman data"; return false; }
    auto get_code = [&](int po) -> bool {
        if (po == rlc) {
            if (lc < 8) { if (in >= ie) return false; c = (c << 8) | *in++; lc += 8; }
            lc -= 8;
            int cs = (int)((c >> lc) & 0xff);
            if (out + cs > oe || out == outb) return false;
            const uint16_t s = out[-1];
            while (cs-- > 0) *out++ = s;
        } else {
            if (out >= oe) return false;
            *out++ = (uint16_t)po;
        }
        return true;
    };
    while (in < ie) {
        c = (c << 8) | *in++;
        lc += 8;
        while (lc >= kFastBits) {
            const FastEntry &pl = dec[(size_t)((c >> (lc - kFastBits)) & kFastMask)];
            if (pl.length) {
                lc -= pl.length;
                if (!get_code(pl.symbol)) { err = "exr/piz: corrupt Huffman data"; return false; }
            } else {
                if (pl.candidates.empty()) { err = "exr/piz: corrupt Huffman data (no such code)"; return false; }
                bool found = false;
                for (int sym : pl.candidates) {
                    const int l = code_length(codes[sym]);
                    while (lc < l && in < ie) { c = (c << 8) | *in++; lc += 8; }
                    if (lc >= l && code_bits(codes[sym]) == ((c >> (lc - l)) & ((1ull << l) - 1))) {
                        lc -= l;
                        if (!get_code(sym)) { err = "exr/piz: corrupt Huffman data"; return false; }
                        found = true;
                        break;
                    }
                }
                if (!found) { err = "exr/piz: corrupt Huffman data (long code)"; return false; }
            }
        }
    }
    const int i = (8 - ni) & 7;     // padding bits of the last byte
    c >>= i;
    lc -= i;
    while (lc > 0) {
        const FastEntry &pl = dec[(size_t)((c << (kFastBits - lc)) & kFastMask)];
        if (!pl.length) { err = "exr/piz: corrupt Huffman data (tail)"; return false; }
        lc -= pl.length;
        if (lc < 0) { err = "exr/piz: corrupt Huffman data (tail)"; return false; }
        if (!get_code(pl.symbol)) { err = "exr/piz: corrupt Huffman data"; return false; }
    }
    if ((size_t)(out - outb) != no) { err = "exr/piz: Huffman data decodes to the wrong size"; return false; }
    return true;
}

bool huf_uncompress(const uint8_t *comp, size_t ncomp, uint16_t *raw, size_t nraw, std::string &err)
{
    if (ncomp == 0) { if (nraw) { err = "exr/piz: empty Huffman block"; return false; } return true; }
    if (ncomp < 20) { err = "exr/piz: truncated Huffman header"; return false; }
    uint32_t hdr[5];
    memcpy(hdr, comp, 20);
    const int im = (int)hdr[0], iM = (int)hdr[1], nbits = (int)hdr[3];
    if (im < 0 || im >= kSymbolCount || iM < 0 || iM >= kSymbolCount || im > iM || nbits < 0) { err = "exr/piz: bad Huffman header"; return false; }
    const uint8_t *p = comp + 20, *end = comp + ncomp;
    std::vector<uint64_t> codes(kSymbolCount);
    if (!read_length_table(p, end, im, iM, codes, err)) return false;
    if ((size_t)nbits > 8 * (size_t)(end - p)) { err = "exr/piz: Huffman bit count beyond the block"; return false; }
    std::vector<FastEntry> dec(kFastEntries);
    if (!build_fast_table(codes, im, iM, dec, err)) return false;
    return decode_symbols(codes, dec, p, end, nbits, iM, nraw, raw, err);
}

// ---- wavelet ---------------------------------------------------------------------------------
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}

inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b)
{
    constexpr int A_OFFSET = 1 << 15, MOD_MASK = (1 << 16) - 1;
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & MOD_MASK;
    const int aa = (d + bb - A_OFFSET) & MOD_MASK;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}

void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx)
{
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1;
    p2 = p;
    p >>= 1;
    while (p >= 1) {
        uint16_t *py = in;
        uint16_t *ey = in + (long)oy * (ny - p2);
        const long oy1 = (long)oy * p, oy2 = (long)oy * p2, ox1 = (long)ox * p, ox2 = (long)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py;
            uint16_t *ex = py + (long)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) {
                    wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11);
                    wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11);
                } else {
                    wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11);
                    wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11);
                }
            }
            if (nx & p) {
                uint16_t *p10 = px + oy1;
                if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t *px = py;
            uint16_t *ex = py + (long)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

// One PIZ chunk -> `nl` scanlines in the scanline-interleaved layout the other schemes produce
// (per line: channel after channel).  chan_size[c] = 16-bit words per pixel (1 HALF, 2 FLOAT/UINT).
bool piz_decode_block(const uint8_t *comp, size_t ncomp, int width, int nl, const std::vector<int> &chan_size,
                      std::vector<uint8_t> &out, std::string &err)
{
    size_t words = 0;
    for (int s : chan_size) words += (size_t)width * nl * s;
    std::vector<uint16_t> tmp(words);
    if (ncomp < 4) { err = "exr/piz: truncated chunk"; return false; }
    uint16_t min_nz, max_nz;
    memcpy(&min_nz, comp, 2);
    memcpy(&max_nz, comp + 2, 2);
    const uint8_t *p = comp + 4, *end = comp + ncomp;
    std::vector<uint8_t> bitmap(kBitmapBytes, 0);
    if (max_nz >= kBitmapBytes) { err = "exr/piz: bad bitmap range"; return false; }
    if (min_nz <= max_nz) {
        const size_t n = (size_t)max_nz - min_nz + 1;
        if ((size_t)(end - p) < n) { err = "exr/piz: truncated bitmap"; return false; }
        memcpy(bitmap.data() + min_nz, p, n);
        p += n;
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if ((size_t)(end - p) < 4) { err = "exr/piz: truncated chunk"; return false; }
    int32_t length;
    memcpy(&length, p, 4);
    p += 4;
    if (length < 0 || (size_t)length > (size_t)(end - p)) { err = "exr/piz: bad Huffman length"; return false; }
    if (!huf_uncompress(p, (size_t)length, tmp.data(), words, err)) return false;
    // wavelet, channel by channel (each channel's nl rows are contiguous in tmp)
    size_t off = 0;
    std::vector<size_t> start(chan_size.size());
    for (size_t c = 0; c < chan_size.size(); ++c) {
        start[c] = off;
        for (int j = 0; j < chan_size[c]; ++j)
            wav2_decode(tmp.data() + off + j, width, chan_size[c], nl, width * chan_size[c], max_value);
        off += (size_t)width * nl * chan_size[c];
    }
    for (auto &v : tmp) v = lut[v];
    // back to scanline order
    size_t line_words = 0;
    for (int s : chan_size) line_words += (size_t)width * s;
    out.resize(line_words * nl * 2);
    uint16_t *o = (uint16_t *)out.data();
    std::vector<size_t> cur = start;
    for (int y = 0; y < nl; ++y)
        for (size_t c = 0; c < chan_size.size(); ++c) {
            const size_t n = (size_t)width * chan_size[c];
            memcpy(o, tmp.data() + cur[c], n * 2);
            o += n;
            cur[c] += n;
        }
    return true;
}

}  // namespace codec
}  // namespace mid
