// exr.cpp -- minimal scanline OpenEXR codec over zlib (the reference calls tinyexr's LoadEXR /
// SaveEXR, src/main.cpp:155,1699,1744,1887; tinyexr is an un-vendored submodule, absent here).
//
// Read: single-part scanline files and single-level (ONE_LEVEL) tiled files, compression NONE / RLE / ZIPS / ZIP /
// PIZ (piz.cpp) / PXR24, channel types UINT / HALF / FLOAT, any data window, increasing or decreasing line order
// -- the set tinyexr's LoadEXR accepts.  Channels R,G,B,A are looked up by name
// (a layer prefix "xxx.R" is accepted when no plain names exist); a missing A reads as 1.0 and a
// single-channel file is replicated into RGB -- the behaviour of tinyexr's LoadEXR that the reference
// relies on (README.md:59 "alpha is kept").  Mip/rip-mapped tiles, multi-part, deep and B44/DWA files are
// rejected with a message naming the feature.
// Write: channels A,B,G,R as FLOAT, ZIP blocks of 16 lines (NONE when the image is smaller than
// 16x16), the attribute set tinyexr's SaveEXR(data, w, h, 4, 0, ...) emits.
#include "image_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace mid {
namespace codec {

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    if (got != out.size()) { err = "short read on " + path; return false; }
    return true;
}

bool write_file(const std::string &path, const std::vector<uint8_t> &data, std::string &err)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    const size_t put = fwrite(data.data(), 1, data.size(), f);
    if (fclose(f) != 0 || put != data.size()) { err = "short write on " + path; return false; }
    return true;
}

static float half_to_float(uint16_t h)
{
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else {   // subnormal half -> normal float
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024)) { mm <<= 1; ++sh; }
            u = s | (uint32_t)(127 - 15 - sh + 1) << 23 | (mm & 1023) << 13;
        }
    } else if (e == 31) u = s | 0x7f800000u | m << 13;
    else u = s | (e + 112) << 23 | m << 13;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) ok = false; return ok; }
    uint8_t u8() { return need(1) ? *p++ : 0; }
    int32_t i32() { if (!need(4)) return 0; int32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    std::string str() { std::string s; while (need(1) && *p) s.push_back((char)*p++); if (ok) ++p; return s; }
};

struct Channel { std::string name; int type; int xs, ys; };

// The ZIP/RLE post-processing of OpenEXR: undo the byte-delta predictor, then re-interleave the
// two half-streams (even bytes first, odd bytes second).
static void unpredict_and_interleave(std::vector<uint8_t> &t, std::vector<uint8_t> &out)
{
    const size_t n = t.size();
    for (size_t i = 1; i < n; ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    out.resize(n);
    const size_t half = (n + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < n;) {
        out[i++] = t[a++];
        if (i < n) out[i++] = t[b++];
    }
}

static bool rle_decode(const uint8_t *in, size_t n, std::vector<uint8_t> &out, size_t expect)
{
    out.clear();
    out.reserve(expect);
    size_t i = 0;
    while (i < n) {
        const int8_t c = (int8_t)in[i++];
        if (c < 0) {
            const size_t cnt = (size_t)(-(int)c);
            if (i + cnt > n || out.size() + cnt > expect) return false;
            out.insert(out.end(), in + i, in + i + cnt);
            i += cnt;
        } else {
            const size_t cnt = (size_t)c + 1;
            if (i >= n || out.size() + cnt > expect) return false;
            out.insert(out.end(), cnt, in[i++]);
        }
    }
    return out.size() == expect;
}

// PXR24 (ImfPxr24Compressor): per scanline and channel the values are delta-coded and split into byte planes (most
// significant first) -- 2 planes for HALF, 3 for FLOAT (the float's low 8 mantissa bits are dropped: lossy), 4 for
// UINT -- and the whole block is deflated.  Output: the standard scanline layout.
static bool pxr24_decode(const uint8_t *in, size_t n_in, long bw, long nl, const std::vector<int> &types,
                         std::vector<uint8_t> &out, std::string &err)
{
    size_t planes_per_line = 0, line_bytes = 0;
    for (int t : types) { planes_per_line += t == 1 ? 2 : t == 2 ? 3 : 4; line_bytes += (size_t)bw * (t == 1 ? 2 : 4); }
    const size_t packed = planes_per_line * (size_t)bw * (size_t)nl;
    std::vector<uint8_t> tmp(packed);
    uLongf got = (uLongf)packed;
    if (uncompress(tmp.data(), &got, in, (uLong)n_in) != Z_OK || got != packed) { err = "exr: corrupt PXR24 chunk"; return false; }
    out.resize(line_bytes * (size_t)nl);
    const uint8_t *p = tmp.data();
    uint8_t *q = out.data();
    for (long l = 0; l < nl; ++l)
        for (int t : types) {
            const int np = t == 1 ? 2 : t == 2 ? 3 : 4;
            const uint8_t *pl[4] = {p, p + bw, p + 2 * bw, p + 3 * bw};
            p += (size_t)np * bw;
            uint32_t pixel = 0;
            for (long x = 0; x < bw; ++x) {
                if (t == 1) {
                    pixel += ((uint32_t)pl[0][x] << 8) | pl[1][x];
                    const uint16_t hv = (uint16_t)pixel;
                    memcpy(q, &hv, 2); q += 2;
                } else {
                    if (t == 2) pixel += ((uint32_t)pl[0][x] << 24) | ((uint32_t)pl[1][x] << 16) | ((uint32_t)pl[2][x] << 8);
                    else pixel += ((uint32_t)pl[0][x] << 24) | ((uint32_t)pl[1][x] << 16) | ((uint32_t)pl[2][x] << 8) | pl[3][x];
                    memcpy(q, &pixel, 4); q += 4;
                }
            }
        }
    return true;
}

bool exr_decode(const std::vector<uint8_t> &file, int &w, int &h, std::vector<float> &rgba, std::string &err)
{
    return exr_decode_to(file, w, h, [&](size_t n) { rgba.assign(n, 0.f); return rgba.data(); }, err);
}

bool exr_decode_to(const std::vector<uint8_t> &file, int &w, int &h, const FloatAlloc &alloc, std::string &err)
{
    Reader r{file.data(), file.data() + file.size()};
    if (file.size() < 8 || r.i32() != 20000630) { err = "exr: not an OpenEXR file"; return false; }
    const int32_t ver = r.i32();
    if ((ver & 0xff) != 2) { err = "exr: unsupported version"; return false; }
    const bool tiled = (ver & 0x200) != 0;
    if (ver & 0x800) { err = "exr: deep data is not supported"; return false; }
    if (ver & 0x1000) { err = "exr: multi-part files are not supported"; return false; }

    std::vector<Channel> chans;
    int compression = -1, line_order = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    bool have_dw = false;
    uint32_t tile_w = 0, tile_h = 0;
    int tile_mode = -1;
    for (;;) {
        const std::string name = r.str();
        if (!r.ok) { err = "exr: truncated header"; return false; }
        if (name.empty()) break;
        const std::string type = r.str();
        const int32_t size = r.i32();
        if (!r.ok || size < 0 || !r.need((size_t)size)) { err = "exr: truncated attribute " + name; return false; }
        Reader a{r.p, r.p + size};
        r.p += size;
        if (name == "channels") {
            for (;;) {
                Channel c;
                c.name = a.str();
                if (!a.ok) { err = "exr: bad channel list"; return false; }
                if (c.name.empty()) break;
                c.type = a.i32();
                a.u8(); a.u8(); a.u8(); a.u8();
                c.xs = a.i32(); c.ys = a.i32();
                if (!a.ok) { err = "exr: bad channel list"; return false; }
                chans.push_back(c);
            }
        } else if (name == "compression") compression = a.u8();
        else if (name == "dataWindow") { for (int i = 0; i < 4; ++i) dw[i] = a.i32(); have_dw = a.ok; }
        else if (name == "lineOrder") line_order = a.u8();
        else if (name == "tiles") { tile_w = (uint32_t)a.i32(); tile_h = (uint32_t)a.i32(); tile_mode = a.u8(); if (!a.ok) tile_mode = -1; }
    }
    (void)line_order;   // chunks carry their own y; the offset table is indexed by block either way
    if (chans.empty() || !have_dw || compression < 0) { err = "exr: missing channels/dataWindow/compression"; return false; }
    static const char *cname[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
    if (compression > 5) { err = std::string("exr: compression ") + (compression < 10 ? cname[compression] : "?") + " is not supported (NONE/RLE/ZIPS/ZIP/PIZ/PXR24 only)"; return false; }
    if (tiled) {
        if (tile_mode < 0 || tile_w == 0 || tile_h == 0 || tile_w > 65536 || tile_h > 65536) { err = "exr: tiled file without a valid tiles attribute"; return false; }
        if ((tile_mode & 0xf) != 0) { err = "exr: mip/rip-mapped tiles are not supported (ONE_LEVEL only)"; return false; }
    }
    const long W = (long)dw[2] - dw[0] + 1, H = (long)dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || W > 65536 || H > 65536) { err = "exr: bad data window"; return false; }
    if ((double)W * H * 2 > (double)file.size() * 1100.0) { err = "exr: data window larger than the file can hold"; return false; }
    size_t px_bytes = 0;                       // bytes of one pixel over all channels; a line of a block bw wide holds bw * px_bytes
    std::vector<size_t> chpre(chans.size());   // bytes per pixel of the channels stored before channel c
    std::vector<int> types(chans.size());
    for (size_t c = 0; c < chans.size(); ++c) {
        if (chans[c].xs != 1 || chans[c].ys != 1) { err = "exr: subsampled channels are not supported"; return false; }
        if (chans[c].type < 0 || chans[c].type > 2) { err = "exr: unknown pixel type"; return false; }
        chpre[c] = px_bytes;
        types[c] = chans[c].type;
        px_bytes += chans[c].type == 1 ? 2 : 4;
    }
    // channel -> RGBA slot
    int slot[4] = {-1, -1, -1, -1};
    static const char *want[4] = {"R", "G", "B", "A"};
    for (int k = 0; k < 4; ++k)
        for (size_t c = 0; c < chans.size(); ++c)
            if (chans[c].name == want[k]) slot[k] = (int)c;
    if (slot[0] < 0 && slot[1] < 0 && slot[2] < 0)   // accept "layer.R" style names of the first layer that has them
        for (int k = 0; k < 4; ++k)
            for (size_t c = 0; c < chans.size() && slot[k] < 0; ++c) {
                const std::string &n = chans[c].name;
                if (n.size() > 2 && n[n.size() - 2] == '.' && n.compare(n.size() - 1, 1, want[k]) == 0) slot[k] = (int)c;
            }
    const bool gray = slot[0] < 0 && slot[1] < 0 && slot[2] < 0;
    if (gray && chans.size() != 1 + (slot[3] >= 0 ? 1u : 0u)) { err = "exr: no R/G/B channels found"; return false; }
    int gray_ch = -1;
    if (gray) for (size_t c = 0; c < chans.size(); ++c) if ((int)c != slot[3]) gray_ch = (int)c;

    const int lines_per_block = compression == 3 || compression == 5 ? 16 : compression == 4 ? 32 : 1;
    std::vector<int> chan_words(chans.size());
    for (size_t c = 0; c < chans.size(); ++c) chan_words[c] = chans[c].type == 1 ? 1 : 2;
    const size_t ntx = tiled ? (size_t)((W + tile_w - 1) / tile_w) : 1, nty = tiled ? (size_t)((H + tile_h - 1) / tile_h) : 1;
    const size_t nblocks = tiled ? ntx * nty : (size_t)((H + lines_per_block - 1) / lines_per_block);
    if (nblocks > file.size() / 8) { err = "exr: offset table larger than the file"; return false; }
    std::vector<uint64_t> offsets(nblocks);
    for (auto &o : offsets) o = r.u64();
    if (!r.ok) { err = "exr: truncated offset table"; return false; }

    float *const rgba = alloc((size_t)W * H * 4);
    if (!rgba) { err = "exr: out of memory"; return false; }
    for (size_t i = 0; i < (size_t)W * H; ++i) { rgba[i * 4] = rgba[i * 4 + 1] = rgba[i * 4 + 2] = 0.f; rgba[i * 4 + 3] = 1.0f; }   // missing alpha = 1
    std::vector<std::string> errs(nblocks);
    parallel_for(nblocks, [&](size_t b) {
        std::string &err = errs[b];
        auto work = [&]() -> bool {
        std::vector<uint8_t> tmp, raw;
        // (written so that an offset near 2^64 cannot wrap the sum; 8 / 20 bytes = the scan-line / tile chunk header)
        const size_t hdr = tiled ? 20 : 8;
        if (offsets[b] > file.size() || file.size() - offsets[b] < hdr) { err = "exr: chunk offset beyond end of file"; return false; }
        Reader c{file.data() + offsets[b], file.data() + file.size()};
        long x0 = 0, y0, bw = W, nl;
        int32_t size;
        if (tiled) {   // tile chunk: tile coordinates, level (0,0), size
            const int32_t tx = c.i32(), ty = c.i32(), lx = c.i32(), ly = c.i32();
            size = c.i32();
            if (!c.ok || size < 0 || !c.need((size_t)size)) { err = "exr: truncated tile chunk"; return false; }
            if (lx != 0 || ly != 0 || tx < 0 || ty < 0 || (size_t)tx >= ntx || (size_t)ty >= nty) { err = "exr: tile outside the level-0 grid"; return false; }
            // the offset table is indexed by tile coordinates: entry b must be tile b, so no two of the parallel
            // workers ever write the same pixels
            if ((size_t)ty * ntx + (size_t)tx != b) { err = "exr: tile chunk does not match its offset-table entry"; return false; }
            x0 = (long)tx * tile_w; y0 = (long)ty * tile_h;
            bw = std::min<long>(tile_w, W - x0); nl = std::min<long>(tile_h, H - y0);
        } else {
            const int32_t y = c.i32();
            size = c.i32();
            if (!c.ok || size < 0 || !c.need((size_t)size)) { err = "exr: truncated chunk"; return false; }
            y0 = (long)y - dw[1];
            if (y0 < 0 || y0 >= H) { err = "exr: chunk outside the data window"; return false; }
            // entry b of the line-offset table is the block of rows [b*lines_per_block, ...): a chunk that starts
            // elsewhere is corrupt, and accepting it would let two parallel workers write the same rows
            if (y0 % lines_per_block != 0 || (size_t)(y0 / lines_per_block) != b) { err = "exr: chunk does not match its offset-table entry"; return false; }
            nl = std::min<long>(lines_per_block, H - y0);
        }
        const size_t line_bytes = px_bytes * (size_t)bw;
        const size_t expect = line_bytes * (size_t)nl;
        const uint8_t *data;
        if ((size_t)size == expect || compression == 0) {
            if ((size_t)size != expect) { err = "exr: raw chunk has the wrong size"; return false; }
            data = c.p;                                      // stored uncompressed
        } else if (compression == 4) {
            if (!piz_decode_block(c.p, (size_t)size, (int)bw, (int)nl, chan_words, raw, err)) return false;
            if (raw.size() != expect) { err = "exr: PIZ chunk decodes to the wrong size"; return false; }
            data = raw.data();
        } else if (compression == 5) {
            if (!pxr24_decode(c.p, (size_t)size, bw, nl, types, raw, err)) return false;
            data = raw.data();
        } else {
            tmp.resize(expect);
            if (compression == 1) {
                if (!rle_decode(c.p, (size_t)size, tmp, expect)) { err = "exr: corrupt RLE chunk"; return false; }
            } else {
                uLongf got = (uLongf)expect;
                if (uncompress(tmp.data(), &got, c.p, (uLong)size) != Z_OK || got != expect) { err = "exr: corrupt ZIP chunk"; return false; }
            }
            unpredict_and_interleave(tmp, raw);
            data = raw.data();
        }
        for (long l = 0; l < nl; ++l) {
            const uint8_t *line = data + (size_t)l * line_bytes;
            float *out = rgba + ((size_t)(y0 + l) * W + (size_t)x0) * 4;
            auto read_ch = [&](int ch, int dst_lo, int dst_hi) {
                const uint8_t *q = line + chpre[ch] * (size_t)bw;
                for (long x = 0; x < bw; ++x) {
                    float v;
                    if (chans[ch].type == 1) { uint16_t hv; memcpy(&hv, q + 2 * x, 2); v = half_to_float(hv); }
                    else if (chans[ch].type == 2) memcpy(&v, q + 4 * x, 4);
                    else { uint32_t u; memcpy(&u, q + 4 * x, 4); v = (float)u; }
                    for (int k = dst_lo; k <= dst_hi; ++k) out[x * 4 + k] = v;
                }
            };
            if (gray) read_ch(gray_ch, 0, 2);
            else for (int k = 0; k < 3; ++k) if (slot[k] >= 0) read_ch(slot[k], k, k);
            if (slot[3] >= 0) read_ch(slot[3], 3, 3);
        }
            return true;
        };
        try { (void)work(); } catch (const std::exception &e) { err = std::string("exr: ") + e.what(); }
    });
    for (auto &e : errs) if (!e.empty()) { err = e; return false; }
    w = (int)W; h = (int)H;
    return true;
}

static void put_bytes(std::vector<uint8_t> &v, const void *p, size_t n) { v.insert(v.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
static void put_str(std::vector<uint8_t> &v, const char *s) { put_bytes(v, s, strlen(s) + 1); }
static void put_i32(std::vector<uint8_t> &v, int32_t x) { put_bytes(v, &x, 4); }
static void put_f32(std::vector<uint8_t> &v, float x) { put_bytes(v, &x, 4); }
static void put_attr(std::vector<uint8_t> &v, const char *name, const char *type, const std::vector<uint8_t> &val)
{
    put_str(v, name); put_str(v, type); put_i32(v, (int32_t)val.size()); put_bytes(v, val.data(), val.size());
}

bool exr_encode(const float *rgba, int w, int h, std::vector<uint8_t> &file, std::string &err)
{
    if (w <= 0 || h <= 0 || !rgba) { err = "exr: bad image"; return false; }
    const bool zip = !(w < 16 && h < 16);
    file.clear();
    put_i32(file, 20000630);
    put_i32(file, 2);
    {   // header
        std::vector<uint8_t> ch;
        for (const char *n : {"A", "B", "G", "R"}) {
            put_str(ch, n); put_i32(ch, 2 /*FLOAT*/); ch.insert(ch.end(), 4, 0); put_i32(ch, 1); put_i32(ch, 1);
        }
        ch.push_back(0);
        put_attr(file, "channels", "chlist", ch);
        put_attr(file, "compression", "compression", {(uint8_t)(zip ? 3 : 0)});
        std::vector<uint8_t> box;
        put_i32(box, 0); put_i32(box, 0); put_i32(box, w - 1); put_i32(box, h - 1);
        put_attr(file, "dataWindow", "box2i", box);
        put_attr(file, "displayWindow", "box2i", box);
        put_attr(file, "lineOrder", "lineOrder", {0});
        std::vector<uint8_t> f1; put_f32(f1, 1.0f);
        put_attr(file, "pixelAspectRatio", "float", f1);
        std::vector<uint8_t> v2; put_f32(v2, 0.f); put_f32(v2, 0.f);
        put_attr(file, "screenWindowCenter", "v2f", v2);
        put_attr(file, "screenWindowWidth", "float", f1);
        file.push_back(0);
    }
    const int lpb = zip ? 16 : 1;
    const size_t nblocks = (size_t)((h + lpb - 1) / lpb);
    const size_t table = file.size();
    file.resize(table + nblocks * 8);
    const size_t line_bytes = (size_t)w * 16;
    static const int order[4] = {3, 2, 1, 0};   // A, B, G, R from RGBA
    std::vector<std::vector<uint8_t>> payload(nblocks);
    std::vector<std::string> errs(nblocks);
    parallel_for(nblocks, [&](size_t b) {
        try {
            std::vector<uint8_t> raw, t, z;
            const int y0 = (int)b * lpb, nl = std::min(lpb, h - y0);
            raw.resize(line_bytes * nl);
            for (int l = 0; l < nl; ++l)
                for (int c = 0; c < 4; ++c) {
                    float *dst = (float *)(raw.data() + (size_t)l * line_bytes + (size_t)c * w * 4);
                    const float *src = rgba + (size_t)(y0 + l) * w * 4 + order[c];
                    for (int x = 0; x < w; ++x) dst[x] = src[(size_t)x * 4];
                }
            if (zip) {
                const size_t n = raw.size(), half = (n + 1) / 2;
                t.resize(n);
                for (size_t i = 0, a = 0, bb = half; i < n;) {   // even bytes first, odd bytes second
                    t[a++] = raw[i++];
                    if (i < n) t[bb++] = raw[i++];
                }
                uint8_t prev = t[0];
                for (size_t i = 1; i < n; ++i) { const uint8_t cur = t[i]; t[i] = (uint8_t)(cur - prev + 128); prev = cur; }
                uLongf clen = compressBound((uLong)n);
                z.resize(clen);
                if (compress2(z.data(), &clen, t.data(), (uLong)n, 6) != Z_OK) { errs[b] = "exr: zlib compress failed"; return; }
                if (clen < n) { z.resize(clen); payload[b].swap(z); return; }   // otherwise stored raw, as the format prescribes
            }
            payload[b].swap(raw);
        } catch (const std::exception &e) { errs[b] = std::string("exr: ") + e.what(); }
    });
    for (auto &e : errs) if (!e.empty()) { err = e; return false; }
    for (size_t b = 0; b < nblocks; ++b) {
        const uint64_t off = file.size();
        memcpy(file.data() + table + b * 8, &off, 8);
        put_i32(file, (int32_t)b * lpb);
        put_i32(file, (int32_t)payload[b].size());
        put_bytes(file, payload[b].data(), payload[b].size());
    }
    return true;
}

}  // namespace codec
}  // namespace mid
