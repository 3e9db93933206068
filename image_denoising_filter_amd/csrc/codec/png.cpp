// png.cpp -- minimal PNG codec over zlib for the image I/O boundary of the reference
// (lodepng::decode / lodepng::encode call sites: src/main.cpp:196,1717,1783,1916).
//
// Decode: every colour type / bit depth / interlace mode of the PNG specification, converted to
// 8-bit RGBA the way lodepng's default decode() does (16-bit samples keep their most significant
// byte, low bit depths are scaled to 0..255, palette + tRNS give alpha).  Encode: 8-bit RGBA,
// colour type 6, per-row adaptive filter (minimum sum of absolute differences), zlib level 6.
// lodepng is an un-vendored submodule of the reference (absent here); only the decoded pixels,
// not the compressed bytes, are part of the drop-in contract.
#include "image_io.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mid {
namespace codec {

static uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the per-scanline filters in place: `raw` holds (1 + rowbytes) * h bytes.
static bool unfilter(uint8_t *raw, size_t rowbytes, size_t h, size_t bpp, std::vector<uint8_t> &out, std::string &err)
{
    out.assign(rowbytes * h, 0);
    const uint8_t *prev = nullptr;
    for (size_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[y * (rowbytes + 1)];
        const uint8_t *in = raw + y * (rowbytes + 1) + 1;
        uint8_t *o = out.data() + y * rowbytes;
        for (size_t i = 0; i < rowbytes; ++i) {
            const int a = i >= bpp ? o[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int v;
            switch (ft) {
            case 0: v = in[i]; break;
            case 1: v = in[i] + a; break;
            case 2: v = in[i] + b; break;
            case 3: v = in[i] + ((a + b) >> 1); break;
            case 4: v = in[i] + paeth(a, b, c); break;
            default: err = "png: bad filter type"; return false;
            }
            o[i] = (uint8_t)v;
        }
        prev = o;
    }
    return true;
}

struct PngInfo {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> plte, trns;
    int channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
    size_t bits_pp() const { return (size_t)channels() * depth; }
};

// One decoded (unfiltered) sub-image -> RGBA8 pixels scattered into `dst` at (x0 + i*dx, y0 + j*dy).
static void to_rgba(const PngInfo &pi, const uint8_t *rows, size_t rowbytes, uint32_t sw, uint32_t sh,
                    uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy, uint8_t *dst)
{
    const int depth = pi.depth, maxv = (1 << (depth > 8 ? 8 : depth)) - 1;
    for (uint32_t j = 0; j < sh; ++j) {
        const uint8_t *r = rows + (size_t)j * rowbytes;
        for (uint32_t i = 0; i < sw; ++i) {
            uint32_t s[4] = {0, 0, 0, 0};    // raw samples
            uint32_t s16[4] = {0, 0, 0, 0};  // full-precision samples (for tRNS colour-key compare)
            const int nch = pi.channels();
            for (int c = 0; c < nch; ++c) {
                if (depth == 8) { s[c] = s16[c] = r[(size_t)i * nch + c]; }
                else if (depth == 16) { const uint8_t *q = r + ((size_t)i * nch + c) * 2; s16[c] = (uint32_t)q[0] << 8 | q[1]; s[c] = q[0]; }
                else {
                    const size_t bit = (size_t)i * depth;   // nch == 1 for depths < 8
                    s[c] = s16[c] = (r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                }
            }
            uint8_t R, G, B, A = 255;
            switch (pi.ctype) {
            case 0:
                R = G = B = depth < 8 ? (uint8_t)(s[0] * 255 / maxv) : (uint8_t)s[0];
                if (pi.trns.size() >= 2 && s16[0] == ((uint32_t)pi.trns[0] << 8 | pi.trns[1])) A = 0;
                break;
            case 2:
                R = s[0]; G = s[1]; B = s[2];
                if (pi.trns.size() >= 6 && s16[0] == ((uint32_t)pi.trns[0] << 8 | pi.trns[1]) &&
                    s16[1] == ((uint32_t)pi.trns[2] << 8 | pi.trns[3]) && s16[2] == ((uint32_t)pi.trns[4] << 8 | pi.trns[5])) A = 0;
                break;
            case 3: {
                const size_t k = s[0];
                if (k * 3 + 2 < pi.plte.size()) { R = pi.plte[k * 3]; G = pi.plte[k * 3 + 1]; B = pi.plte[k * 3 + 2]; }
                else { R = G = B = 0; }
                if (k < pi.trns.size()) A = pi.trns[k];
                break;
            }
            case 4: R = G = B = s[0]; A = s[1]; break;
            default: R = s[0]; G = s[1]; B = s[2]; A = s[3]; break;
            }
            uint8_t *o = dst + 4 * ((size_t)(y0 + j * dy) * pi.w + (x0 + i * dx));
            o[0] = R; o[1] = G; o[2] = B; o[3] = A;
        }
    }
}

bool png_decode(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgba, std::string &err)
{
    return png_decode_to(file, w, h, [&](size_t n) { rgba.assign(n, 0); return rgba.data(); }, err);
}

bool png_decode_to(const std::vector<uint8_t> &file, int &w, int &h, const ByteAlloc &alloc, std::string &err)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) { err = "png: not a PNG file"; return false; }
    PngInfo pi;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    bool seen_ihdr = false, seen_iend = false;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        if (len > file.size() - pos - 12) { err = "png: truncated chunk"; return false; }
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        const uint32_t crc = be32(&file[pos + 8 + len]);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) { err = "png: chunk CRC mismatch"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "png: bad IHDR"; return false; }
            pi.w = be32(data); pi.h = be32(data + 4); pi.depth = data[8]; pi.ctype = data[9];
            if (data[10] != 0 || data[11] != 0 || data[12] > 1) { err = "png: unsupported compression/filter/interlace method"; return false; }
            pi.interlace = data[12];
            seen_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) pi.plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) pi.trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) { seen_iend = true; break; }
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || !seen_iend || idat.empty()) { err = "png: missing IHDR/IDAT/IEND"; return false; }
    if (pi.w == 0 || pi.h == 0 || pi.w > 65536 || pi.h > 65536) { err = "png: bad dimensions"; return false; }
    if ((double)pi.w * pi.h / 8 > (double)idat.size() * 1100.0) { err = "png: dimensions larger than the data can hold"; return false; }
    const int d = pi.depth, t = pi.ctype;
    const bool ok = (t == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (t == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                    ((t == 2 || t == 4 || t == 6) && (d == 8 || d == 16));
    if (!ok) { err = "png: illegal colour type / bit depth"; return false; }
    if (t == 3 && pi.plte.empty()) { err = "png: palette image without PLTE"; return false; }

    // sub-images: the whole picture, or the seven Adam7 passes
    static const uint32_t ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Sub { uint32_t w, h, x0, y0, dx, dy; size_t rowbytes; };
    std::vector<Sub> subs;
    const size_t bits = pi.bits_pp();
    if (!pi.interlace) subs.push_back({pi.w, pi.h, 0, 0, 1, 1, (pi.w * bits + 7) / 8});
    else
        for (int p = 0; p < 7; ++p) {
            const uint32_t sw = (pi.w + adx[p] - 1 - ax0[p]) / adx[p], sh = (pi.h + ady[p] - 1 - ay0[p]) / ady[p];
            if (pi.w > ax0[p] && pi.h > ay0[p] && sw && sh) subs.push_back({sw, sh, ax0[p], ay0[p], adx[p], ady[p], (sw * bits + 7) / 8});
        }
    size_t raw_size = 0;
    for (auto &s : subs) raw_size += (s.rowbytes + 1) * s.h;
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    const int zrc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || got != raw_size) { err = "png: zlib stream is corrupt or has the wrong size"; return false; }

    uint8_t *const rgba = alloc((size_t)pi.w * pi.h * 4);
    if (!rgba) { err = "png: out of memory"; return false; }
    memset(rgba, 0, (size_t)pi.w * pi.h * 4);
    const size_t bpp = bits >= 8 ? bits / 8 : 1;
    size_t off = 0;
    std::vector<uint8_t> rows;
    for (auto &s : subs) {
        if (!unfilter(raw.data() + off, s.rowbytes, s.h, bpp, rows, err)) return false;
        to_rgba(pi, rows.data(), s.rowbytes, s.w, s.h, s.x0, s.y0, s.dx, s.dy, rgba);
        off += (s.rowbytes + 1) * s.h;
    }
    w = (int)pi.w; h = (int)pi.h;
    return true;
}

static void put_chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *data, size_t len)
{
    put_be32(out, (uint32_t)len);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    put_be32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), out.data() + start, (uInt)(len + 4)));
}

bool png_encode(const uint8_t *rgba, int w, int h, std::vector<uint8_t> &file, std::string &err)
{
    if (w <= 0 || h <= 0 || !rgba) { err = "png: bad image"; return false; }
    const size_t rowbytes = (size_t)w * 4, bpp = 4;
    std::vector<uint8_t> raw((rowbytes + 1) * h);
    parallel_for((size_t)h, [&](size_t yy) {            // the filter choice of a row depends on the row above only
        const int y = (int)yy;
        std::vector<uint8_t> cand(rowbytes);
        const uint8_t *cur = rgba + (size_t)y * rowbytes, *prev = y ? cur - rowbytes : nullptr;
        uint8_t *dst = raw.data() + (size_t)y * (rowbytes + 1);
        long best = -1;
        for (int ft = 0; ft < 5; ++ft) {
            long sum = 0;
            for (size_t i = 0; i < rowbytes; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
                int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? ((a + b) >> 1) : paeth(a, b, c);
                const uint8_t v = (uint8_t)(cur[i] - pred);
                cand[i] = v;
                sum += v < 128 ? v : 256 - v;
            }
            if (best < 0 || sum < best) { best = sum; dst[0] = (uint8_t)ft; memcpy(dst + 1, cand.data(), rowbytes); }
        }
    });
    // One zlib stream built from independently deflated segments (pigz style): every segment but the last
    // ends on a full flush, so the raw-deflate pieces concatenate into a valid stream; header and Adler-32
    // of the whole payload are added around them.  Any inflater (lodepng, libpng, Pillow) reads it.
    const size_t seg_bytes = std::max<size_t>((size_t)1 << 20, (rowbytes + 1) * 16);
    const size_t nseg = (raw.size() + seg_bytes - 1) / seg_bytes;
    std::vector<std::vector<uint8_t>> parts(nseg);
    std::vector<int> bad(nseg, 0);
    parallel_for(nseg, [&](size_t si) {
        const size_t off = si * seg_bytes, len = std::min(seg_bytes, raw.size() - off);
        z_stream zs{};
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad[si] = 1; return; }
        parts[si].resize(deflateBound(&zs, (uLong)len) + 16);
        zs.next_in = (Bytef *)(raw.data() + off); zs.avail_in = (uInt)len;
        zs.next_out = parts[si].data(); zs.avail_out = (uInt)parts[si].size();
        const int rc = deflate(&zs, si + 1 == nseg ? Z_FINISH : Z_FULL_FLUSH);
        if ((si + 1 == nseg && rc != Z_STREAM_END) || (si + 1 != nseg && (rc != Z_OK || zs.avail_in != 0))) bad[si] = 1;
        parts[si].resize(zs.total_out);
        deflateEnd(&zs);
    });
    for (int b : bad) if (b) { err = "png: zlib compress failed"; return false; }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x9c);
    for (auto &pz : parts) z.insert(z.end(), pz.begin(), pz.end());
    const uLong ad = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    z.push_back(ad >> 24); z.push_back(ad >> 16); z.push_back(ad >> 8); z.push_back(ad);
    const size_t clen = z.size();
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    file.assign(sig, sig + 8);
    uint8_t ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    put_chunk(file, "IHDR", ihdr, 13);
    put_chunk(file, "IDAT", z.data(), clen);
    put_chunk(file, "IEND", nullptr, 0);
    return true;
}

}  // namespace codec
}  // namespace mid
