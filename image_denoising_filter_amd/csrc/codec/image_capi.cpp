// image_capi.cpp -- mid_image_load / mid_image_save / mid_image_free (host only).
#include "../common.hpp"
#include "image_io.hpp"

#include <cstdlib>
#include <exception>

using namespace mid;

static bool has_ext(const std::string &p, const char *ext)
{
    const size_t n = strlen(ext);
    if (p.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
        if (tolower((unsigned char)p[p.size() - n + i]) != ext[i]) return false;
    return true;
}

extern "C" int mid_image_load(const char *path, mid_image *out)
{
    MID_REQUIRE(path && out, "image_load: NULL argument");
    out->width = out->height = 0; out->format = 0; out->data = nullptr;
    std::vector<uint8_t> file;
    std::string err;
    if (!codec::read_file(path, file, err)) return set_error(MID_ERR_IO, "%s", err.c_str());
    int w = 0, h = 0;
    try {   // a corrupt header can ask for an absurd allocation: no exception may cross the C ABI
    if (has_ext(path, ".exr")) {                      // m_isHDR = extension == ".exr", src/main.cpp:1380
        std::vector<float> px;
        if (!codec::exr_decode(file, w, h, px, err)) return set_error(MID_ERR_IO, "%s: %s", path, err.c_str());
        out->data = malloc(px.size() * sizeof(float));
        if (!out->data) return set_error(MID_ERR_IO, "out of host memory");
        memcpy(out->data, px.data(), px.size() * sizeof(float));
        out->format = MID_FMT_RGBA32F;
    } else {
        std::vector<uint8_t> px;
        if (!codec::png_decode(file, w, h, px, err)) return set_error(MID_ERR_IO, "%s: %s", path, err.c_str());
        out->data = malloc(px.size());
        if (!out->data) return set_error(MID_ERR_IO, "out of host memory");
        memcpy(out->data, px.data(), px.size());
        out->format = MID_FMT_RGBA8;
    }
    } catch (const std::exception &e) {
        return set_error(MID_ERR_IO, "%s: %s", path, e.what());
    }
    out->width = w; out->height = h;
    return MID_OK;
}

extern "C" void mid_image_free(mid_image *img)
{
    if (img && img->data) { free(img->data); img->data = nullptr; }
}

extern "C" int mid_image_save(const char *path, const void *data, int32_t w, int32_t h, int32_t format)
{
    MID_REQUIRE(path && data && w > 0 && h > 0, "image_save: bad argument");
    std::vector<uint8_t> file;
    std::string err;
    bool ok;
    try {
    if (format == MID_FMT_RGBA32F) ok = codec::exr_encode((const float *)data, w, h, file, err);
    else if (format == MID_FMT_RGBA8) ok = codec::png_encode((const uint8_t *)data, w, h, file, err);
    else return set_error(MID_ERR_INVALID, "image_save: unknown format %d", format);
    if (!ok) return set_error(MID_ERR_IO, "%s: %s", path, err.c_str());
    if (!codec::write_file(path, file, err)) return set_error(MID_ERR_IO, "%s", err.c_str());
    } catch (const std::exception &e) {
        return set_error(MID_ERR_IO, "%s: %s", path, e.what());
    }
    return MID_OK;
}
