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

// Shared body of mid_image_load / mid_image_load_pinned: `get(bytes)` provides the pixel memory (called once, after
// the header checks), `drop(ptr)` takes it back when decoding fails afterwards.
template <class Get, class Drop>
static int load_impl(const char *path, mid_image *out, Get get, Drop drop)
{
    MID_REQUIRE(path && out, "image_load: NULL argument");
    out->width = out->height = 0; out->format = 0; out->data = nullptr;
    std::vector<uint8_t> file;
    std::string err;
    if (!codec::read_file(path, file, err)) return set_error(MID_ERR_IO, "%s", err.c_str());
    int w = 0, h = 0;
    void *mem = nullptr;
    bool ok = false;
    try {   // a corrupt header can ask for an absurd allocation: no exception may cross the C ABI
        if (has_ext(path, ".exr")) {                      // m_isHDR = extension == ".exr", src/main.cpp:1380
            ok = codec::exr_decode_to(file, w, h, [&](size_t n) { mem = get(n * sizeof(float)); return (float *)mem; }, err);
            out->format = MID_FMT_RGBA32F;
        } else {
            ok = codec::png_decode_to(file, w, h, [&](size_t n) { mem = get(n); return (uint8_t *)mem; }, err);
            out->format = MID_FMT_RGBA8;
        }
    } catch (const std::exception &e) {
        err = e.what();
        ok = false;
    }
    if (!ok) {
        if (mem) drop(mem);
        return set_error(MID_ERR_IO, "%s: %s", path, err.c_str());
    }
    out->data = mem;
    out->width = w; out->height = h;
    return MID_OK;
}

extern "C" int mid_image_load(const char *path, mid_image *out)
{
    return load_impl(path, out, [](size_t bytes) { return malloc(bytes ? bytes : 1); }, [](void *p) { free(p); });
}

extern "C" int mid_image_load_pinned(mid_ctx *ctx, const char *path, mid_image *out)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    return load_impl(path, out,
                     [](size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; },
                     [](void *p) { (void)hipHostFree(p); });
}

extern "C" int mid_image_free_pinned(mid_ctx *ctx, mid_image *img)
{
    Bind b(ctx, nullptr);
    if (b.rc) return b.rc;
    if (img && img->data) { MID_HIP(hipHostFree(img->data)); img->data = nullptr; }
    return MID_OK;
}

extern "C" int mid_image_threads(int n)
{
    int &cap = codec::thread_cap();
    const int before = cap;
    if (n >= 0) cap = n > 256 ? 256 : n;
    return before;
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
