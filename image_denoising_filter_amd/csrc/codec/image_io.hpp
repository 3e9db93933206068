// image_io.hpp -- file codecs behind the reference's image I/O boundary (LoadImages
// src/main.cpp:145-229, SaveEXR :1699, lodepng::encode :1717).  Host-only code, no HIP.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <mutex>
#include <system_error>
#include <thread>

namespace mid {
namespace codec {

// How many host threads an image call made by THIS thread may use (mid_image_threads): 0 = the default, the machine's
// hardware concurrency capped at 16.  Per calling thread, so that a host which decodes or encodes many files at once -- one
// per thread, the way mi_denoise --animation does -- sets 1 on its workers without touching anyone else's calls.
inline int &thread_cap()
{
    static thread_local int cap = 0;
    return cap;
}

// Blocks of an image file are independent: run fn(i) for i in [0,n) on up to thread_cap() (default 16) host threads.
inline void parallel_for(size_t n, const std::function<void(size_t)> &fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t cap = thread_cap() > 0 ? (size_t)thread_cap() : std::min<size_t>(hw ? hw : 4, 16);
    const size_t nt = std::min<size_t>(cap, n);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr failed;                // the first exception a block raised (std::bad_alloc, say): rethrown on the calling thread
    std::mutex failed_mu;
    auto worker = [&] {
        try { for (size_t i = next++; i < n; i = next++) fn(i); }
        catch (...) { std::lock_guard<std::mutex> l(failed_mu); if (!failed) failed = std::current_exception(); next = n; }
    };
    std::vector<std::thread> th;
    try { for (size_t t = 1; t < nt; ++t) th.emplace_back(worker); }
    catch (const std::system_error &) {}      // no more threads to be had (EAGAIN under a thread limit): the ones that started, and this one, do the work
    worker();                                 // the calling thread is one of the workers
    for (auto &t : th) t.join();
    if (failed) std::rethrow_exception(failed);
}

// The decoders write straight into memory the caller provides: `alloc(n)` is called exactly once, after the header
// has been validated, and must return room for n elements (or nullptr = out of memory).  That is how frames are
// decoded directly into pinned staging memory (mid_image_load_pinned), like the reference memcpy's decoded pixels
// straight into its mapped staging buffer (src/main.cpp:1105-1142), with no intermediate copy.
using ByteAlloc = std::function<uint8_t *(size_t)>;
using FloatAlloc = std::function<float *(size_t)>;
bool png_decode_to(const std::vector<uint8_t> &file, int &w, int &h, const ByteAlloc &alloc, std::string &err);
bool exr_decode_to(const std::vector<uint8_t> &file, int &w, int &h, const FloatAlloc &alloc, std::string &err);

bool png_decode(const std::vector<uint8_t> &file, int &w, int &h, std::vector<uint8_t> &rgba, std::string &err);
bool png_encode(const uint8_t *rgba, int w, int h, std::vector<uint8_t> &file, std::string &err);

// Scanline OpenEXR: NONE / RLE / ZIPS / ZIP / PIZ, HALF / FLOAT / UINT channels -> RGBA float
// (missing alpha = 1.0, a single channel is replicated), like tinyexr's LoadEXR.
bool exr_decode(const std::vector<uint8_t> &file, int &w, int &h, std::vector<float> &rgba, std::string &err);
// 4 x FLOAT channels (A,B,G,R), ZIP (NONE below 16x16), like tinyexr's SaveEXR(data,w,h,4,0,...).
bool exr_encode(const float *rgba, int w, int h, std::vector<uint8_t> &file, std::string &err);

// One PIZ chunk -> nl scanlines, scanline-interleaved (see piz.cpp for the verification status).
bool piz_decode_block(const uint8_t *comp, size_t ncomp, int width, int nl, const std::vector<int> &chan_size,
                      std::vector<uint8_t> &out, std::string &err);

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err);
bool write_file(const std::string &path, const std::vector<uint8_t> &data, std::string &err);

}  // namespace codec
}  // namespace mid
