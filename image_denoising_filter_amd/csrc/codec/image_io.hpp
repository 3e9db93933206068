// image_io.hpp -- file codecs behind the reference's image I/O boundary (LoadImages
// src/main.cpp:145-229, SaveEXR :1699, lodepng::encode :1717).  Host-only code, no HIP.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mid {
namespace codec {

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
