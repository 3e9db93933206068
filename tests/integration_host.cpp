// integration_host.cpp -- TEST RESOURCE: a C++20 host program that drives libmi_denoise.so with the call sequence INTEGRATION.md
// gives for the body of ComputeApplication::RunOnGPU (reference: src/main.cpp:1307-1730), from plain std::vector buffers like the
// reference's imageData / imageDataHDR / layerData / resultHDRData (pageable memory: the library bounces it).  Compiled and run
// by tests/test_gpu_integration_host.py, which holds its outputs against the Python-driven calls of the same entry points.
//   integration_host <mode> <w> <h> <hdr:0|1> <in.raw> <out.raw> [extra.raw ...]
//   mode: bilateral | linear | layers (extras = RGBA8 layers) | nlm (extras = further frames; in.raw is the target)
//         sharded (INTEGRATION.md "An animation over several GPUs", one rank: in.raw + extras are the sequence, k = 2; out.raw holds
//         every output frame, then 4 floats: the call's device timeline from mid_comm_last_timeline)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "mi_denoise.h"

#define MID_CHECK(call) do { if ((call) != MID_OK) \
    throw std::runtime_error(std::string(#call) + ": " + mid_last_error()); } while (0)

struct Pixel { float r, g, b, a; };                                     // src/main.cpp:39-41

static std::vector<uint8_t> slurp(const char *path, size_t want)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<uint8_t> v((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (v.size() != want) throw std::runtime_error(std::string(path) + ": unexpected size");
    return v;
}

int main(int argc, char **argv)
try {
    if (argc < 7) return 2;
    const std::string mode = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    const bool m_isHDR = atoi(argv[4]) != 0, m_linear = mode == "linear", m_useLayers = mode == "layers", m_nlmFilter = mode == "nlm";
    const int fmt = m_isHDR ? MID_FMT_RGBA32F : MID_FMT_RGBA8;
    const size_t inBytes = size_t(w) * h * (m_isHDR ? sizeof(Pixel) : sizeof(uint32_t));
    const size_t outBytes = size_t(w) * h * sizeof(Pixel);
    std::vector<std::vector<uint8_t>> imageData;                        // imageData / imageDataHDR of the reference
    imageData.push_back(slurp(argv[5], inBytes));
    std::vector<std::vector<uint8_t>> layerData;
    for (int i = 7; i < argc; ++i) {
        if (m_useLayers) layerData.push_back(slurp(argv[i], size_t(w) * h * 4));
        else imageData.push_back(slurp(argv[i], inBytes));
    }
    auto hostPtr = [&](int i) -> const void * { return imageData[i].data(); };

    mid_ctx *ctx = nullptr;
    MID_CHECK(mid_ctx_create(0, &ctx));
    if (mode == "sharded") {
        const int rank = 0, world = 1, nFrames = (int)imageData.size();
        uint8_t id[MID_COMM_ID_BYTES];
        if (rank == 0) MID_CHECK(mid_comm_unique_id(id));                   // ncclGetUniqueId
        mid_comm *comm = nullptr;
        MID_CHECK(mid_comm_create(ctx, id, rank, world, &comm));            // ncclCommInitRank on ctx's device
        int start, count;
        MID_CHECK(mid_shard_block(nFrames, world, rank, &start, &count));
        std::vector<void *> dBlock(count), dOutv(count);
        for (int i = 0; i < count; ++i) {
            MID_CHECK(mid_alloc(ctx, inBytes, &dBlock[i]));
            MID_CHECK(mid_alloc(ctx, outBytes, &dOutv[i]));
            MID_CHECK(mid_memcpy_h2d(ctx, dBlock[i], hostPtr(start + i), inBytes, nullptr));
        }
        mid_nlm_params p{w, h, 0.5f, -7, 7, -3, 3, fmt};
        MID_CHECK(mid_nlm_temporal_sharded(comm, &p, dBlock.data(), nFrames, /*k*/2, (mid_pixel *const *)dOutv.data(), nullptr));
        MID_CHECK(mid_stream_sync(ctx, nullptr));
        float t[4];
        MID_CHECK(mid_comm_last_timeline(comm, t));
        std::ofstream o(argv[6], std::ios::binary);
        std::vector<Pixel> frame(size_t(w) * h);
        for (int i = 0; i < count; ++i) {
            MID_CHECK(mid_memcpy_d2h(ctx, frame.data(), dOutv[i], outBytes, nullptr));
            o.write((const char *)frame.data(), (std::streamsize)outBytes);
            mid_free(ctx, dBlock[i]); mid_free(ctx, dOutv[i]);
        }
        o.write((const char *)t, sizeof t);
        MID_CHECK(mid_comm_destroy(comm));
        mid_ctx_destroy(ctx);
        return o.good() ? 0 : 3;
    }
    void *dTarget = nullptr, *dOut = nullptr;
    MID_CHECK(mid_alloc(ctx, inBytes, &dTarget));
    MID_CHECK(mid_alloc(ctx, outBytes, &dOut));
    (void)mid_range_push("integration_host");                           // a ROCTx range when a profiler listens, nothing otherwise
    MID_CHECK(mid_memcpy_h2d(ctx, dTarget, hostPtr(0), inBytes, nullptr));

    if (!m_nlmFilter && !m_useLayers) {
        mid_bilateral_params p{w, h, 2.0f, 0.2f, 20, m_linear ? MID_LAYOUT_LINEAR : MID_LAYOUT_TEXTURE, fmt};
        MID_CHECK(mid_bilateral(ctx, &p, dTarget, (mid_pixel *)dOut, nullptr));
    } else if (m_useLayers) {
        std::vector<void *> dLayers(layerData.size());
        for (size_t i = 0; i < layerData.size(); ++i) {
            MID_CHECK(mid_alloc(ctx, size_t(w) * h * 4, &dLayers[i]));
            MID_CHECK(mid_memcpy_h2d(ctx, dLayers[i], layerData[i].data(), size_t(w) * h * 4, nullptr));
        }
        mid_bilateral_params p{w, h, 2.0f, 0.2f, 20, MID_LAYOUT_TEXTURE, fmt};
        MID_CHECK(mid_bilateral_layers(ctx, &p, dTarget, (const uint32_t *const *)dLayers.data(), (int)dLayers.size(), (mid_pixel *)dOut, nullptr));
        for (void *d : dLayers) mid_free(ctx, d);
    } else {
        mid_nlm_params p{w, h, 0.5f, -7, 7, -3, 3, fmt};
        const int nFrames = (int)imageData.size();
        void *dW = nullptr, *dNb = nullptr;
        MID_CHECK(mid_alloc(ctx, size_t(w) * h * sizeof(mid_weightinfo), &dW));
        MID_CHECK(mid_alloc(ctx, inBytes, &dNb));
        MID_CHECK(mid_memset(ctx, dW, 0, size_t(w) * h * sizeof(mid_weightinfo), nullptr));
        for (int f = 0; f < nFrames; ++f) {
            MID_CHECK(mid_memcpy_h2d(ctx, dNb, hostPtr(f), inBytes, nullptr));
            MID_CHECK(mid_nlm_accum(ctx, &p, dTarget, dNb, (mid_weightinfo *)dW, nullptr));
        }
        mid_normalize_params np{w, h};
        MID_CHECK(mid_normalize(ctx, &np, (const mid_weightinfo *)dW, (mid_pixel *)dOut, nullptr));
        mid_free(ctx, dW); mid_free(ctx, dNb);
    }

    std::vector<Pixel> resultHDRData(size_t(w) * h);
    std::vector<unsigned char> resultData(size_t(w) * h * 4);
    if (m_isHDR) {
        MID_CHECK(mid_memcpy_d2h(ctx, resultHDRData.data(), dOut, outBytes, nullptr));
    } else {
        void *dU8 = nullptr;
        MID_CHECK(mid_alloc(ctx, size_t(w) * h * 4, &dU8));
        MID_CHECK(mid_pack_u8(ctx, (const float *)dOut, size_t(w) * h * 4, (uint8_t *)dU8, nullptr));
        MID_CHECK(mid_memcpy_d2h(ctx, resultData.data(), dU8, size_t(w) * h * 4, nullptr));
        mid_free(ctx, dU8);
    }
    MID_CHECK(mid_stream_sync(ctx, nullptr));
    (void)mid_range_pop();
    mid_free(ctx, dTarget); mid_free(ctx, dOut);
    mid_ctx_destroy(ctx);

    std::ofstream o(argv[6], std::ios::binary);
    if (m_isHDR) o.write((const char *)resultHDRData.data(), (std::streamsize)outBytes);
    else o.write((const char *)resultData.data(), (std::streamsize)resultData.size());
    return o.good() ? 0 : 3;
} catch (const std::exception &e) {
    fprintf(stderr, "integration_host: %s\n", e.what());
    return EXIT_FAILURE;                                                // as main() of the reference does (:1987-1991)
}
