/*
 * oracle.c -- CPU restatement of the reference's per-pixel filters (see oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product path.
 * a7 is pinned against oracle/_ref and tests/golden; a1-a6 are "parity unpinned"
 * (no reference test, golden vector or runnable shader build exists).
 *
 * Policies where the reference is formally undefined:
 *   - GLSL pow(x, 2)  == x*x (negative bases are undefined in GLSL; every driver lowers to a mul)
 *   - out-of-bounds texelFetch == vec4(0) and STILL contributes its weight to the norm
 *   - the WeightInfo buffer starts at zero (the reference never clears a fresh allocation)
 *   - float -> u8 is C truncation; clamped to [0,255] only where the C cast is undefined
 *
 * Build with -ffp-contract=off: every statement below is one IEEE fp32 (or, for a7,
 * fp64) operation in the order the reference text gives.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

typedef struct { float x, y, z, w; } vec4;

static inline vec4 fetch2d(const float *img, int w, int h, int x, int y)
{
    vec4 r = {0.f, 0.f, 0.f, 0.f};
    if (x < 0 || y < 0 || x >= w || y >= h) return r;      /* policy: OOB texel = 0 */
    const float *p = img + 4 * ((long)y * w + x);
    r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    return r;
}

static inline vec4 fetch1d(const float *img, long n, long idx)
{
    vec4 r = {0.f, 0.f, 0.f, 0.f};
    if (idx < 0 || idx >= n) return r;                       /* policy: OOB texel = 0 */
    const float *p = img + 4 * idx;
    r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    return r;
}

/* One bilateral tap, shaders/bialteral.comp:55-68 (identical text in the linear and layers
 * shaders): returns resultWeight. i,j are the loop counters, a the centre, b the tap. */
static inline float bilateral_weight(int i, int j, vec4 a, vec4 b, float sigma_s, float sigma_c)
{
    float spatialDistance = sqrtf((float)i * (float)i + (float)j * (float)j);   /* :55 */
    float ts = spatialDistance / sigma_s;
    float spatialWeight = expf(-0.5f * (ts * ts));                               /* :56 */
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float colorDistance = sqrtf(dx * dx + dy * dy + dz * dz);                    /* :60-62 */
    float tc = colorDistance / sigma_c;
    float colorWeight = expf(-0.5f * (tc * tc));                                 /* :63 */
    return spatialWeight * colorWeight;                                          /* :65 */
}

/* a1 -- shaders/bialteral.comp:29-82.  i = x offset (outer), j = y offset (inner). */
void orc_bilateral_texture(const float *img, int w, int h, int radius,
                           float sigma_s, float sigma_c, float *out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            vec4 texColor = fetch2d(img, w, h, x, y);                            /* :31 */
            float normWeight = 0.f;
            vec4 wc = {0.f, 0.f, 0.f, 0.f};
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j) {
                    vec4 cur = fetch2d(img, w, h, x + i, y + j);                 /* :58-59 */
                    float rw = bilateral_weight(i, j, texColor, cur, sigma_s, sigma_c);
                    wc.x += cur.x * rw; wc.y += cur.y * rw;                      /* :67 */
                    wc.z += cur.z * rw; wc.w += cur.w * rw;
                    normWeight += rw;                                            /* :68 */
                }
            float *o = out + 4 * ((long)y * w + x);                              /* :81 */
            o[0] = wc.x / normWeight; o[1] = wc.y / normWeight;                  /* :72 */
            o[2] = wc.z / normWeight; o[3] = wc.w / normWeight;
        }
}

/* a2 -- shaders/bialteral_linear.comp:29-81.  Flat index c + j + i*w: i = row offset
 * (outer), j = column offset (inner); columns overflow into the adjacent row. */
void orc_bilateral_linear(const float *img, int w, int h, int radius,
                          float sigma_s, float sigma_c, float *out)
{
    const long n = (long)w * h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            long c = (long)x + (long)y * w;                                      /* :79 */
            vec4 texColor = fetch1d(img, n, c);                                  /* :31 */
            float normWeight = 0.f;
            vec4 wc = {0.f, 0.f, 0.f, 0.f};
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j) {
                    vec4 cur = fetch1d(img, n, c + j + (long)i * w);             /* :58 */
                    float rw = bilateral_weight(i, j, texColor, cur, sigma_s, sigma_c);
                    wc.x += cur.x * rw; wc.y += cur.y * rw;
                    wc.z += cur.z * rw; wc.w += cur.w * rw;
                    normWeight += rw;
                }
            float *o = out + 4 * c;
            o[0] = wc.x / normWeight; o[1] = wc.y / normWeight;
            o[2] = wc.z / normWeight; o[3] = wc.w / normWeight;
        }
}

static inline vec4 fetch2d_u8(const uint8_t *img, int w, int h, int x, int y)
{
    vec4 r = {0.f, 0.f, 0.f, 0.f};
    if (x < 0 || y < 0 || x >= w || y >= h) return r;
    const uint8_t *p = img + 4 * ((long)y * w + x);
    r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f;  /* UNORM decode, src/texture.cpp:16 */
    r.z = (float)p[2] / 255.0f; r.w = (float)p[3] / 255.0f;
    return r;
}

/* a3 -- shaders/bialteral_layers.comp:27-71. */
void orc_bilateral_layers_accum(const float *img, const uint8_t *layer, int w, int h,
                                int radius, float sigma_s, float sigma_c, orc_weightinfo *W)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            vec4 layerColor = fetch2d_u8(layer, w, h, x, y);                     /* :29 */
            float normWeight = 0.f;
            vec4 wc = {0.f, 0.f, 0.f, 0.f};
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j) {
                    vec4 cur = fetch2d_u8(layer, w, h, x + i, y + j);            /* :47 */
                    float rw = bilateral_weight(i, j, layerColor, cur, sigma_s, sigma_c);
                    vec4 col = fetch2d(img, w, h, x + i, y + j);                 /* :55 */
                    wc.x += col.x * rw; wc.y += col.y * rw;
                    wc.z += col.z * rw; wc.w += col.w * rw;
                    normWeight += rw;                                            /* :56 */
                }
            orc_weightinfo *o = W + ((long)y * w + x);
            o->wc[0] += wc.x; o->wc[1] += wc.y; o->wc[2] += wc.z; o->wc[3] += wc.w;  /* :60 */
            o->nw += normWeight;                                                 /* :61 */
        }
}

/* a4 -- shaders/nonlocal.comp:28-72.  One invocation (one pixel) of the shader. */
static void nlm_invocation(const float *target, const float *neighbour, int w, int h, float h2,
                           int search_lo, int search_hi, int patch_lo, int patch_hi, int px, int py, orc_weightinfo *W)
{
    float normWeight = 0.001f;                                           /* :32 */
    vec4 wc = {0.f, 0.f, 0.f, 0.f};
    for (int y = py + search_lo; y < py + search_hi; ++y)                /* :36 */
        for (int x = px + search_lo; x < px + search_hi; ++x) {          /* :38 */
            float colorDistance = 0.0f;
            for (int j = patch_lo; j < patch_hi; ++j)                    /* :42 */
                for (int i = patch_lo; i < patch_hi; ++i) {              /* :44 */
                    vec4 t = fetch2d(target, w, h, px + i, py + j);      /* :46 */
                    vec4 n = fetch2d(neighbour, w, h, x + i, y + j);     /* :47 */
                    float dx = t.x - n.x, dy = t.y - n.y, dz = t.z - n.z;
                    colorDistance += dx * dx + dy * dy + dz * dz;        /* :49-51 */
                }
            float weight = expf(-colorDistance / h2);                    /* :55 */
            vec4 c = fetch2d(neighbour, w, h, x, y);                     /* :56 */
            wc.x += c.x * weight; wc.y += c.y * weight;
            wc.z += c.z * weight; wc.w += c.w * weight;
            normWeight += weight;                                        /* :57 */
        }
    orc_weightinfo *o = W + ((long)py * w + px);
    o->wc[0] += wc.x; o->wc[1] += wc.y; o->wc[2] += wc.z; o->wc[3] += wc.w;  /* :61 */
    o->nw += normWeight;                                                 /* :62 */
}

void orc_nlm_accum(const float *target, const float *neighbour, int w, int h, float hparam,
                   int search_lo, int search_hi, int patch_lo, int patch_hi, orc_weightinfo *W)
{
    const float h2 = hparam * hparam;                                            /* pow(h,2.f) :55 */
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px)
            nlm_invocation(target, neighbour, w, h, h2, search_lo, search_hi, patch_lo, patch_hi, px, py, W);
}

/* The same invocations spread over OpenMP threads (rows are independent: identical results); used by
 * bench.py's cpu_baseline so that the CPU figure is for the SAME workload as the GPU metric. */
void orc_nlm_accum_mt(const float *target, const float *neighbour, int w, int h, float hparam,
                      int search_lo, int search_hi, int patch_lo, int patch_hi, orc_weightinfo *W, int num_threads)
{
    const float h2 = hparam * hparam;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px)
            nlm_invocation(target, neighbour, w, h, h2, search_lo, search_hi, patch_lo, patch_hi, px, py, W);
}

/* a5 -- shaders/normalize.comp:29-44. */
void orc_normalize(const orc_weightinfo *W, int w, int h, float *out)
{
    const long n = (long)w * h;
    for (long c = 0; c < n; ++c) {
        float *o = out + 4 * c;
        if (W[c].nw == 0.0f) {                                                   /* :36 */
            o[0] = 1.0f; o[1] = 0.0f; o[2] = 1.0f; o[3] = 1.0f;                  /* :38 */
        } else {
            o[0] = W[c].wc[0] / W[c].nw; o[1] = W[c].wc[1] / W[c].nw;            /* :42 */
            o[2] = W[c].wc[2] / W[c].nw; o[3] = W[c].wc[3] / W[c].nw;
        }
    }
}

/* a6 */
void orc_unpack_u8_unorm(const uint8_t *in, long n, float *out)
{
    for (long i = 0; i < n; ++i) out[i] = (float)in[i] / 255.0f;
}

void orc_unpack_u8_cpu(const uint8_t *in, long n, float *out)
{
    for (long i = 0; i < n; ++i) out[i] = (float)in[i] * (1.0f / 255.0f);       /* main.cpp:1804 */
}

void orc_pack_u8(const float *in, long n, uint8_t *out)
{
    for (long i = 0; i < n; ++i) {
        float v = 255.0f * in[i];                                               /* main.cpp:99 */
        /* (unsigned char)v truncates toward zero and is defined for -1 < v < 256.
         * Outside that (and for NaN) the C cast is undefined: clamp. */
        if (!(v > -1.0f)) out[i] = 0;
        else if (v >= 256.0f) out[i] = 255;
        else out[i] = (uint8_t)v;
    }
}

/* a7 -- src/main.cpp:1819-1865.  Mixed precision exactly as the C++ overloads resolve:
 * pow(int,int), pow(float,int) -> double; sqrt/exp on double; results stored to float. */
void orc_cpu_bilateral(const float *in, int w, int h, int radius, float spatialSigma,
                       float colorSigma, int blue_bug, int num_threads, float *out)
{
    const long n = (long)w * h;
    const int windowSize = radius;                                              /* :1819 */
    memset(out, 0, (size_t)n * 4 * sizeof(float));                              /* Pixel{} :1815 */
    (void)num_threads;
    for (int y = windowSize; y <= h - windowSize; ++y) {                        /* :1824 */
        if (y >= h) break;  /* row h does not exist in the output; the reference writes past it (UB) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads)
#endif
        for (int x = windowSize; x <= w - windowSize; ++x) {                    /* :1828 */
            long c = (long)y * w + x;
            if (c >= n) continue;
            vec4 texColor = fetch1d(in, n, c);                                  /* :1830 */
            float normWeight = 0.0f;
            float wr = 0.f, wg = 0.f, wb = 0.f;
            for (int i = -windowSize; i <= windowSize; ++i)
                for (int j = -windowSize; j <= windowSize; ++j) {
                    float spatialDistance =
                        (float)sqrt((double)(float)pow((double)i, 2.0) + pow((double)j, 2.0)); /* :1844 */
                    float spatialWeight =
                        (float)exp(-0.5 * pow((double)(spatialDistance / spatialSigma), 2.0)); /* :1845 */
                    vec4 cur = fetch1d(in, n, (long)w * (i + y) + j + x);       /* :1847 */
                    double db = blue_bug ? pow((double)(texColor.z - texColor.z), 2.0)
                                         : pow((double)(texColor.z - cur.z), 2.0);
                    float colorDistance = (float)sqrt(pow((double)(texColor.x - cur.x), 2.0)
                                                    + pow((double)(texColor.y - cur.y), 2.0)
                                                    + db);                      /* :1848-1850 */
                    float colorWeight =
                        (float)exp(-0.5 * pow((double)(colorDistance / colorSigma), 2.0));     /* :1851 */
                    float resultWeight = spatialWeight * colorWeight;           /* :1853 */
                    wr += cur.x * resultWeight;                                 /* :1855-1857 */
                    wg += cur.y * resultWeight;
                    wb += cur.z * resultWeight;
                    normWeight += resultWeight;                                 /* :1859 */
                }
            float *o = out + 4 * c;
            o[0] = wr / normWeight; o[1] = wg / normWeight;                     /* :1863 */
            o[2] = wb / normWeight; o[3] = 1.0f;
        }
    }
}
