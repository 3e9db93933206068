/*
 * oracle.h -- CPU restatement of the reference's per-pixel filters.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (libmi_denoise.so, the mi_denoise
 * CLI, the Python package's product path) may include, link or call this.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker.
 *
 * Parity status:
 *   orc_cpu_bilateral (a7)  -- PINNED: checked bit-for-bit against oracle/_ref (the
 *                              reference's own CPU loop src/main.cpp:1827-1864 compiled
 *                              from where it lies) and against tests/golden/cpu_bilateral_*.
 *   everything else (a1-a6) -- "parity unpinned": the reference holds no test, golden
 *                              vector or runnable build for its GLSL shaders (no Vulkan /
 *                              GLSL toolchain here), so these follow the shader text
 *                              statement by statement under the policies of SURVEY.md 8a.
 *
 * All images are row-major RGBA float (16 B/pixel) unless stated; citations are
 * relative to /root/reference.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* std430 WeightInfo {vec4 weightColor; float normWeight;} = 32 B stride
 * (shaders/nonlocal.comp:10-14, src/main.cpp:43-46,1399). */
typedef struct { float wc[4]; float nw; float pad[3]; } orc_weightinfo;

/* a1: shaders/bialteral.comp:29-82 (2-D texelFetch, OOB -> zero texel). */
void orc_bilateral_texture(const float *img, int w, int h, int radius,
                           float sigma_s, float sigma_c, float *out);
/* a2: shaders/bialteral_linear.comp:29-81 (flat index, rows wrap, idx outside [0,N) -> zero). */
void orc_bilateral_linear(const float *img, int w, int h, int radius,
                          float sigma_s, float sigma_c, float *out);
/* a3: shaders/bialteral_layers.comp:27-71; layer is RGBA8 (decoded as UNORM c/255). W += ... */
void orc_bilateral_layers_accum(const float *img, const uint8_t *layer_rgba8, int w, int h,
                                int radius, float sigma_s, float sigma_c, orc_weightinfo *W);
/* a4: shaders/nonlocal.comp:28-72.  Ranges are half-open [lo,hi): the reference is
 * search [-7,7), patch [-3,3); the 21x21/7x7 metric is search [-10,11), patch [-3,4). W += ... */
void orc_nlm_accum(const float *target, const float *neighbour, int w, int h, float hparam,
                   int search_lo, int search_hi, int patch_lo, int patch_hi, orc_weightinfo *W);
/* same arithmetic, rows spread over OpenMP threads (bench.py cpu_baseline only) */
void orc_nlm_accum_mt(const float *target, const float *neighbour, int w, int h, float hparam,
                      int search_lo, int search_hi, int patch_lo, int patch_hi, orc_weightinfo *W, int num_threads);
/* a5: shaders/normalize.comp:29-44. */
void orc_normalize(const orc_weightinfo *W, int w, int h, float *out);

/* a6: u8 paths. */
void orc_unpack_u8_unorm(const uint8_t *in, long n_values, float *out); /* c/255     src/texture.cpp:16 */
void orc_unpack_u8_cpu(const uint8_t *in, long n_values, float *out);   /* c*(1/255) src/main.cpp:1804-1807 */
void orc_pack_u8(const float *in, long n_values, uint8_t *out);         /* trunc     src/main.cpp:97-103 */

/* a7: src/main.cpp:1819-1865 (double-precision libm, float accumulators, blue-channel bug,
 * inclusive upper bounds; flat indices >= N read as zero pixels). */
void orc_cpu_bilateral(const float *in, int w, int h, int radius, float sigma_s, float sigma_c,
                       int blue_bug, int num_threads, float *out);

#ifdef __cplusplus
}
#endif
#endif
