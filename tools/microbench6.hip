// microbench6.hip -- LDS instruction cost by width on gfx950 (conflict-free, lane-consecutive addresses):
// how expensive are the narrow reads/writes an LDS transposition of the NLM box sums would need, compared
// with the ds_read_b128 the tile reads use?   hipcc --offload-arch=gfx950 -O3 tools/microbench6.hip -o /tmp/mb6
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed)
{
    __shared__ float sm[256 * 4 * 4 + 64];
    for (int i = threadIdx.x; i < 256 * 16; i += 256) sm[i] = seed + i;
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    // per-wave private region of 1024 floats
    float *base = sm + wv * 1024;
    unsigned a32 = (unsigned)(size_t)(base + lane);            // b32: dword per lane
    unsigned a64 = (unsigned)(size_t)(base + lane * 2);
    unsigned a128 = (unsigned)(size_t)(base + lane * 4);
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a32), "n"(0)); asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(r)); }
        }
        if (KIND == 1) {   // 8 x ds_read_b32, one wait
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a32), "n"(256 * 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[i];
        }
        if (KIND == 2) {   // 8 x ds_read2_b32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:65" : "=v"(r[i]) : "v"(a32));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
        }
        if (KIND == 3) {   // 8 x ds_write_b32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write_b32 %0, %1" :: "v"(a32), "v"(v0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (KIND == 4) {   // 8 x ds_write2_b32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:65" :: "v"(a32), "v"(v0), "v"(v1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (KIND == 5) {   // 8 x ds_read_b64
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(r[i]) : "v"(a64));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
        }
        if (KIND == 6) {   // 8 x ds_read_b128
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(a128));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].w;
        }
        if (KIND == 7) {   // 8 x ds_write_b128
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 v = {v0, v1, v2, v3};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"(a128), "v"(v) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (KIND == 8) {   // 8 x ds_write_b64
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v = {v0, v1};
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1" :: "v"(a64), "v"(v) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + sm[threadIdx.x];
}

template <int KIND>
int run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    float *d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = ms * 1e-3 * 2.4e9 / ((double)ITERS * 8 * 4 * waves_per_simd);   // clk per wave-level LDS instruction, per CU
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f clk per instruction per CU (@2.4GHz)\n", name, waves_per_simd, ms, per_cu);
    CK(hipFree(d));
    return 0;
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (run<1>("ds_read_b32", w) || run<2>("ds_read2_b32", w) || run<5>("ds_read_b64", w) || run<6>("ds_read_b128", w) ||
            run<3>("ds_write_b32", w) || run<4>("ds_write2_b32", w) || run<8>("ds_write_b64", w) || run<7>("ds_write_b128", w)) return 1;
    }
    return 0;
}
