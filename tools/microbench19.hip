// microbench19.hip -- how many workgroups does a CU hold as a function of their LDS allocation?  (round 6)
//
// The plain bilateral at r = 8 asks for 40,960 B of LDS per 512-thread workgroup: four of them are exactly the CU's 160 KB, yet the kernel's
// measured occupancy (5.6 waves per SIMD) and its launch arithmetic (2.66 rounds per 1080p frame) say THREE are resident; the layer modes ask
// for 81,920 B = exactly half.  This probe measures instead of assuming: workgroups of NT threads with L bytes of dynamic LDS spin for a fixed
// number of shader cycles (s_memtime); a grid of 256 CUs x 24 workgroups then takes 24 / (resident workgroups per CU) spin periods.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mb19 tools/microbench19.hip && /tmp/mb19
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void spin(unsigned long long cycles, unsigned *sink)
{
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[0] = blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && lds[0] == 0xffffffffu) *sink = 1;
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned long long cyc = 200000;                    // 2 ms at 100 MHz (s_memtime counts the constant 100 MHz clock)
    const int per_cu = 24, grid = p.multiProcessorCount * per_cu;
    auto run = [&](int nt, size_t lds, float *ms) -> int {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(nt), lds, 0, cyc, sink);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(spin, dim3(grid), dim3(nt), lds, 0, cyc, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(ms, a, b));
        return 0;
    };
    float one = 0.f;
    {   // one period: a grid that certainly fits in one round
        hipLaunchKernelGGL(spin, dim3(p.multiProcessorCount), dim3(64), 1024, 0, cyc, sink);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(spin, dim3(p.multiProcessorCount), dim3(64), 1024, 0, cyc, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&one, a, b));
    }
    printf("one spin period = %.3f ms\n", one);
    for (int nt : {512, 256, 1024}) {
        printf("-- workgroups of %d threads, grid = %d CUs x %d\n", nt, p.multiProcessorCount, per_cu);
        for (size_t lds : {(size_t)16384, (size_t)27648, (size_t)32768, (size_t)36864, (size_t)39936, (size_t)40448, (size_t)40960, (size_t)41472, (size_t)49152, (size_t)53248,
                           (size_t)54784, (size_t)65536, (size_t)69888, (size_t)71680, (size_t)77824, (size_t)79872, (size_t)80896, (size_t)81920, (size_t)82944, (size_t)98304, (size_t)139776, (size_t)159744, (size_t)163840}) {
            float ms = 0.f;
            if (run(nt, lds, &ms)) return 1;
            const float rounds = ms / one;
            printf("   LDS %6zu B: %.2f periods -> %.2f workgroups per CU resident (LDS alone would allow %zu)\n", lds, rounds, per_cu / rounds, (size_t)(160 * 1024) / lds);
        }
    }
    return 0;
}
