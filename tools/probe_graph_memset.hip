// probe_graph_memset.hip -- development probe behind LABNOTES R6.10: does a hipMemsetAsync captured into a graph clear its buffer on every launch?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_graph_memset.hip -o /tmp/probe_graph_memset && /tmp/probe_graph_memset
// Sequence of tests/test_gpu_z_recording.py's first failure: capture [memset W], then eager work on the same stream (other memsets,
// kernels, copies), then launch the graph and read W back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float *p; size_t n; const void *table[192]; };      // a kernarg block the size of the NLM kernels' (frame tables by value)
__global__ void add1big(const Big b) { float *p = b.p; const size_t n = b.n; for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] += 1.0f; }
__global__ void add1(float *p, size_t n) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] += 1.0f; }
__global__ void fill(float *p, size_t n, float v) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v; }
static int check(const char *what, float *d, size_t n, float want = 0.0f)
{
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; float first_bad = 0;
    for (size_t i = 0; i < n; ++i) if (h[i] != want) { if (!bad) first_bad = h[i]; ++bad; }
    printf("%-58s %zu of %zu words wrong%s", what, bad, n, bad ? "" : "\n");
    if (bad) printf(" (first: %g = 0x%08x)\n", first_bad, *(unsigned *)&first_bad);
    return 0;
}
int main()
{
    const size_t n = 97 * 141 * 8, bytes = n * 4;
    float *W0, *other, *pool; CK(hipMalloc(&W0, bytes)); CK(hipMalloc(&other, 1 << 20)); CK(hipMalloc(&pool, 64 << 20));
    float *W = W0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 16; ++mode) {
        W = (mode & 8) ? pool + ((20u << 20) + 437760u * 3u + 512u) / 4 : W0;       // modes 8..15: the buffer is a piece of a larger allocation (a torch tensor is)
        if ((mode & 4) && !(mode & 2)) continue;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, (mode & 1) ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(W, 0, bytes, s));
        const int adds = (mode & 2) ? 5 : 0;                     // modes 2, 3: the memset followed by kernels that read-modify-write the buffer
        for (int i = 0; i < adds; ++i) {
            if (mode & 4) { Big b{}; b.p = W; b.n = n; for (int j = 0; j < 192; ++j) b.table[j] = (const void *)(other + j); hipLaunchKernelGGL(add1big, dim3(64), dim3(256), 0, s, b); }
            else hipLaunchKernelGGL(add1, dim3(64), dim3(256), 0, s, W, n);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("capture mode %s, %s, graph = memset%s\n", (mode & 8) ? "buffer inside a 64 MB allocation" : "buffer = its own allocation", (mode & 1) ? "global" : "thread-local",
               adds ? ((mode & 4) ? " + 5 kernels with 1.5 KB of arguments" : " + 5 kernels") : "");
        hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, s, W, n, 7.0f); CK(hipStreamSynchronize(s));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        if (check("  launch right after instantiate:", W, n, (float)adds)) return 1;
        // eager work in between, as the test has it
        CK(hipMemsetAsync(W, 0, bytes, s));
        CK(hipMemsetAsync(other, 0x5a, 1 << 20, s));
        hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, s, other, (size_t)(1 << 18), 3.0f);
        CK(hipStreamSynchronize(s));
        std::vector<char> junk(1 << 20, 0x33);
        CK(hipMemcpy(other, junk.data(), junk.size(), hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, s, W, n, 7.0f); CK(hipStreamSynchronize(s));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            if (check("  launch after eager memsets / copies on the stream:", W, n, (float)adds)) return 1;
        }
        // many eager launches with large argument blocks between two launches of the graph: is anything the memset node needs kept in a ring they recycle?
        if (mode == 6 || mode == 14) {
            for (int burst : {10, 100, 1000, 10000, 100000}) {
                Big b{}; b.p = other; b.n = 256; for (int j = 0; j < 192; ++j) b.table[j] = (const void *)(other + j);
                for (int i = 0; i < burst; ++i) { hipLaunchKernelGGL(add1big, dim3(1), dim3(256), 0, s, b); if ((i & 1023) == 1023) CK(hipStreamSynchronize(s)); }
                hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, s, W, n, 7.0f); CK(hipStreamSynchronize(s));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                char what[96]; snprintf(what, sizeof what, "  launch after %d eager launches with 1.5 KB of arguments:", burst);
                if (check(what, W, n, (float)adds)) return 1;
            }
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
