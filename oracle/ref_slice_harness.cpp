// ref_slice_harness.cpp -- harness around the reference's own CPU bilateral loop.
//
// TEST INFRASTRUCTURE ONLY (checker + cpu_baseline "reference" leg of bench.py).
//
// The x-loop body (src/main.cpp:1827-1864: the `#pragma omp parallel for` line through the
// closing brace of the x loop) is NOT copied into this repo: oracle/Makefile extracts those
// lines from /root/reference at build time into a temporary file that this translation unit
// #includes via REF_SLICE, and only the resulting .so lands in oracle/_ref/ (git-ignored).
// What this file supplies is what surrounds that slice in RunOnCPU (src/main.cpp:1732-1866):
// the Pixel struct (:39-41), the input/output vectors, w, h, numThreads, windowSize (:1819,
// made a parameter) and the y loop (:1824).  No stand-in for any missing header or library
// is needed: the slice only calls <cmath>.
//
// The reference over-reads its input (row h, and column w wraps into the next row,
// SURVEY.md 8a-a7 "Bug 2").  Here the input is over-allocated with zero rows so that the
// over-read is defined; oracle.c's orc_cpu_bilateral states the same thing as "flat index
// >= N reads a zero pixel", and the two are compared on the whole frame.
#include <cmath>
#include <cstring>
#include <vector>

#ifndef REF_SLICE
#error "REF_SLICE must name the extracted slice (see oracle/Makefile)"
#endif

struct Pixel { float r, g, b, a; };

extern "C" void ref_cpu_bilateral(const float *in, int w, int h, int radius, int numThreads,
                                  float *out)
{
    const size_t n = (size_t)w * h;
    std::vector<Pixel> inputPixels(n + (size_t)w * (radius + 2));   // zero rows past the end
    std::memcpy((void *)inputPixels.data(), in, n * sizeof(Pixel));
    std::vector<Pixel> outputPixels(n);                             // Pixel{} = zeros (:1815)

    const int windowSize{radius};                                   // :1819 (10 in the reference)

    for (int y = windowSize; y <= h - windowSize; ++y)              // :1824
    {
#include REF_SLICE
    }

    std::memcpy(out, (const void *)outputPixels.data(), n * sizeof(Pixel));
}
