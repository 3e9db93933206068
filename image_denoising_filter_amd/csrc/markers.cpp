// markers.cpp -- ROCTx ranges around the stages of the hot path, for `rocprofv3 --marker-trace`.
//
// The reference brackets every submit with timestamp queries (vkCmdWriteTimestamp before the copy, between copy and
// dispatch, after the dispatch: src/main.cpp:793-796,812-814,842-844) and prints the two differences.  The hipEvent
// timers (mid_timer_*, timings_ms) are this library's counterpart of those numbers; the ranges here are the counterpart
// for a TRACE: named host intervals -- "upload 17", "nlm 15", "download 14", one range per pipeline call, one per CLI mode
// -- that a profiler lines up with the kernel and copy records of the same run (tools/pipeline_trace.py).
//
// No link-time and no load-time dependency: the two entry points are looked up ONCE in what the process has ALREADY
// loaded (dlsym(RTLD_DEFAULT)): `rocprofv3 --marker-trace` preloads librocprofiler-sdk-roctx.so, an application that
// links libroctx64 itself has them too.  In any other process the lookup finds nothing, nothing is loaded, and a range
// costs one predictable branch.
#include "common.hpp"
#include <dlfcn.h>

namespace mid {

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        push = (int (*)(const char *))dlsym(RTLD_DEFAULT, "roctxRangePushA");
        pop = (int (*)())dlsym(RTLD_DEFAULT, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx &roctx() { static const Roctx r; return r; }
}  // namespace

bool markers_active() { return roctx().push != nullptr; }

void range_push(const char *name)
{
    if (roctx().push) (void)roctx().push(name ? name : "");
}

void range_pop()
{
    if (roctx().pop) (void)roctx().pop();
}

Range::Range(const char *fmt, ...) : on(roctx().push != nullptr)
{
    if (!on) return;
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (void)roctx().push(buf);
}

Range::~Range()
{
    if (on) (void)roctx().pop();
}

}  // namespace mid

// For callers that bracket their own stages (the CLI does, per reference mode): 1 when a ROCTx is present, else 0.
extern "C" int mid_range_push(const char *name) { mid::range_push(name); return mid::markers_active() ? 1 : 0; }
extern "C" int mid_range_pop(void) { mid::range_pop(); return mid::markers_active() ? 1 : 0; }
