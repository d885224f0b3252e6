// Error plumbing of libgg (thread-local last-error string).
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/gg.h"
#include "prof.h"

static thread_local char g_err[1024] = "";

void gg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gg_last_error(void) { return g_err; }
extern "C" int gg_version(void) { return GG_VERSION; }

// ---------------------------------------------------------------------------------------------------------------
// Optional per-category kernel timing with HIP events recorded on the launch stream (bench.py roofline numbers).
// Disabled by default: zero events, zero overhead.
namespace {
struct Rec { int cat; hipEvent_t a, b; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
hipEvent_t get_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace
GgProfScope::GgProfScope(int cat, double flops, double bytes, void* stream) : idx_(-1), stream_(stream) {
    if (!g_on) return;
    Rec r{cat, get_event(), get_event(), flops, bytes};
    if (!r.a || !r.b) return;
    hipEventRecord(r.a, (hipStream_t)stream);
    g_recs.push_back(r);
    idx_ = (int)g_recs.size() - 1;
}
GgProfScope::~GgProfScope() {
    if (idx_ >= 0) hipEventRecord(g_recs[idx_].b, (hipStream_t)stream_);
}
bool gg_prof_is_on() { return g_on; }
extern "C" int gg_prof_enable(int on) { g_on = on != 0; return 0; }
extern "C" int gg_prof_reset(void) { g_recs.clear(); g_pool_next = 0; return 0; }
extern "C" int gg_prof_count(void) { return (int)g_recs.size(); }
extern "C" int gg_prof_record(int index, int* cat, double* ms, double* flops, double* bytes) {
    if (index < 0 || index >= (int)g_recs.size()) { gg_set_error("gg_prof_record: index %d out of range", index); return -1; }
    Rec& r = g_recs[index];
    float e = 0;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) { gg_set_error("gg_prof_record: event read failed"); return -2; }
    if (cat) *cat = r.cat;
    if (ms) *ms = e;
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    return 0;
}
extern "C" int gg_prof_read(int cat, double* ms, int64_t* launches, double* flops, double* bytes) {
    double t = 0, f = 0, b = 0; int64_t n = 0;
    for (auto& r : g_recs) {
        if (r.cat != cat) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { gg_set_error("gg_prof_read: event sync failed"); return -2; }
        float e = 0;
        if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) { gg_set_error("gg_prof_read: elapsed failed"); return -2; }
        t += e; f += r.flops; b += r.bytes; ++n;
    }
    if (ms) *ms = t;
    if (launches) *launches = n;
    if (flops) *flops = f;
    if (bytes) *bytes = b;
    return 0;
}
