#include "prof.h"

#include <stdint.h>

#include <vector>

extern "C" void merv_set_error(const char* msg);

namespace merv {
int g_prof_mask = 0;
namespace {
struct Rec { hipEvent_t a, b; int cls; double flops, bytes; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

int prof_begin_slow(int cls, hipStream_t s, double flops, double bytes) {
    Rec r{get_event(), get_event(), cls, flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_end_slow(int idx, hipStream_t s) {
    if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].b, s);
}
}  // namespace merv

using namespace merv;

extern "C" void merv_prof_enable(int32_t class_mask) { g_prof_mask = class_mask; }

extern "C" void merv_prof_reset(void) {
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
}

// Sums the recorded launches of one class. Blocks until their events have completed.
extern "C" int merv_prof_read(int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (auto& r : g_recs) {
        if (r.cls != cls) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { merv_set_error("merv_prof_read: event sync failed"); return 2; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { merv_set_error("merv_prof_read: elapsed failed"); return 2; }
        ms += t; fl += r.flops; by += r.bytes; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}
