// Optional per-launch HIP-event timing of the library's kernels (used by bench.py's roofline leg).
// Disabled by default: zero cost. When a class is enabled every launch of that class is bracketed by two
// events recorded on the launch stream itself.
#pragma once
#include <hip/hip_runtime.h>

namespace merv {
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_TATTN = 2, PROF_LN = 3, PROF_OTHER = 4, PROF_NCLS = 5 };
extern int g_prof_mask;
void prof_begin_slow(int cls, hipStream_t s, double flops, double bytes);
void prof_end_slow(int cls, hipStream_t s);
struct ProfScope {
    int cls; hipStream_t s; bool on;
    ProfScope(int c, hipStream_t st, double flops, double bytes) : cls(c), s(st), on((g_prof_mask >> c) & 1) {
        if (on) prof_begin_slow(cls, s, flops, bytes);
    }
    ~ProfScope() { if (on) prof_end_slow(cls, s); }
};
}  // namespace merv
