// Optional per-launch HIP-event timing of the library's kernels (used by bench.py's roofline leg).
// Disabled by default: zero cost. When a class is enabled every launch of that class is bracketed by two
// events recorded on the launch stream itself.
#pragma once
#include <hip/hip_runtime.h>

namespace merv {
// Classes 0-3 bracket a library CALL (a GEMM call may be two kernel launches: eight-phase part + remaining rows); classes 5+
// bracket ONE kernel launch each and partition the step's kernels for bench.py's roofline.by_kernel (include/merv_hip.h lists
// them). Enable call-level and kernel-level classes in separate passes: nested brackets would time each other's event records.
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_TATTN = 2, PROF_LN = 3, PROF_OTHER = 4,
       PROF_K_GEMM8_PLAIN = 5,   // gemm_bf16_8phase_kernel, no activation (qkv, proj, fc2, temporal, projector)
       PROF_K_GEMM8_ACT = 6,     // gemm_bf16_8phase_kernel with an activation epilogue (fc1)
       PROF_K_GEMM_SMALL = 7,    // gemm_bf16_kernel: rows the eight-phase launch left over, patch embedding, small problems
       PROF_K_ATTN_RES = 8,      // attn_kernel with all K / V rows resident in LDS (257 / 261-token sequences)
       PROF_K_ATTN_STREAM = 9,   // attn_kernel streaming K / V tiles (196 and 3137 tokens)
       PROF_K_STATS = 10,        // row_stats_kernel + stats_finalize_kernel (LayerNorm statistics of the folded form)
       PROF_K_POOLFUSE = 11,     // pool_kernel + fusion_score_kernel + fusion_mix_kernel
       PROF_K_MOVE = 12,         // im2col, prefix rows, token gather: pure data movement
       PROF_NCLS = 13 };
extern int g_prof_mask;
int prof_begin_slow(int cls, hipStream_t s, double flops, double bytes);
void prof_end_slow(int idx, hipStream_t s);
struct ProfScope {
    hipStream_t s; int idx;
    ProfScope(int c, hipStream_t st, double flops, double bytes) : s(st), idx(((g_prof_mask >> c) & 1) ? prof_begin_slow(c, st, flops, bytes) : -1) {}
    ~ProfScope() { if (idx >= 0) prof_end_slow(idx, s); }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};
}  // namespace merv
