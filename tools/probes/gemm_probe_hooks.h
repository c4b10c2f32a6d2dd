// Probe versions of the hooks merv_amd/csrc/gemm.hip leaves empty in the product build. Force-include this header into a diagnostic
// build of gemm.hip and choose what to switch on with -D flags:
//   tools/probes/build_ab.sh NAME -include tools/probes/gemm_probe_hooks.h -DMERV_ABL_NOSTORE        (a second library, MERV_HIP_LIB)
//   tools/probes/gemm_stamps.hip / gemm_energy_bound.hip                                              (stand-alone executables)
// Every ablation computes WRONG results by design; none of this is compiled into merv_amd/lib/libmerv_hip.so.
//   MERV_GEMM_STAMPS        per-wave s_memtime stamps (1..11), s_memrealtime + hardware ids at block entry / exit, into g_gemm_stamps
//   MERV_GEMM_STAMPS_LIGHT  only the entry / exit stamps (1, 11, real 0 / 12): the in-kernel clock with the kernel's overlaps intact
//   MERV_ABL_WRAPROWS       every output / residual row wraps into the first 4096 rows (8 MB at N = 1024: stays in the L2s)
//   MERV_ABL_WRAPOPS        operand rows wrap into g_probe_wrap_a / g_probe_wrap_w BYTES of A / W (runtime, 0 = off): the L2-resident
//                           operand footprint of the energy-bound probe (same MFMAs, same LDS traffic, fabric operand traffic -> ~0)
//   MERV_ABL_PLAINSTORE     L2-allocating output stores instead of the streaming (nontemporal) ones
//   MERV_ABL_ABLOCKED       the eight-phase kernel reads A in a K-tile-blocked layout (g_probe_a_blocked, runtime): contiguous 32 KB per (m-tile, K-tile)
//   MERV_ABL_A_NT=bits      cache-policy bits on the eight-phase kernel's A pieces (2 = nt, 1 = sc0, 16 = sc1)
//   MERV_ABL_NOSTORE        the whole epilogue, but nothing is stored (the condition is a runtime value: nothing is dead code)
//   MERV_ABL_HALFDMA        W pieces after K-tile 0 are never loaded (is the K-loop load-path-bound?)
//   MERV_ABL_NOEPI          prologue + K-loop + block turnover only
//   MERV_ABL_QUAD_ORDER=n   issue order of the eight-phase kernel's 16 MFMAs per phase (correct results: same sums)
//   MERV_ABL_REST=1|2       the remaining rows are not computed at all / an empty launch in their place
#pragma once
#define MERV_GEMM_PROBE_HOOKS 1

#if defined(MERV_GEMM_STAMPS) || defined(MERV_GEMM_STAMPS_LIGHT)
__device__ unsigned long long* g_gemm_stamps = nullptr;  // [block][8 waves][16]
#define MERV_GSTAMP_(k, INSTR)                                                                                       \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        unsigned long long t__;                                                                                      \
        asm volatile(INSTR " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (g_gemm_stamps && (threadIdx.x & 63) == 0)                                                                \
            g_gemm_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = t__;                         \
    } while (0)
#ifdef MERV_GEMM_STAMPS_LIGHT
// entry (1) / K-loop end (4) / exit (11) shader-clock stamps; with the real-time stamp taken at the same three points (0 / 13 / 12) the
// clock of the K-loop and of the epilogue can be told apart
#define MERV_GSTAMP(k) do { if constexpr ((k) == 1 || (k) == 11) MERV_GSTAMP_(k, "s_memtime"); if constexpr ((k) == 4) { MERV_GSTAMP_(4, "s_memtime"); MERV_GSTAMP_(13, "s_memrealtime"); } } while (0)
#define MERV_PROBE_DRAIN_STORES() do { } while (0)
#else
#define MERV_GSTAMP(k) MERV_GSTAMP_(k, "s_memtime")
#define MERV_PROBE_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#define MERV_GSTAMP_REAL(k) MERV_GSTAMP_(k, "s_memrealtime")
#define MERV_GSTAMP_HWID(k)                                                                                          \
    do {                                                                                                             \
        if (g_gemm_stamps && (threadIdx.x & 63) == 0)                                                                \
            g_gemm_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] =                               \
                ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32) |    \
                (unsigned)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));                        \
    } while (0)
#else
#define MERV_GSTAMP(k) do { } while (0)
#define MERV_GSTAMP_REAL(k) do { } while (0)
#define MERV_GSTAMP_HWID(k) do { } while (0)
#define MERV_PROBE_DRAIN_STORES() do { } while (0)
#endif

#ifdef MERV_ABL_WRAPROWS
#define MERV_PROBE_OUT_ROW(r) ((r) & 4095)
#else
#define MERV_PROBE_OUT_ROW(r) (r)
#endif

#ifdef MERV_ABL_WRAPOPS
__device__ int g_probe_wrap_a = 0, g_probe_wrap_w = 0;  // bytes of A / W the operand rows wrap into (0: no wrap)
// rows = bytes / (K elements x 2 B); at least one 8-row DMA piece
#define MERV_PROBE_WRAP_(r, p, bytes) ((bytes) > 0 ? (r) % (((bytes) / ((p).K * 2)) > 8 ? ((bytes) / ((p).K * 2)) : 8) : (r))
#define MERV_PROBE_A_ROW(r, p) MERV_PROBE_WRAP_(r, p, g_probe_wrap_a)
#define MERV_PROBE_W_ROW(r, p) MERV_PROBE_WRAP_(r, p, g_probe_wrap_w)
#else
#define MERV_PROBE_A_ROW(r, p) (r)
#define MERV_PROBE_W_ROW(r, p) (r)
#endif

#ifdef MERV_ABL_ABLOCKED  // A read as if stored [m-tile of 256 rows][K-tile][256 rows x 128 B] when g_probe_a_blocked != 0 (timing only: random data)
__device__ int g_probe_a_blocked = 0;
#define MERV_PROBE_A_OFFSET(r, p, es) (g_probe_a_blocked ? ((size_t)((r) >> 8) * ((p).K / 64) * 32768 + (size_t)((r) & 255) * 128) : (size_t)(r) * (p).lda * (es))
#define MERV_PROBE_A_KSTEP __builtin_amdgcn_readfirstlane(g_probe_a_blocked ? 32768 : 128)  // read once, ahead of the K-loop
#else
#define MERV_PROBE_A_OFFSET(r, p, es) ((size_t)(r) * (p).lda * (es))
#define MERV_PROBE_A_KSTEP 128
#endif

#ifdef MERV_ABL_A_NT  // the A pieces of the eight-phase kernel with the nt (streaming) cache policy: do they stop evicting the W panel from L2?
#define MERV_PROBE_A_DMA_AUX MERV_ABL_A_NT
#else
#define MERV_PROBE_A_DMA_AUX 0
#endif

#ifdef MERV_ABL_NOSTORE
#define MERV_PROBE_STORE_COND(p) ((p).group_m == 12345)
#else
#define MERV_PROBE_STORE_COND(p) true
#endif

#ifdef MERV_ABL_PLAINSTORE  // L2-allocating output stores instead of streaming ones
#define MERV_PROBE_STORE16(v, ptr) (*(ptr) = (v))
#else
#define MERV_PROBE_STORE16(v, ptr) __builtin_nontemporal_store(v, ptr)
#endif

#ifdef MERV_ABL_HALFDMA
#define MERV_PROBE_SKIP_W_DMA(t) ((t) > 0)
#else
#define MERV_PROBE_SKIP_W_DMA(t) false
#endif

#ifdef MERV_ABL_QUAD_ORDER  // issue order of the 16 MFMAs of a phase (product: 4; 0 = the order of rounds 2-5)
#define MERV_PROBE_QUAD_ORDER MERV_ABL_QUAD_ORDER
#else
#define MERV_PROBE_QUAD_ORDER 4
#endif

#ifdef MERV_ABL_NOEPI
#define MERV_PROBE_NO_EPILOGUE 1
#else
#define MERV_PROBE_NO_EPILOGUE 0
#endif

#ifdef MERV_ABL_REST
#define MERV_PROBE_REST_MODE MERV_ABL_REST
#include <hip/hip_runtime.h>
__global__ void merv_probe_noop_kernel() {}
static inline hipError_t merv_probe_rest_launch(hipStream_t s, hipError_t e) {
    if (MERV_ABL_REST == 1) return e;
    hipLaunchKernelGGL(merv_probe_noop_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}
#define MERV_PROBE_REST_LAUNCH(s, e) merv_probe_rest_launch(s, e)
#else
#define MERV_PROBE_REST_MODE 0
#define MERV_PROBE_REST_LAUNCH(s, e) (e)
#endif
