// bf16 MFMA GEMM with fused epilogue for the MERV encoder stack (gfx950).
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )         A: [M,K] bf16 row-major, W: [N,K] bf16 row-major
//
// W is stored exactly as a torch nn.Linear weight ([out_features, in_features]); this one kernel serves the
// patch/tubelet embedding (after im2col), QKV / attention-out projections, both MLP layers and the
// 3davg+linear projector -- every library GEMM the reference launches through timm / transformers
// (dinov2_video.py:138, siglip.py:148, vivit.py:104, modeling_video.py:98-186,
// nn_utils.py:25,330).
//
// Kernels in this file (DESIGN.md section 4 has the measurements):
//   gemm_bf16_8phase_kernel   256x256 tile, 8 waves (2x4, 128x64 each), two 64 KB LDS buffers, four phases per K-tile
//                             with ping-pong wave groups; the bulk of every launch. MX = true: the same schedule on
//                             MXFP8 operands (v_mfma_scale_f32_16x16x128_f8f6f4).
//   gemm_bf16_kernel          BM x BN tile with an NSTAGE-deep LDS ring and one counted-vmcnt barrier per K-step:
//                             256x128 / 3 stages / staggered half-blocks (remaining rows with many tiles, row-remapped
//                             patch embedding), 128x128 / 4 stages (at most one block per CU), 128x128 / 2 stages.
//   launch_gemm               plan: complete rounds of the chip -> eight-phase kernel, remaining rows -> smaller tile.
// Common to all: v_mfma_f32_16x16x32_bf16, operands staged global->LDS by LDS-DMA (global_load_lds_dwordx4), counted
// vmcnt + raw s_barrier (never a __syncthreads() while a DMA is outstanding). The LDS image is lane-linear (DMA
// constraint), so the bank swizzle chunk ^= (row & 7) is applied to the per-lane *source* address and to the ds_read
// address (cdna_hip_programming.md section 5.4 rule 21). Operands are swapped (D^T = W * A^T) so that every lane ends
// up with 4 consecutive n of one output row; the epilogue transposes through LDS for 16-byte, full-row accesses.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include <math.h>
#include <type_traits>

// Probe hooks. The product build compiles every one of them to nothing / the identity; the diagnostic builds under tools/probes
// (in-kernel stamps, ablations, operand-wrap energy probe) force-include tools/probes/gemm_probe_hooks.h, which defines
// MERV_GEMM_PROBE_HOOKS and its own versions.
#ifndef MERV_GEMM_PROBE_HOOKS
#define MERV_GSTAMP(k) do { } while (0)          // s_memtime stamp k of this wave
#define MERV_GSTAMP_REAL(k) do { } while (0)     // s_memrealtime stamp
#define MERV_GSTAMP_HWID(k) do { } while (0)     // XCC_ID / HW_ID
#define MERV_PROBE_OUT_ROW(r) (r)                // output / residual row an epilogue access goes to
#define MERV_PROBE_A_ROW(r, p) (r)               // A row a DMA piece reads
#define MERV_PROBE_W_ROW(r, p) (r)               // W row a DMA piece reads
#define MERV_PROBE_A_OFFSET(r, p, es) ((size_t)(r) * (p).lda * (es))  // byte offset of A row r at k = 0 (eight-phase kernel)
#define MERV_PROBE_A_KSTEP ROW_BYTES             // bytes between consecutive K-tiles of an A row
#define MERV_PROBE_A_DMA_AUX 0                   // cache-policy bits of the eight-phase kernel's A pieces (2 = nt)
#define MERV_PROBE_STORE_COND(p) true            // ANDed into the store predicate
#define MERV_PROBE_STORE16(v, ptr) __builtin_nontemporal_store(v, ptr)  // the epilogue's 16-byte output stores (streaming: no L2 allocation)
#define MERV_PROBE_SKIP_W_DMA(t) false           // eight-phase kernel: drop the W pieces of K-tile t
#define MERV_PROBE_NO_EPILOGUE 0                 // prologue + K-loop only
#define MERV_PROBE_QUAD_ORDER 4                  // issue order of the eight-phase kernel's 16 MFMAs per phase (quad_order)
#define MERV_PROBE_DRAIN_STORES() do { } while (0)  // stamped builds wait for their stores before the last stamp
#define MERV_PROBE_REST_MODE 0                   // launch_gemm: 0 remaining rows launched, 1 not computed, 2 an empty launch instead
#define MERV_PROBE_REST_LAUNCH(s, e) (e)
#endif

namespace merv {

namespace {

constexpr int BK = 64;                 // K elements per stage
constexpr int ROW_BYTES = BK * 2;      // 128 B per staged row

// One LDS-DMA wave-instruction moves 64 lanes x 16 B = 8 rows x 128 B.
// `rowblk` = index of that 8-row block inside the tile; lds_tile = tile base (wave-uniform).
MERV_DEVICE void dma_rows8(const bf16_t* __restrict__ g, int ld, int row0, int row_max, int rowblk, int kcol,
                           char* lds_tile, int lane) {
    const int r_in = lane >> 3;
    const int row = rowblk * 8 + r_in;
    const int chunk = (lane & 7) ^ (row & 7);  // source swizzle; LDS destination stays linear
    int grow = row0 + row;
    grow = grow < row_max ? grow : row_max;    // clamp: out-of-range rows re-read the last valid row
    const bf16_t* src = g + (size_t)grow * ld + kcol + chunk * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + rowblk * 1024), 16, 0, 0);
}

// Sum over each aligned group of 8 lanes, result in all 8: three DPP adds (quad_perm xor 1, xor 2, row_half_mirror), no LDS
// crossbar (__shfl_xor lowers to ds_bpermute here: ~100 cycles each, 6 per row chunk).
MERV_DEVICE float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // lanes ^1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // lanes ^2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // lane i <-> 7 - i
    return v;
}

// ---- epilogue modes (round 4) ----
// With all eight waves of a block in the epilogue the SIMDs are VALU-throughput-bound (two waves per SIMD: a v_pk_*_f32 or a
// v_cvt_pk_bf16_f32 retires every ~4.5-4.9 cycles per SIMD, tools/probes/valu_rate.hip), and the generic form spends four packed
// operations + one conversion per output pair whatever the launch asked for. The launcher therefore picks one of four forms:
//   EPI_GENERIC  every optional term from its runtime pointer (fold on {1, 0} / colsum 0 when absent, LayerScale 1): MXFP8 kernels and
//                the combinations the encoder stack never produces
//   EPI_PLAIN    bias only: the accumulators START at bias[n] (init_acc: the prologue waits for the first DMA anyway), so the
//                epilogue is activation + conversion -- and needs no operand loads (one L2 round trip less per tile)
//   EPI_LS       the same, then x LayerScale (DINOv2 proj / fc2)
//   EPI_FOLD     folded LayerNorm: fma(acc, rstd, fma(-mean rstd, colsum, bias)) -- two packed operations instead of three
// Every tile configuration uses the same form for the same launch, so a row gives the same bits whichever kernel computes it.
enum : int { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_LS = 2, EPI_FOLD = 3 };

// column of accumulator fragment i, register r, lane group fq, relative to the wave tile's first column
template <bool DIRECT>
MERV_DEVICE int frag_col(int i, int fq) { return DIRECT ? 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : 16 * i + 4 * fq; }

// EPI_PLAIN / EPI_LS: the accumulators start at the bias. The wave tile's 64 bias values are wave-uniform addresses, so they come
// by SCALAR loads (their own counter: nothing here touches the vmcnt queue the prologue's LDS-DMAs are counted on -- a vector load
// would make hipcc drain that queue at its first use, and an asm load's destination registers are copied by hipcc before the data
// lands) and each lane picks its lane group's four columns per fragment. wn0 must be provably wave-uniform at the call site.
template <int EPI, bool DIRECT, int MI>
MERV_DEVICE void acc_init(const GemmArgs& p, f32x4 (&acc)[4][MI], int lane, int wn0) {
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_LS) {
        if (p.bias) {
            // constant address space (read-only for the whole launch): s_load_dwordx16 whatever stores / DMAs precede
            typedef __attribute__((address_space(4))) const f32x16 cblock_t;
            const int fq = lane >> 4;
            auto pick = [&](float b0, float b1, float b2, float b3) { return fq == 0 ? b0 : fq == 1 ? b1 : fq == 2 ? b2 : b3; };
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // 32-column halves of the wave tile: two 16-float blocks each
                const f32x16 B0 = *(cblock_t*)(p.bias + wn0 + 32 * h), B1 = *(cblock_t*)(p.bias + wn0 + 32 * h + 16);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (DIRECT)  // fragment 2 h + e, lane group fq: column 32 h + 8 fq + 4 e + r
                            v[r] = pick(B0[4 * e + r], B0[8 + 4 * e + r], B1[4 * e + r], B1[8 + 4 * e + r]);
                        else                   // fragment 2 h + e: column 16 (2 h + e) + 4 fq + r
                            v[r] = e == 0 ? pick(B0[r], B0[4 + r], B0[8 + r], B0[12 + r]) : pick(B1[r], B1[4 + r], B1[8 + r], B1[12 + r]);
                    }
#pragma unroll
                    for (int j = 0; j < MI; ++j) acc[2 * h + e][j] = v;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// per-column operands of the epilogue for this lane (index i = accumulator fragment), by mode
template <int EPI, bool DIRECT>
struct EpiCols {
    float4 bias[4], ls[4], cs[4];
    MERV_DEVICE void load(const GemmArgs& p, int wn0, int fq) {
        if constexpr (EPI == EPI_GENERIC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wn0 + frag_col<DIRECT>(i, fq);
                bias[i] = p.bias ? *(const float4*)(p.bias + c) : float4{0.f, 0.f, 0.f, 0.f};
                ls[i] = p.lscale ? *(const float4*)(p.lscale + c) : float4{1.f, 1.f, 1.f, 1.f};
                cs[i] = p.row_stats ? *(const float4*)(p.ln_colsum + c) : float4{0.f, 0.f, 0.f, 0.f};
            }
        } else if constexpr (EPI == EPI_LS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ls[i] = *(const float4*)(p.lscale + wn0 + frag_col<DIRECT>(i, fq));
        } else if constexpr (EPI == EPI_FOLD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wn0 + frag_col<DIRECT>(i, fq);
                bias[i] = p.bias ? *(const float4*)(p.bias + c) : float4{0.f, 0.f, 0.f, 0.f};
                cs[i] = *(const float4*)(p.ln_colsum + c);
            }
        }
    }
};
// one accumulator quad -> its two finished pairs (fp32, before the bf16 conversion)
template <int EPI, int ACT, bool DIRECT>
MERV_DEVICE void epi_quad(const f32x4 v, const EpiCols<EPI, DIRECT>& c, int i, const float2 rs, f32x2& lo, f32x2& hi) {
    if constexpr (EPI == EPI_GENERIC) {
        // fma(acc, rstd, (-mean rstd) * colsum) + bias: the roundings of rounds 1-3 ({1, 0} / colsum 0 without a fold: acc exactly)
        const f32x2 rx = {rs.x, rs.x};
        lo = __builtin_elementwise_fma(f32x2{v[0], v[1]}, rx, f32x2{c.cs[i].x, c.cs[i].y} * rs.y) + f32x2{c.bias[i].x, c.bias[i].y};
        hi = __builtin_elementwise_fma(f32x2{v[2], v[3]}, rx, f32x2{c.cs[i].z, c.cs[i].w} * rs.y) + f32x2{c.bias[i].z, c.bias[i].w};
    } else if constexpr (EPI == EPI_FOLD) {
        const f32x2 rx = {rs.x, rs.x}, ry = {rs.y, rs.y};
        lo = __builtin_elementwise_fma(f32x2{v[0], v[1]}, rx,
                                       __builtin_elementwise_fma(f32x2{c.cs[i].x, c.cs[i].y}, ry, f32x2{c.bias[i].x, c.bias[i].y}));
        hi = __builtin_elementwise_fma(f32x2{v[2], v[3]}, rx,
                                       __builtin_elementwise_fma(f32x2{c.cs[i].z, c.cs[i].w}, ry, f32x2{c.bias[i].z, c.bias[i].w}));
    } else {  // the bias is already in the accumulator
        lo = f32x2{v[0], v[1]};
        hi = f32x2{v[2], v[3]};
    }
    activate4<ACT>(lo, hi);
    if constexpr (EPI == EPI_GENERIC || EPI == EPI_LS) {
        lo = lo * f32x2{c.ls[i].x, c.ls[i].y};
        hi = hi * f32x2{c.ls[i].z, c.ls[i].w};
    }
}
template <int EPI>
constexpr bool epi_needs_row_stats = (EPI == EPI_GENERIC || EPI == EPI_FOLD);

// ---- epilogue through LDS (shared by all tile configurations) ----
// ALLVALID (round 5, eight-phase launches whose M is a multiple of the tile): no row of the tile lies past M, so the output stores are
// unconditional -- hipcc's waitcnt pass counts an instruction behind a per-lane `if (valid)` as "maybe not issued" and waits for one more
// OLDER operation per such store, which made the later rows of a part wait for the part's own earlier stores -- and both parts' residual rows
// are requested up front (PIPE with two parts): part 1's rows then sit AHEAD of part 0's stores in the wave's in-order memory queue.
template <int WTM_FULL, int WTN, bool REMAP, int ACT, int EPI, int MSPLIT = 1, int WHOLE = 0>
MERV_DEVICE void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[WTN / 16][WTM_FULL / 16], char* smem, int wave, int lane, int m0,
                               int n0, int wr, int wc) {
    // MSPLIT > 1: the wave's rows are finished in MSPLIT passes of WTM rows each (bounds the registers the residual
    // rows and offsets take next to a 128-register accumulator)
    constexpr int WTM = WTM_FULL / MSPLIT;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    const int frow = lane & 15, fq = lane >> 4;
    // The accumulators hold D^T fragments (4 consecutive n per lane, 32-byte row segments per instruction): stored
    // straight to global memory that costs ~6 us per 256x128 tile (partial-line accesses), as much as 7 K-steps.
    // Instead each wave transposes its WTM x 64 sub-tile through its own LDS region (the stage ring is free now)
    // and reads it back row-contiguous: every global access of the epilogue is then 16 B per lane, 128 B per row.
    // Residual rows are loaded the same way BEFORE the transpose so their latency hides under it; C may alias the
    // residual (x += ...), which is safe because each element is read and written by the same lane.
    static_assert(WTN == 64, "epilogue staging assumes 64-column wave tiles");
    constexpr int EP_IT = WTM / 8;  // 16-byte chunks per lane: WTM rows x 8 chunks / 64 lanes
    const int wn0 = n0 + wc * WTN;
    // opaque copy of the lane id: keeps the compiler from hoisting the epilogue's index arithmetic (integer
    // divisions) above the K-loop, where it would sit in registers -- or scratch -- for the whole kernel
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int ec = elane & 7;  // this lane's 16-byte chunk (8 columns) of each row it handles
    // Per-column operands and (folded LayerNorm: v = acc * rstd + (-mean * rstd) * colsum[n]) per-row statistics are requested HERE,
    // ahead of the staging barrier and the residual loads, so their latency is hidden (loaded at first use they stalled every part
    // of every tile for a full global-load round trip: +3 % on qkv / fc1 launches). The row of acc[i][j] is j * 16 + frow of its part.
    EpiCols<EPI, false> cols;
    cols.load(p, wn0, elane >> 4);
    float2 rs_all[MSPLIT][MI];
#pragma unroll
    for (int part = 0; part < MSPLIT; ++part)
#pragma unroll
        for (int j = 0; j < MI; ++j) rs_all[part][j] = float2{1.f, 0.f};
    if constexpr (epi_needs_row_stats<EPI>) {
        if (EPI == EPI_FOLD || p.row_stats) {
            const int efrow = elane & 15;
#pragma unroll
            for (int part = 0; part < MSPLIT; ++part)
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    int m = m0 + wr * WTM_FULL + part * WTM + j * 16 + efrow;
                    m = m < p.M ? m : p.M - 1;
                    rs_all[part][j] = *(const float2*)(p.row_stats + 2 * (size_t)m);
                }
        }
    }
    char* stg = smem + wave * (WTM * 128);  // [WTM rows][128 B], 16-byte chunks XOR-swizzled by (row & 7)
    // MSPLIT >= 4 (A/B builds, -DMERV_GEMM_EPI_PARTS=4): residual rows are requested ONE PART AHEAD -- part p + 1's loads are issued
    // before part p is staged and stored, into the other half of a two-part register buffer (the registers of one part of twice the rows)
    // WHOLE: 0 = any tile, every optional term decided at run time. Otherwise a bit set, all of it compile time: bit 0 = every row of the tile is
    // valid, bit 1 = a residual, bit 2 = LayerNorm partials out, bit 3 = the row-indexed add, bits 4 / 5 = an MXFP8 output beside / instead of bf16 C. Behind a run-time
    // test a memory instruction is "maybe not issued" to hipcc's waitcnt pass, and the tests in every row's body kept it from batching the part's
    // read-backs: the static forms are straight-line code (stage, read back two ahead, store).
    constexpr bool ALLVALID = WHOLE != 0;
    constexpr bool PIPE = MSPLIT >= 4 || ((WHOLE & 2) && MSPLIT >= 2);
    const bool has_res = WHOLE ? (WHOLE & 2) != 0 : p.res != nullptr;
    // MXFP8 output: bit 4 = the result goes out as bf16 C AND as MXFP8 (the residual stream + its quantised copy for the next, LayerNorm-folded MX GEMM),
    // bit 5 = as MXFP8 only (fc1 -> fc2). Static forms with these bits exist for the MX kernels only (launch_8phase2); bf16 launches with an MXFP8
    // output take the run-time form.
    const bool mx_out = WHOLE ? (WHOLE & 48) != 0 : p.mx_out_q != nullptr;
    const bool store_c = WHOLE ? (WHOLE & 32) == 0 : (p.mx_out_q == nullptr || p.mx_out_keep_c != 0);
    const bool has_stats = WHOLE ? (WHOLE & 4) != 0 : p.stats_out != nullptr;
    const bool has_row_add = WHOLE ? (WHOLE & 8) != 0 : p.row_add != nullptr;
    u32x4 res_ahead[PIPE ? 2 : 1][EP_IT];
    auto res_rows = [&](int part, u32x4(&dst)[EP_IT]) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int m = m0 + wr * WTM_FULL + part * WTM + (elane >> 3) + 8 * it;
            int rr = m < p.M ? m : p.M - 1;
            if constexpr (REMAP) {
                if (p.res_row_mod > 0) rr = rr % p.res_row_mod;
            }
            dst[it] = has_res ? *(const u32x4*)(p.res + (size_t)((uint32_t)MERV_PROBE_OUT_ROW(rr) * (uint32_t)p.ldres + wn0 + ec * 8)) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    if constexpr (PIPE) {
        if (has_res) res_rows(0, res_ahead[0]);
    }
#pragma unroll
    for (int part = 0; part < MSPLIT; ++part) {
        if (part == 1) MERV_GSTAMP(8);  // part 0's stores are issued
        uint32_t c_off[EP_IT];  // element offsets (the launcher checks they fit 32 bits)
        bool valid[EP_IT];
        u32x4 resv[EP_IT];
        uint32_t r_off[EP_IT];
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int r = (elane >> 3) + 8 * it;
            const int m = m0 + wr * WTM_FULL + part * WTM + r;
            valid[it] = ALLVALID || m < p.M;
            const int mc = valid[it] ? m : p.M - 1;  // clamp instead of branching: loads stay unconditional
            int orow = mc, rr = mc;
            if constexpr (REMAP) {  // patch-embedding launch only: scatter past prefix tokens, position row m % P
                if (p.out_group > 0) orow = (mc / p.out_group) * p.out_stride + p.out_off + (mc % p.out_group);
                if (p.res_row_mod > 0) rr = mc % p.res_row_mod;
            }
            c_off[it] = (uint32_t)MERV_PROBE_OUT_ROW(orow) * (uint32_t)p.ldc + wn0 + ec * 8;
            r_off[it] = (uint32_t)MERV_PROBE_OUT_ROW(rr) * (uint32_t)p.ldres + wn0 + ec * 8;
        }
        // one wave-uniform branch around ALL residual loads (a per-element select would serialise them behind
        // vmcnt(0) waits: cdna_hip_programming.md, "Three .s-level traps" (c))
        if constexpr (PIPE) {
            if (has_res) {  // (uniform) this part's rows were requested a part ago; request the next part's
                if (part + 1 < MSPLIT) res_rows(part + 1, res_ahead[(part + 1) & 1]);
#pragma unroll
                for (int it = 0; it < EP_IT; ++it) resv[it] = res_ahead[part & 1][it];
            }
        } else if (has_res) {
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) resv[it] = *(const u32x4*)(p.res + (size_t)r_off[it]);
        } else {
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) resv[it] = u32x4{0u, 0u, 0u, 0u};
        }
        // row-indexed add (GemmArgs::row_add): the part's rows are consecutive and row_add_div >= their number, so they belong to at
        // most two groups; both candidate rows of the table are loaded once (this lane's 8 columns) and selected per output row
        float ra[2][8];
        int ra_boundary = 0x7fffffff;
        if (has_row_add) {  // uniform
            const int mb = __builtin_amdgcn_readfirstlane(m0 + wr * WTM_FULL + part * WTM) + p.row_add_row0;
            const int f0 = mb / p.row_add_div, i0 = f0 % p.row_add_mod, i1 = i0 + 1 == p.row_add_mod ? 0 : i0 + 1;
            ra_boundary = (f0 + 1) * p.row_add_div - p.row_add_row0;  // in this launch's row numbering
            const float* t0 = p.row_add + (size_t)i0 * p.N + wn0 + ec * 8;
            const float* t1 = p.row_add + (size_t)i1 * p.N + wn0 + ec * 8;
            const float4 a0 = *(const float4*)t0, a1 = *(const float4*)(t0 + 4), b0 = *(const float4*)t1, b1 = *(const float4*)(t1 + 4);
            ra[0][0] = a0.x; ra[0][1] = a0.y; ra[0][2] = a0.z; ra[0][3] = a0.w; ra[0][4] = a1.x; ra[0][5] = a1.y; ra[0][6] = a1.z; ra[0][7] = a1.w;
            ra[1][0] = b0.x; ra[1][1] = b0.y; ra[1][2] = b0.z; ra[1][3] = b0.w; ra[1][4] = b1.x; ra[1][5] = b1.y; ra[1][6] = b1.z; ra[1][7] = b1.w;
        }
        // part 0: every wave is done with the stage ring; later parts: this wave's reads of its staging region returned
        if (part == 0) MERV_GSTAMP(5);  // epilogue operands and part 0's residual rows requested
        if (part == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (part == 0) MERV_GSTAMP(6);  // staging barrier passed
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                f32x2 lo, hi;
                epi_quad<EPI, ACT, false>(acc[i][part * MI + j], cols, i, rs_all[part][j], lo, hi);
                u32x2 o;
                o[0] = pack2bf(lo[0], lo[1]);
                o[1] = pack2bf(hi[0], hi[1]);
                const int row = j * 16 + frow;
                const int chunk = (2 * i + (fq >> 1)) ^ (row & 7);
                *(u32x2*)(stg + row * 128 + chunk * 16 + 8 * (fq & 1)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: in-wave ordering is enough
        if (part == 0) MERV_GSTAMP(7);  // part 0 scaled, activated, packed and staged
        else MERV_GSTAMP(9);
        float2 row_part = float2{0.f, 0.f};
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int r = (elane >> 3) + 8 * it;
            u32x4 t = *(const u32x4*)(stg + r * 128 + ((ec ^ (r & 7)) * 16));
            if (has_res) {
                // bf16(linear) + bf16(residual), rounded once more: the reference's own order under autocast
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    t[q] = pack2bf(bflo(t[q]) + bflo(resv[it][q]), bfhi(t[q]) + bfhi(resv[it][q]));
            }
            if (has_row_add) {  // uniform: bf16(x + table row), x already rounded by the residual add
                const bool second = m0 + wr * WTM_FULL + part * WTM + r >= ra_boundary;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    t[q] = pack2bf(bflo(t[q]) + (second ? ra[1][2 * q] : ra[0][2 * q]), bfhi(t[q]) + (second ? ra[1][2 * q + 1] : ra[0][2 * q + 1]));
            }
            if (has_stats) {  // uniform: {sum, M2} of this row's 64 columns (the 8 lanes ec = 0..7 hold 8 values each)
                float f[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { f[2 * q] = bflo(t[q]); f[2 * q + 1] = bfhi(t[q]); }
                const float sm = sum8_dpp(((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7])));
                const float mu = sm * (1.f / 64.f);
                float m2 = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float d = f[q] - mu; m2 = fmaf(d, d, m2); }
                m2 = sum8_dpp(m2);
                if (ec == it) row_part = float2{sm, m2};  // lane 8 g + it keeps row g + 8 it of this part
            }
            if (mx_out) {  // uniform: the result goes out as MXFP8 (4 lanes = one 32-column block of the row)
                float r[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { r[2 * q] = bflo(t[q]); r[2 * q + 1] = bfhi(t[q]); }
                int sb;
                const u32x2 q8 = mx_quantize8(r, sb);
                const int m = m0 + wr * WTM_FULL + part * WTM + (elane >> 3) + 8 * it;
                const int col = wn0 + ec * 8;
                if (valid[it]) {
                    *(u32x2*)(p.mx_out_q + (size_t)m * p.N + col) = q8;
                    if ((ec & 3) == 0) p.mx_out_scales[mx_scale_offset(m, col >> 5, p.mx_out_groups)] = (uint8_t)sb;
                    if (store_c) MERV_PROBE_STORE16(t, (u32x4*)(p.C + (size_t)c_off[it]));  // (uniform) bf16 C beside its MXFP8 copy
                }
            } else if (valid[it] && MERV_PROBE_STORE_COND(p)) {
                // streaming store: the output (hundreds of MB per launch) is not re-read by this kernel, and written without
                // L2 allocation the round's write burst drains ~2 us sooner per tile (7.5 vs 9.4 us fixed cost, +1.2 % end to end)
                MERV_PROBE_STORE16(t, (u32x4*)(p.C + (size_t)c_off[it]));
            }
        }
        if (has_stats) {
            // lane l holds row (l >> 3) + 8 (l & 7): transpose the 8 x 8 lane grid so that lane L holds row L, and the part's
            // partials go out as ONE store of 8 * EP_IT consecutive float2 (layout [N / 64][M][2]: rows contiguous per column tile)
            const int src = 8 * (elane & 7) + (elane >> 3);
            const float sm = __shfl(row_part.x, src, 64), m2 = __shfl(row_part.y, src, 64);
            const int m = m0 + wr * WTM_FULL + part * WTM + elane;
            if (elane < 8 * EP_IT && m < p.M)
                *(float2*)(p.stats_out + 2 * ((size_t)(wn0 >> 6) * p.stats_ld + m)) = float2{sm, m2};
        }
    }
}

// ---- direct epilogue (round 4): no LDS transpose, no staging barrier ----
// W rows are PERMUTED at the DMA source (w_row_perm32 below): inside every 32-row group of the block's W tile, LDS row slot
// 16 e + 4 g + r holds W row 8 g + 4 e + r. The LDS image, the ds_read_b128 addresses and the bank pattern are what they were
// (the source swizzle uses the slot), but the D^T fragments now pair up: MFMA i = 2 nh + e of a wave leaves in lane (frow, g),
// register r, column 32 nh + 8 g + 4 e + r of row frow -- so acc[2 nh][j] and acc[2 nh + 1][j] together are 8 CONSECUTIVE
// columns = one 16-byte bf16 chunk, and the four lane groups g of a row cover a 64-byte row segment. Residual rows are loaded
// and results stored straight from / to registers, 16 B per lane, 16 rows x 64 B per instruction; nothing crosses LDS and the
// waves of a block no longer meet at a barrier between their last MFMA and their last store.
MERV_DEVICE int w_row_perm32(int slot) {  // slot = 16 e + 4 g + r  ->  W row 8 g + 4 e + r (within a 32-row group)
    return (slot & ~31) + 8 * ((slot >> 2) & 3) + 4 * ((slot >> 4) & 1) + (slot & 3);
}
// butterfly over the four 16-lane rows of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48): VALU only
MERV_DEVICE float sum_rows4(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
MERV_DEVICE float max_rows4(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int WTM_FULL, bool REMAP, int ACT, int EPI, int MSPLIT = 1, int WHOLE = 0>
MERV_DEVICE void gemm_epilogue_direct(const GemmArgs& p, f32x4 (&acc)[4][WTM_FULL / 16], int lane, int m0, int n0, int wr, int wc) {
    constexpr int WTM = WTM_FULL / MSPLIT;
    constexpr int MI = WTM / 16;
    // WHOLE: as gemm_epilogue (0 = run-time tests; else static: bit 0 whole tile and bf16 output, bit 1 residual, bit 2 LayerNorm partials out)
    const bool has_res = WHOLE ? (WHOLE & 2) != 0 : p.res != nullptr;
    const bool has_stats = WHOLE ? (WHOLE & 4) != 0 : p.stats_out != nullptr;
    const bool mx_out = WHOLE ? false : p.mx_out_q != nullptr;
    const bool store_c = WHOLE ? true : (p.mx_out_q == nullptr || p.mx_out_keep_c != 0);
    // opaque copy of the lane id: keeps the epilogue's index arithmetic below the K-loop (see gemm_epilogue)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int frow = elane & 15, fq = elane >> 4;
    const int wn0 = n0 + wc * 64;
    const int nl = wn0 + 8 * fq;  // this lane's chunk of the 32-column half nh: columns nl + 32 nh .. + 7
    // per-column operands, index i = 2 nh + e like the accumulators (columns nl + 32 nh + 4 e .. + 3), and the folded LayerNorm's rows
    EpiCols<EPI, true> cols;
    cols.load(p, wn0, fq);
    float2 rs_all[MSPLIT][MI];
#pragma unroll
    for (int part = 0; part < MSPLIT; ++part)
#pragma unroll
        for (int j = 0; j < MI; ++j) rs_all[part][j] = float2{1.f, 0.f};
    if constexpr (epi_needs_row_stats<EPI>) {
        if (EPI == EPI_FOLD || p.row_stats) {
#pragma unroll
            for (int part = 0; part < MSPLIT; ++part)
#pragma unroll
                for (int j = 0; j < MI; ++j) {
                    int m = m0 + wr * WTM_FULL + part * WTM + j * 16 + frow;
                    m = m < p.M ? m : p.M - 1;
                    rs_all[part][j] = *(const float2*)(p.row_stats + 2 * (size_t)m);
                }
        }
    }
#pragma unroll
    for (int part = 0; part < MSPLIT; ++part) {
        if (part == 1) MERV_GSTAMP(8);
        uint32_t c_off[MI], r_off[MI];  // element offsets of this lane's nh = 0 chunk (the launcher checks they fit 32 bits)
        bool valid[MI];
        u32x4 resv[MI][2];
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + wr * WTM_FULL + part * WTM + j * 16 + frow;
            valid[j] = WHOLE != 0 || m < p.M;
            const int mc = valid[j] ? m : p.M - 1;  // clamp instead of branching: loads stay unconditional
            int orow = mc, rr = mc;
            if constexpr (REMAP) {  // patch-embedding launch only: scatter past prefix tokens, position row m % P
                if (p.out_group > 0) orow = (mc / p.out_group) * p.out_stride + p.out_off + (mc % p.out_group);
                if (p.res_row_mod > 0) rr = mc % p.res_row_mod;
            }
            c_off[j] = (uint32_t)MERV_PROBE_OUT_ROW(orow) * (uint32_t)p.ldc + nl;
            r_off[j] = (uint32_t)MERV_PROBE_OUT_ROW(rr) * (uint32_t)p.ldres + nl;
        }
        // one wave-uniform branch around ALL residual loads
        if (has_res) {
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) resv[j][nh] = *(const u32x4*)(p.res + (size_t)r_off[j] + 32 * nh);
        } else {
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) resv[j][nh] = u32x4{0u, 0u, 0u, 0u};
        }
        if (part == 0) MERV_GSTAMP(5);
        if (part == 0) MERV_GSTAMP(6);
        static_assert(MI <= 4, "one lane group per 16-row fragment keeps its LayerNorm partials");
        float2 row_part = float2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            u32x4 t[2];
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * nh + e;
                    f32x2 lo, hi;
                    epi_quad<EPI, ACT, true>(acc[i][part * MI + j], cols, i, rs_all[part][j], lo, hi);
                    t[nh][2 * e] = pack2bf(lo[0], lo[1]);
                    t[nh][2 * e + 1] = pack2bf(hi[0], hi[1]);
                }
                if (has_res) {
                    // bf16(linear) + bf16(residual), rounded once more: the reference's own order under autocast
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        t[nh][q] = pack2bf(bflo(t[nh][q]) + bflo(resv[j][nh][q]), bfhi(t[nh][q]) + bfhi(resv[j][nh][q]));
                }
            }
            const int m = m0 + wr * WTM_FULL + part * WTM + j * 16 + frow;
            if (has_stats) {  // uniform: {sum, M2} of this row's 64 columns: 16 values here, the other 48 in lanes l ^ 16, ^ 32, ^ 48
                float f[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) { f[2 * q] = bflo(t[q >> 2][q & 3]); f[2 * q + 1] = bfhi(t[q >> 2][q & 3]); }
                const float s8a = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
                const float s8b = ((f[8] + f[9]) + (f[10] + f[11])) + ((f[12] + f[13]) + (f[14] + f[15]));
                const float sm = sum_rows4(s8a + s8b);
                const float mu = sm * (1.f / 64.f);
                float m2a = 0.f, m2b = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float da = f[q] - mu, db = f[8 + q] - mu;
                    m2a = fmaf(da, da, m2a);
                    m2b = fmaf(db, db, m2b);
                }
                const float m2 = sum_rows4(m2a + m2b);
                if (fq == j) row_part = float2{sm, m2};  // lane 16 j + frow keeps row 16 j + frow of this part
            }
            if (mx_out) {  // uniform: the result goes out as MXFP8 (the four lanes of a row = one 32-column block per nh)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float r[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { r[2 * q] = bflo(t[nh][q]); r[2 * q + 1] = bfhi(t[nh][q]); }
                    float amax = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) amax = fmaxf(amax, fabsf(r[q]));
                    amax = max_rows4(amax);
                    const int ex = mx_shared_exponent(amax);
                    const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
                    const u32x2 q8 = u32x2{mx_pack4(r[0], r[1], r[2], r[3], inv), mx_pack4(r[4], r[5], r[6], r[7], inv)};
                    const int col = nl + 32 * nh;
                    if (valid[j]) {
                        *(u32x2*)(p.mx_out_q + (size_t)m * p.N + col) = q8;
                        if (fq == 0) p.mx_out_scales[mx_scale_offset(m, col >> 5, p.mx_out_groups)] = (uint8_t)(ex + 127);
                    }
                }
                if (store_c && valid[j]) {  // (uniform) bf16 C beside its MXFP8 copy
                    MERV_PROBE_STORE16(t[0], (u32x4*)(p.C + (size_t)c_off[j]));
                    MERV_PROBE_STORE16(t[1], (u32x4*)(p.C + (size_t)c_off[j] + 32));
                }
            } else if (valid[j] && MERV_PROBE_STORE_COND(p)) {
                // streaming stores (no L2 allocation: see gemm_epilogue)
                MERV_PROBE_STORE16(t[0], (u32x4*)(p.C + (size_t)c_off[j]));
                MERV_PROBE_STORE16(t[1], (u32x4*)(p.C + (size_t)c_off[j] + 32));
            }
        }
        if (has_stats) {  // one store of 16 MI consecutive float2 per part (layout [N / 64][M][2])
            const int m = m0 + wr * WTM_FULL + part * WTM + elane;
            if (elane < 16 * MI && m < p.M)
                *(float2*)(p.stats_out + 2 * ((size_t)(wn0 >> 6) * p.stats_ld + m)) = row_part;
        }
        if (part == 0) MERV_GSTAMP(7);
        else MERV_GSTAMP(9);
    }
}

// Which epilogue a launch takes (tools/gemm_ksweep.py, tools/gemm_bench.py, same-box pairs, EXPERIMENTS section 1): the direct form wins
// where the epilogue is VALU-heavy (activation launches: fc1 -3 %), ties with a residual, and loses 1.5 us per tile on plain
// launches (half-line streaming stores). A/B builds: -DMERV_GEMM_EPILOGUE=0 (always through LDS) / 1 (always direct).
#ifndef MERV_GEMM_EPILOGUE
#define MERV_GEMM_EPILOGUE 2
#endif
// Parts the eight-phase kernel's LDS epilogue finishes a wave's 128 rows in: 2 x 64 rows, each part requesting its own residual rows
// (rounds 2-5); -DMERV_GEMM_EPI_PARTS=4: 4 x 32 rows with the residual rows requested one part ahead (round-5 A/B form, EXPERIMENTS.md section 1).
#ifndef MERV_GEMM_EPI_PARTS
#define MERV_GEMM_EPI_PARTS 2
#endif
// the static (whole-tile) forms: 4 x 32 rows, a residual's rows one part ahead -- with exact waits the finer pipeline wins (GEMM time -0.9 %), where
// the run-time form lost 1 us per tile to vmcnt(0)s (round 5, first half)
#ifndef MERV_GEMM_EPI_PARTS_WHOLE
#define MERV_GEMM_EPI_PARTS_WHOLE 4
#endif
#ifndef MERV_GEMM_EPI_PARTS_DIRECT_WHOLE
#define MERV_GEMM_EPI_PARTS_DIRECT_WHOLE 2  // (the register epilogue of the activation launches)
#endif
template <int ACT>
constexpr bool gemm_direct_epilogue = (MERV_GEMM_EPILOGUE == 1) || (MERV_GEMM_EPILOGUE == 2 && ACT != ACT_NONE);

template <int N>
MERV_DEVICE void wait_dma_barrier() {
    // my DMAs except the youngest N have landed and my LDS reads have returned; after the barrier that holds for
    // every wave of the block
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// BM x BN block tile, WAVES_M x WAVES_N waves, NSTAGE LDS stages (dynamic shared memory), ONE barrier per K-step:
//   wait(my DMA of tile kt) ; barrier  => tile kt is complete in LDS AND every wave has finished reading tile kt-1
//   issue DMA of tile kt+NSTAGE-1 into the stage tile kt-1 occupied, one 1-KiB piece after each row of MFMAs
//   (a DMA piece costs ~60-180 issue cycles: spread out, the SIMD's other wave fills the gap with its MFMAs)
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool STAGGER, bool REMAP, int ACT, int EPI, int WHOLE = 0>  // WHOLE: static epilogue form (gemm_epilogue)
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_kernel(GemmArgs p) {
    constexpr bool DIRECT = gemm_direct_epilogue<ACT>;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // per-wave output tile
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_PIECES = BM / 8 / NW, W_PIECES = BN / 8 / NW;
    constexpr int DPS = A_PIECES + W_PIECES;  // LDS-DMA instructions per wave per stage
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over waves");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    // ---- tile selection: XCD-contiguous ids, then grouped (GM m-tiles per column sweep) ordering ----
    const int tilesM = (p.M + BM - 1) / BM, tilesN = p.N / BN;
    const int nwg = tilesM * tilesN;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int GM = p.group_m > 0 ? p.group_m : ((BM >= 256) ? 4 : 8);
    const int per_group = GM * tilesN;
    const int grp = id / per_group;
    const int first_m = grp * GM;
    const int gsz = (tilesM - first_m) < GM ? (tilesM - first_m) : GM;
    const int in_grp = id - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // Per-lane DMA source pointers of this thread's pieces at kt = 0 (row clamp and source swizzle folded in once):
    // a K-step then costs one 64-bit add per piece instead of the whole address computation.
    const bf16_t* a_src[A_PIECES];
    const bf16_t* w_src[W_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        int grow = m0 + row;
        grow = grow < p.M - 1 ? grow : p.M - 1;  // out-of-range rows re-read the last valid row
        a_src[i] = p.A + (size_t)grow * p.lda + (((lane & 7) ^ (row & 7)) * 8);
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);  // LDS row slot (the swizzle follows the slot)
        const int wrow = DIRECT ? w_row_perm32(row) : row;  // W row it receives (gemm_epilogue_direct)
        w_src[i] = p.W + (size_t)(n0 + wrow) * p.ldw + (((lane & 7) ^ (row & 7)) * 8);
    }
    // piece `pc` (0..DPS-1) of tile kt into stage `buf`
    auto dma_piece = [&](int kt, int buf, int pc) {
        char* a_tile = smem + buf * STAGE_BYTES;
        const bf16_t* src = pc < A_PIECES ? a_src[pc < A_PIECES ? pc : 0] : w_src[pc >= A_PIECES ? pc - A_PIECES : 0];
        char* dst = pc < A_PIECES ? a_tile + (pc * NW + wave) * 1024 : a_tile + A_BYTES + ((pc - A_PIECES) * NW + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kt * BK),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    static_assert(NI == 4, "64-column wave tiles");
    f32x4 acc[NI][MI];

    const int nkt = p.K / BK;
    // per-lane fragment addressing: row = l & 15 inside each 16-row fragment, 16-byte chunk (l >> 4) + 4*kk
    const int frow = lane & 15;
    const int fq = lane >> 4;
    const int sw = lane & 7;  // == row & 7 for every fragment row this lane reads (fragment bases are multiples of 16)

#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nkt) {
#pragma unroll
            for (int pc = 0; pc < DPS; ++pc) dma_piece(s0, s0, pc);
        }
    // the accumulators start at the bias (EPI_PLAIN / EPI_LS) while the first tiles are in flight
    acc_init<EPI, DIRECT>(p, acc, lane, n0 + wc * WTN);

    // fragments of one K=32 half of stage `buf`
    auto read_half = [&](int buf, int kk, bf16x8(&af)[MI], bf16x8(&wf)[NI]) {
        const char* a_tile = smem + buf * STAGE_BYTES + (wr * WTM + frow) * ROW_BYTES;
        const char* w_tile = smem + buf * STAGE_BYTES + A_BYTES + (wc * WTN + frow) * ROW_BYTES;
        const int coff = (((kk * 4) + fq) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < MI; ++j) af[j] = *(const bf16x8*)(a_tile + j * 16 * ROW_BYTES + coff);
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8*)(w_tile + i * 16 * ROW_BYTES + coff);
    };
    // MFMAs of one half; `share` (0/1) selects which half of the prefetched tile's DMA pieces is interleaved
    // (PREFETCH is a compile-time flag so the steady-state body is one straight-line scheduling region)
    auto mma_half = [&](const bf16x8(&af)[MI], const bf16x8(&wf)[NI], int share, int kt_next, int buf_next, auto tag) {
        constexpr bool PREFETCH = decltype(tag)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < MI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            if constexpr (PREFETCH) {
                constexpr int ROWS = 2 * NI;
                const int r = share * NI + i;
#pragma unroll
                for (int pc = 0; pc < DPS; ++pc)
                    if (pc * ROWS / DPS == r) dma_piece(kt_next, buf_next, pc);
            }
        }
    };
    auto next_buf = [&](int buf) {
        int b2 = buf + NSTAGE - 1;
        return b2 >= NSTAGE ? b2 - NSTAGE : b2;
    };

    // With two waves per SIMD (8-wave blocks) the block's two halves run half a K-step apart (STAGGER): waves 0..3
    // do [read, MFMA, read, MFMA] after each barrier, waves 4..7 do [MFMA (operands read before the barrier), read,
    // MFMA, read], so on every SIMD one wave's LDS reads / DMA issue sit under its partner's MFMAs instead of both
    // waves stalling and computing in lockstep (MI355X_MICROARCH.md "Two waves per SIMD", item 9).
    const bool late_half = STAGGER && wave >= NW / 2;
    int buf = 0;
    int kt = 0;
    if (!late_half) {
        auto k_step = [&](int kt_, int buf_, auto tag) {
            bf16x8 af[MI], wf[NI];
            read_half(buf_, 0, af, wf);
            mma_half(af, wf, 0, kt_ + NSTAGE - 1, next_buf(buf_), tag);
            read_half(buf_, 1, af, wf);
            mma_half(af, wf, 1, kt_ + NSTAGE - 1, next_buf(buf_), tag);
        };
        // steady state: NSTAGE-1 tiles in flight, the oldest of them is needed now
        for (; kt + NSTAGE - 1 < nkt; ++kt) {
            wait_dma_barrier<(NSTAGE - 2) * DPS>();
            k_step(kt, buf, std::true_type{});
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
        // drain: no more prefetches; `left` tiles (this one included) are still to be multiplied
        for (; kt < nkt; ++kt) {
            const int left = nkt - kt;  // 1 .. NSTAGE-1
            if (left >= 3) wait_dma_barrier<2 * DPS>();
            else if (left == 2) wait_dma_barrier<DPS>();
            else wait_dma_barrier<0>();
            k_step(kt, buf, std::false_type{});
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
    } else {
        bf16x8 haf[MI], hwf[NI];  // second-half operands, held across the barrier
        auto k_step = [&](int kt_, int buf_, auto tag) {
            constexpr bool PREFETCH = decltype(tag)::value;
            __builtin_amdgcn_sched_barrier(0);  // keep the deferred MFMAs on this side of the barrier
            if (kt_ > 0) {
                mma_half(haf, hwf, 0, kt_ + NSTAGE - 1, next_buf(buf_), tag);
            } else if constexpr (PREFETCH) {
#pragma unroll
                for (int pc = 0; pc < DPS; ++pc)
                    if (pc * 2 / DPS == 0) dma_piece(kt_ + NSTAGE - 1, next_buf(buf_), pc);
            }
            bf16x8 af[MI], wf[NI];
            read_half(buf_, 0, af, wf);
            mma_half(af, wf, 1, kt_ + NSTAGE - 1, next_buf(buf_), tag);
            read_half(buf_, 1, haf, hwf);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (; kt + NSTAGE - 1 < nkt; ++kt) {
            wait_dma_barrier<(NSTAGE - 2) * DPS>();
            k_step(kt, buf, std::true_type{});
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
        for (; kt < nkt; ++kt) {
            const int left = nkt - kt;
            if (left >= 3) wait_dma_barrier<2 * DPS>();
            else if (left == 2) wait_dma_barrier<DPS>();
            else wait_dma_barrier<0>();
            k_step(kt, buf, std::false_type{});
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        }
        mma_half(haf, hwf, 0, 0, 0, std::false_type{});
    }

    if constexpr (MERV_PROBE_NO_EPILOGUE) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (DIRECT) {
        static_assert(WTN == 64, "the direct epilogue pairs the four 16-column fragments of a 64-column wave tile");
        gemm_epilogue_direct<WTM, REMAP, ACT, EPI, 1, WHOLE>(p, acc, lane, m0, n0, wr, wc);
    } else {
        gemm_epilogue<WTM, WTN, REMAP, ACT, EPI, 1, WHOLE>(p, acc, smem, wave, lane, m0, n0, wr, wc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 256 tile, eight phases per two K-tiles (cdna_hip_programming.md section 5, "The 256^2 8-phase template").
// 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave, two 64 KB LDS buffers. A K-tile is four phases; each phase
// loads one register sub-tile from LDS, issues one quarter of a later K-tile's LDS-DMA (2 pieces per wave), and
// runs the 16 MFMAs of one 64 x 32 output quadrant between two barriers:
//
//   phase 1: read A rows [0,64) + B cols [0,32) of the wave tile   DMA B-late(t+1)   MFMA quadrant (0,0)
//   phase 2: read B cols [32,64)                                    DMA A-late(t+1)   MFMA quadrant (0,1)
//   phase 3: read A rows [64,128)                                   DMA A-early(t+2)  MFMA quadrant (1,1)
//   phase 4: -                             DMA B-early(t+2), wait tile t+1 landed     MFMA quadrant (1,0)
//
// "early" / "late" quarters of a tile are the row sets read in phase 1 / in phases 2-3 (for every wave at once:
// A-early = tile rows {0-63, 128-191}, B-early = tile columns {64c .. 64c+31}), so a quarter of the CURRENT buffer is
// free to be overwritten by tile t+2 two phases after its only read, and each wave has at most 8 DMAs outstanding.
// Waves 4-7 run one barrier behind waves 0-3 (an extra s_barrier up front, balanced at the end): on every SIMD one
// wave is in its MFMA section (raised priority) while its partner is in its load section.
// Ordering: the only vmcnt wait is in phase 4 (leaves the 4 youngest DMAs = two quarters of tile t+2 in flight) and
// sits before that phase's first barrier; tile t+1 is first read one phase later, after two more barriers, by which
// time the trailing wave group has executed the same wait (rule "read a staged buffer one phase AFTER the wait that
// retires it", one barrier more because of the stagger).
// ---------------------------------------------------------------------------------------------------------
// Issue order of the 16 MFMAs of one phase of the eight-phase kernel: position x -> (k-half kk) << 3 | (W fragment ii) << 2 | (A fragment jj).
// Every order gives the same bits (an accumulator takes k-half 0 before k-half 1); what differs is which operand registers an MFMA shares
// with its predecessors, and on random operands that is ENERGY: a bare MFMA loop over this wave tile (tools/probes/mfma_hold.hip, no
// memory traffic, chip at its power cap) runs 2006 TFLOP/s at 1.99 GHz when consecutive MFMAs share nothing, 2069 in order 0, 2114 at
// 2.09 GHz in order 4 -- and 2456-2465 at 2.39 GHz in every order on all-zero operands (profiles/r05_mfma_issue_order.json). In the
// kernel (bench.py, three alternating same-box passes per library): GEMM time of the step 96.4-96.7 ms in order 0, 95.7-95.9 in order 4.
//   0  k-half, W fragment, A fragment: a W fragment (the MFMA's first operand) serves four MFMAs in a row (rounds 2-5)
//   1  W fragment, A fragment, k-half: the two k-halves of an accumulator back to back (-3.5 %: dependent issue)
//   2  k-half, A fragment, W fragment: an A fragment (second operand) serves two MFMAs in a row (-0.6 % GEMM time)
//   3  A fragment, W fragment, k-half (-1.6 %... slower: dependent pairs again)
//   4  k-half, A fragment, W fragment in snake order: 2 x 2 blocks, every MFMA shares one operand with its predecessor (PRODUCT, -0.8 %)
//   5  k-half, W fragment, A fragment in snake order (-0.4 %)
//   6  A-fragment pair, k-half, A fragment, W fragment (an accumulator returns after 4; -0.7 %)
//   7  k-half, A fragment descending, W fragment (-0.6 %)
constexpr int quad_order(int ord, int x) {
    int kk = 0, ii = 0, jj = 0;
    switch (ord) {
        case 1: ii = x >> 3; jj = (x >> 1) & 3; kk = x & 1; break;
        case 2: kk = x >> 3; jj = (x >> 1) & 3; ii = x & 1; break;
        case 3: jj = x >> 2; ii = (x >> 1) & 1; kk = x & 1; break;
        case 4: kk = x >> 3; jj = (x >> 1) & 3; ii = (x & 1) ^ (jj & 1); break;
        case 5: kk = x >> 3; ii = (x >> 2) & 1; jj = ii ? 3 - (x & 3) : (x & 3); break;
        case 6: jj = 2 * (x >> 3) + ((x >> 1) & 1); kk = (x >> 2) & 1; ii = x & 1; break;
        case 7: kk = x >> 3; jj = 3 - ((x >> 1) & 3); ii = x & 1; break;
        default: kk = x >> 3; ii = (x >> 2) & 1; jj = x & 3; break;
    }
    return kk << 3 | ii << 2 | jj;
}

template <bool REMAP, int ACT, bool MX, int EPI, int WHOLE = 0>
__global__ __launch_bounds__(512) void gemm_bf16_8phase_kernel(GemmArgs p) {
    // MX = true: the same schedule on MXFP8 operands (OCP e4m3 elements, one E8M0 scale per 32 elements of K,
    // v_mfma_scale_f32_16x16x128_f8f6f4: twice the bf16 MFMA rate and half the operand bytes). A K-tile is still 128
    // bytes per row (128 fp8 elements instead of 64 bf16), so the LDS image, the DMA pieces and the two ds_read_b128
    // per fragment are byte-identical: the instruction's operand is [16 bytes of k = 16g..] ++ [16 bytes of k = 64+16g..]
    // for lane group g (probed with exact integer data, tools/probes/mx_layout_probe.hip), i.e. chunks g and 4+g of the
    // row. What is added: one 256-byte scale DMA per wave per K-tile and three ds_read_b32 per wave per K-tile.
    constexpr int BM = 256, BN = 256, WTM = 128, WTN = 64, MI = 8, NI = 4;
    constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = (BM + BN) * ROW_BYTES;  // 32 KB, 64 KB
    constexpr int ES = MX ? 1 : 2;                // bytes per operand element
    constexpr int BKE = ROW_BYTES / ES;           // elements of K per K-tile
    constexpr int SC_BASE = 2 * BUF_BYTES;        // MX: 2 x 2 KB of block scales above the two operand buffers
    constexpr bool DIRECT = gemm_direct_epilogue<ACT> && !MX;  // MX: the W block scales are laid out by (unpermuted) row fragment
    static_assert(MX || (WHOLE & 48) == 0, "static forms with an MXFP8 output exist for the MX kernels only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MERV_GSTAMP_REAL(0);
    MERV_GSTAMP_HWID(14);
    MERV_GSTAMP(1);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int tilesM = (p.M + BM - 1) / BM, tilesN = p.N / BN;
    const int nwg = tilesM * tilesN;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int GM = p.group_m > 0 ? p.group_m : 4;  // 8 is -4 % on the isolated qkv shape (tools/probes/gemm_group_sweep.py) but -1.3 % tokens/s end to end
    const int per_group = GM * tilesN;
    const int grp = id / per_group;
    const int first_m = grp * GM;
    const int gsz = (tilesM - first_m) < GM ? (tilesM - first_m) : GM;
    const int in_grp = id - grp * per_group;
    const int m0 = (first_m + in_grp % gsz) * BM, n0 = (in_grp / gsz) * BN;

    // DMA sources; [s] = early / late quarter, [u] = this wave's two 8-row pieces of the quarter. SADDR form (round 5): the tile's first
    // row at k = 0 is a wave-uniform pointer (SGPRs, advanced by scalar adds) and each piece a 32-bit per-lane byte offset computed
    // once (row inside the tile, clamped; source-side swizzle), so the instruction is global_load_lds_dwordx4 v_off, s[base:base+1] and
    // the K-loop carries NO per-piece 64-bit vector address arithmetic (16 v_lshl_add_u64 per wave and K-tile in the pointer form,
    // issued right in front of the pieces they feed). The probe builds keep the per-lane pointer form (their hooks rewrite rows).
#ifndef MERV_GEMM_SADDR
#ifdef MERV_GEMM_PROBE_HOOKS
#define MERV_GEMM_SADDR 0
#else
#define MERV_GEMM_SADDR 1
#endif
#endif
    constexpr bool SADDR = MERV_GEMM_SADDR != 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* a_src[2][2];
    const char* b_src[2][2];
    uint32_t a_off[2][2], b_off[2][2];
    const char* const a_tile = (const char*)p.A + (size_t)m0 * p.lda * ES;  // wave-uniform
    const char* const b_tile = (const char*)p.W + (size_t)n0 * p.ldw * ES;
    const int r8 = lane >> 3, sw8 = ((lane & 7) ^ r8) * 16;  // source-side swizzle, in bytes
#pragma unroll
    for (int sq = 0; sq < 2; ++sq)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int grow = m0 + u * 128 + sq * 64 + wave * 8 + r8;
            grow = grow < p.M - 1 ? grow : p.M - 1;  // rows past M re-read the last valid row (never stored)
            // LDS row slot ((wave >> 2) + 2 u) * 64 + sq * 32 + (wave & 3) * 8 + r8; DIRECT: it receives the permuted W row
            const int s32 = (wave & 3) * 8 + r8;
            const int nrel = ((wave >> 2) + 2 * u) * 64 + sq * 32 + (DIRECT ? w_row_perm32(s32) : s32);
            if constexpr (SADDR) {
                a_off[sq][u] = (uint32_t)(grow - m0) * (uint32_t)(p.lda * ES) + sw8;  // < 256 rows x row pitch: 32 bits (the launcher checks)
                b_off[sq][u] = (uint32_t)nrel * (uint32_t)(p.ldw * ES) + sw8;
            } else {
                grow = MERV_PROBE_A_ROW(grow, p);
                a_src[sq][u] = (const char*)p.A + MERV_PROBE_A_OFFSET(grow, p, ES) + sw8;
                const int nrow = MERV_PROBE_W_ROW(n0 + nrel, p);
                b_src[sq][u] = (const char*)p.W + (size_t)nrow * p.ldw * ES + sw8;
            }
        }
    auto dma = [&](const char* src, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    auto dma_act = [&](const char* src, int lds_off) {  // the A (activation) pieces: cache-policy bits from the probe hook (product: 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, MERV_PROBE_A_DMA_AUX);
    };
    // the SADDR form goes through inline asm (hipcc selects the vector-address form for the builtin whatever the pointer is made of):
    // M0 = the piece's LDS byte address, written inside the statement and listed in its clobbers (so hipcc never keeps a value of its own
    // live in M0 across a piece: movrel / readlane indexing, a builtin LDS-DMA of a later edit). The explicit vmcnt waits of the K-loop
    // are the only ordering these pieces have ever had (the compiler never waits for a DMA it issued through the builtin either);
    // the epilogue starts behind a vmcnt(0).
    auto dma_saddr = [&](const char* sbase, uint32_t voff, int lds_off) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base + lds_off) : "memory", "m0");
    };
    const int a_kstep = MERV_PROBE_A_KSTEP;  // (product: the constant ROW_BYTES)
    auto dma_a = [&](int sq, int t, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int dst = buf * BUF_BYTES + (u * 16 + sq * 8 + wave) * 1024;
            if constexpr (SADDR) dma_saddr(a_tile + t * ROW_BYTES, a_off[sq][u], dst);
            else dma_act(a_src[sq][u] + t * a_kstep, dst);
        }
    };
    auto dma_b = [&](int sq, int t, int buf) {
        if (MERV_PROBE_SKIP_W_DMA(t)) return;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int dst = buf * BUF_BYTES + A_BYTES + ((((wave >> 2) + 2 * u) * 8 + sq * 4 + (wave & 3))) * 1024;
            if constexpr (SADDR) dma_saddr(b_tile + t * ROW_BYTES, b_off[sq][u], dst);
            else dma(b_src[sq][u] + t * ROW_BYTES, dst);
        }
    };
    // MX block scales, layout [K-tile][64-row group][lane = 16 * kblock + row % 16][(row % 64) / 16] bytes: waves 0-3
    // bring the A groups of this tile's 256 rows, waves 4-7 the W groups (256 B each, 4 B per lane)
    const char* s_base = nullptr;  // wave-uniform: this wave's 64-row group of K-tile 0
    size_t s_stride = 0;           // bytes between K-tiles
    if constexpr (MX) {
        const bool is_w = wave >= 4;
        const int gtot = is_w ? p.mx_groups_w : p.mx_groups_a;
        int gidx = (is_w ? n0 : m0) / 64 + (wave & 3) + (is_w ? 0 : p.mx_group0_a);
        gidx = gidx < gtot ? gidx : gtot - 1;
        s_base = (const char*)(is_w ? p.mx_scale_w : p.mx_scale_a) + (size_t)gidx * 256;
        s_stride = (size_t)gtot * 256;
    }
    auto dma_s = [&](int t, int buf) {
        if constexpr (MX) {
            if constexpr (SADDR)  // (no builtin DMA beside the asm ones: hipcc's own M0 writes must not meet M0 writes it cannot see)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"((uint32_t)(lane * 4)), "s"(s_base + t * s_stride),
                             "s"(lds_base + SC_BASE + buf * 2048 + wave * 256) : "memory", "m0");
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_base + t * s_stride + (size_t)(uint32_t)(lane * 4)),
                                                 (__attribute__((address_space(3))) void*)(smem + SC_BASE + buf * 2048 + wave * 256), 4, 0, 0);
        }
    };

    f32x4 acc[NI][MI];

    const int nkt = p.K / BKE;
    const int frow = lane & 15, fq = lane >> 4, sw = lane & 7;
    const int a_rd = (wr * WTM + frow) * ROW_BYTES, b_rd = A_BYTES + (wc * WTN + frow) * ROW_BYTES;
    const int coff0 = ((0 + fq) ^ sw) * 16, coff1 = ((4 + fq) ^ sw) * 16;

    // prologue: all of tile 0 and the early quarters of tile 1 are requested (the launcher guarantees nkt >= 4), but the first
    // phase only needs tile 0's EARLY quarters (32 of its 64 KB): the wait releases those, and tile 0's late quarters are waited
    // for inside its own first two phases (FIRST mode of k_tile) -- every CU of a round starts its tile at once, so the prologue
    // is bandwidth-bound and half the bytes arrive in about half the time
    dma_s(0, 0);
    dma_a(0, 0, 0); dma_b(0, 0, 0); dma_b(1, 0, 0); dma_a(1, 0, 0);
    dma_a(0, 1, 1); dma_b(0, 1, 1);
    acc_init<EPI, DIRECT>(p, acc, lane, n0 + wc * WTN);  // EPI_PLAIN / EPI_LS: the accumulators start at the bias (scalar loads)
    MERV_GSTAMP(2);  // prologue DMAs issued
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    MERV_GSTAMP(3);  // tile 0's early quarters have landed: first phase released
    if (wr == 1) asm volatile("s_barrier" ::: "memory");  // trailing group: one barrier behind from here on

    typedef int v8i_t __attribute__((ext_vector_type(8)));
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    bf16x8 af[4][2];      // bf16: current 64-row half of the wave's A rows, both K halves
    bf16x8 wf[2][2][2];   // bf16: both 32-column halves of the wave's W rows
    v8i_t af8[4];         // MX: the same fragments as 32-byte operands (chunks g and 4+g of the row)
    v8i_t wf8[2][2];
    int sc_a[2] = {0, 0}, sc_w = 0;  // MX: scale dwords (one byte per row fragment) of the current K-tile
    auto load32 = [&](const char* base) {
        const v4i_t l = *(const v4i_t*)(base + coff0), h = *(const v4i_t*)(base + coff1);
        return v8i_t{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
    };
    auto read_a = [&](int buf, int mh) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const char* base = smem + buf * BUF_BYTES + a_rd + (mh * 4 + jj) * 16 * ROW_BYTES;
            if constexpr (MX) {
                af8[jj] = load32(base);
            } else {
                af[jj][0] = *(const bf16x8*)(base + coff0);
                af[jj][1] = *(const bf16x8*)(base + coff1);
            }
        }
    };
    auto read_w = [&](int buf, int nh) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const char* base = smem + buf * BUF_BYTES + b_rd + (nh * 2 + ii) * 16 * ROW_BYTES;
            if constexpr (MX) {
                wf8[nh][ii] = load32(base);
            } else {
                wf[nh][ii][0] = *(const bf16x8*)(base + coff0);
                wf[nh][ii][1] = *(const bf16x8*)(base + coff1);
            }
        }
    };
    auto read_scales = [&](int buf) {
        if constexpr (MX) {
            const char* sb = smem + SC_BASE + buf * 2048 + lane * 4;
            sc_a[0] = *(const int*)(sb + (2 * wr) * 256);
            sc_a[1] = *(const int*)(sb + (2 * wr + 1) * 256);
            sc_w = *(const int*)(sb + 1024 + wc * 256);
        }
    };
    // The scaled MFMA goes through inline asm with the accumulator as a tied "+v" operand: the compiler's own selection of
    // this instruction never accumulates in place (destination != source C: it doubled the accumulator registers and
    // copied them every iteration). The scale byte is an instruction field: byte b of the dword = op_sel bit (b & 1),
    // op_sel_hi bit (b >> 1), first slot for the first operand. Hazards the compiler cannot see inside asm are avoided
    // by construction: operands come straight from ds_read -- waited for by the counted lgkmcnt waits hipcc puts in front of the asm statements
    // that name them as inputs (there is no blanket lgkmcnt(0) behind a phase's barrier since round 5, MERV_GEMM_PH_LGKM); the block scales
    // sc_a / sc_w are read in phase 1 BEFORE that phase's W fragments and LDS returns in order, so a wait that releases a W fragment has
    // released the scales as well (an edit that moves read_scales behind the fragment reads would break this) --, each accumulator is
    // touched once per K-tile, and the epilogue is separated from the last MFMA by barriers plus explicit s_nops.
#define MERV_MX_ASM(ACC, WOP, AOP, SW, SA, OL0, OL1, OH0, OH1)                                                          \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[" #OL0 "," #OL1 ",0] op_sel_hi:[" #OH0 \
                 "," #OH1 ",0]"                                                                                          \
                 : "+v"(ACC)                                                                                             \
                 : "v"(WOP), "v"(AOP), "v"(SW), "v"(SA))
    auto mx_mma = [&](f32x4& c, const v8i_t& w8, const v8i_t& a8, int sw_, int sa_, auto wb_tag, auto ab_tag) {
        constexpr int wb = decltype(wb_tag)::value, ab = decltype(ab_tag)::value;
        if constexpr (wb == 0 && ab == 0) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 0, 0, 0);
        if constexpr (wb == 0 && ab == 1) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 1, 0, 0);
        if constexpr (wb == 0 && ab == 2) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 0, 0, 1);
        if constexpr (wb == 0 && ab == 3) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 1, 0, 1);
        if constexpr (wb == 1 && ab == 0) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 0, 0, 0);
        if constexpr (wb == 1 && ab == 1) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 1, 0, 0);
        if constexpr (wb == 1 && ab == 2) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 0, 0, 1);
        if constexpr (wb == 1 && ab == 3) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 1, 0, 1);
        if constexpr (wb == 2 && ab == 0) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 0, 1, 0);
        if constexpr (wb == 2 && ab == 1) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 1, 1, 0);
        if constexpr (wb == 2 && ab == 2) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 0, 1, 1);
        if constexpr (wb == 2 && ab == 3) MERV_MX_ASM(c, w8, a8, sw_, sa_, 0, 1, 1, 1);
        if constexpr (wb == 3 && ab == 0) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 0, 1, 0);
        if constexpr (wb == 3 && ab == 1) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 1, 1, 0);
        if constexpr (wb == 3 && ab == 2) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 0, 1, 1);
        if constexpr (wb == 3 && ab == 3) MERV_MX_ASM(c, w8, a8, sw_, sa_, 1, 1, 1, 1);
    };
#define MERV_MX_MMA(I, J, WI, AJ, SA) \
    mx_mma(acc[I][J], wf8[(WI) >> 1][(WI) & 1], af8[AJ], sc_w, SA, std::integral_constant<int, (WI)>{}, std::integral_constant<int, (AJ)>{})
    auto quadrant = [&](auto mh_tag, auto nh_tag) {
        constexpr int mh = decltype(mh_tag)::value, nh = decltype(nh_tag)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MX) {
            const int sa = sc_a[mh];
            MERV_MX_MMA(nh * 2 + 0, mh * 4 + 0, nh * 2 + 0, 0, sa);
            MERV_MX_MMA(nh * 2 + 0, mh * 4 + 1, nh * 2 + 0, 1, sa);
            MERV_MX_MMA(nh * 2 + 0, mh * 4 + 2, nh * 2 + 0, 2, sa);
            MERV_MX_MMA(nh * 2 + 0, mh * 4 + 3, nh * 2 + 0, 3, sa);
            MERV_MX_MMA(nh * 2 + 1, mh * 4 + 0, nh * 2 + 1, 0, sa);
            MERV_MX_MMA(nh * 2 + 1, mh * 4 + 1, nh * 2 + 1, 1, sa);
            MERV_MX_MMA(nh * 2 + 1, mh * 4 + 2, nh * 2 + 1, 2, sa);
            MERV_MX_MMA(nh * 2 + 1, mh * 4 + 3, nh * 2 + 1, 3, sa);
        } else {
            // issue order of the quadrant's 16 MFMAs (an accumulator always takes k-half 0 before k-half 1: same sums, same bits);
            // the orders and what they measured: quad_order() above the kernel
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                constexpr int ORD = MERV_PROBE_QUAD_ORDER;
                const int c = quad_order(ORD, x), kk = c >> 3, ii = (c >> 2) & 1, jj = c & 3;
                acc[nh * 2 + ii][mh * 4 + jj] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nh][ii][kk], af[jj][kk], acc[nh * 2 + ii][mh * 4 + jj], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
#ifndef MERV_GEMM_PH_LGKM
#define MERV_GEMM_PH_LGKM 0  // 1: an explicit lgkmcnt(0) behind the phase's first barrier (rounds 2-5). Without it hipcc's own counted waits release the phase's
                             // MFMAs as their fragments land (lgkmcnt(7), (6), (3) ...): every fragment a phase reads is an operand of that phase's own MFMAs, so
                             // all of its reads have returned before the wave reaches the phase's closing barrier, which is what the buffer recycling needs
                             // (GEMM time -0.3 %, three same-box pairs)
#endif
#define MERV_PH_LOADED()                                                   \
    do {                                                                    \
        if constexpr (MERV_GEMM_PH_LGKM != 0) asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    \
        else asm volatile("s_barrier" ::: "memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#define MERV_PH_DONE()                                  \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        asm volatile("s_barrier" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
    // one K-tile; MODE 0: tiles t+1 and t+2 exist, 1: only t+1 (second-last tile), 2: last tile -- compile-time, so every
    // body is straight-line code (the launcher only takes this kernel for an even number of K-tiles >= 4)
    // FIRST (tile 0 only, MODE 0): tile 0's late quarters were not waited for in the prologue. Its B-late quarter is read in
    // phase 2 and its A-late quarter in phase 3, so phases 1 and 2 each retire one of them before their first barrier (the
    // wave's queue then holds, youngest last: [the quarter wanted][the other late quarter or nothing][tile 1: A-early, B-early,
    // (MX scales), B-late, (A-late)] -> all but the 8 (MX: 9) youngest), one phase ahead of the read as the ordering rule asks.
    using I2 = std::integral_constant<int, 2>;
    auto k_tile = [&](int t, auto buf_tag, auto mode_tag, auto first_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool has1 = MODE <= 1, has2 = MODE == 0;
        static_assert(!FIRST || MODE == 0, "the first tile always has two successors");
        // phase 1
        read_scales(BUF);
        read_w(BUF, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(BUF, 0);
        if constexpr (has1) { dma_s(t + 1, BUF ^ 1); dma_b(1, t + 1, BUF ^ 1); }
        if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MX ? 9 : 8) : "memory");
        MERV_PH_LOADED();
        quadrant(I0{}, I0{});
        MERV_PH_DONE();
        // phase 2
        read_w(BUF, 1);
        if constexpr (has1) dma_a(1, t + 1, BUF ^ 1);
        if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MX ? 9 : 8) : "memory");
        MERV_PH_LOADED();
        quadrant(I0{}, I1{});
        MERV_PH_DONE();
        // phase 3
        read_a(BUF, 1);
        if constexpr (has2) dma_a(0, t + 2, BUF);
        MERV_PH_LOADED();
        quadrant(I1{}, I1{});
        MERV_PH_DONE();
        // phase 4
        if constexpr (has2) {
            dma_b(0, t + 2, BUF);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if constexpr (has1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        MERV_PH_LOADED();
        quadrant(I1{}, I0{});
        MERV_PH_DONE();
    };
    k_tile(0, I0{}, I0{}, std::true_type{});
    k_tile(1, I1{}, I0{}, std::false_type{});
    int t = 2;
    for (; t + 2 < nkt; t += 2) {
        k_tile(t, I0{}, I0{}, std::false_type{});
        k_tile(t + 1, I1{}, I0{}, std::false_type{});
    }
    k_tile(t, I0{}, I1{}, std::false_type{});
    k_tile(t + 1, I1{}, I2{}, std::false_type{});
#undef MERV_PH_LOADED
#undef MERV_PH_DONE
#undef MERV_MX_MMA
#undef MERV_MX_ASM
    if constexpr (MX) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // MFMA results vs. the epilogue's reads
    if (wr == 0) asm volatile("s_barrier" ::: "memory");  // balance the trailing group's extra barrier
    MERV_GSTAMP(4);  // K-loop done

    if constexpr (MERV_PROBE_NO_EPILOGUE) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (DIRECT) {
        gemm_epilogue_direct<WTM, REMAP, ACT, EPI, (WHOLE ? MERV_GEMM_EPI_PARTS_DIRECT_WHOLE : 2), WHOLE>(p, acc, lane, m0, n0, wr, wc);
    } else {
        gemm_epilogue<WTM, WTN, REMAP, ACT, EPI, (WHOLE ? MERV_GEMM_EPI_PARTS_WHOLE : MERV_GEMM_EPI_PARTS), WHOLE>(p, acc, smem, wave, lane, m0, n0, wr, wc);
    }
    MERV_GSTAMP(10);  // part 1's stores are issued
    MERV_PROBE_DRAIN_STORES();
    MERV_GSTAMP(11);  // every store acknowledged
    MERV_GSTAMP_REAL(12);
}

// Per-device state: hipFuncSetAttribute and the CU count belong to a device, and one process may drive several
// (load_vid(device="cuda:1"), one path per GPU). The C ABI makes the stream's device current before any launch.
constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}
inline hipError_t ensure_dynamic_lds(const void* kern, int lds_bytes, bool (&done)[MAX_DEVICES]) {
    const int dev = current_device();
    if (done[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

#ifndef MERV_GEMM_ALLVALID
#define MERV_GEMM_ALLVALID 1
#endif
template <bool REMAP, int ACT, int EPI, bool MX = false, int WHOLE = 0>
hipError_t launch_8phase2(const GemmArgs& a, hipStream_t s) {
    // whole tiles through the LDS epilogue (see gemm_epilogue, WHOLE): bias-only / LayerScale launches with the residual known at compile time
    // Whole tiles (M a multiple of 256, bf16 output): the static epilogue forms the encoder stacks produce (gemm_epilogue, WHOLE) -- every optional
    // term a template flag
    if constexpr (MERV_GEMM_ALLVALID && WHOLE == 0 && !REMAP && !MX && (EPI == EPI_PLAIN || EPI == EPI_LS || EPI == EPI_FOLD)) {
        if (a.M % 256 == 0 && !a.mx_out_q) {
            const int f = 1 | (a.res ? 2 : 0) | (a.stats_out ? 4 : 0) | (a.row_add ? 8 : 0);
            if (f == 1) return launch_8phase2<REMAP, ACT, EPI, MX, 1>(a, s);           // qkv, temporal qkv, the projector, fc1
            if constexpr (EPI != EPI_FOLD && ACT == ACT_NONE) {
                if (f == 3) return launch_8phase2<REMAP, ACT, EPI, MX, 3>(a, s);       // + residual
                if (f == 7) return launch_8phase2<REMAP, ACT, EPI, MX, 7>(a, s);       // + residual + LayerNorm partials (proj, fc2)
                if constexpr (EPI == EPI_PLAIN && !gemm_direct_epilogue<ACT>) {
                    if (f == 15) return launch_8phase2<REMAP, ACT, EPI, MX, 15>(a, s);  // + the next block's temporal embedding (LanguageBind fc2)
                }
            }
        }
    }
    // MXFP8 launches (round 6): the same static forms where the encoder stacks produce them in MX mode with the LayerNorm fold -- qkv / temporal qkv
    // (folded, bf16 out), fc1 (folded, activation, MXFP8 out only), out-projections / fc2 (residual + LayerNorm partials, optionally the MXFP8 copy of the
    // stream and LanguageBind's row-indexed add). Anything else takes the run-time form OF THE SAME EPILOGUE MODE (the modes round differently --
    // EPI_PLAIN starts the accumulators at the bias, EPI_FOLD nests its two fmas --, and a launch's rows must not depend on which form computed them:
    // the ragged rows behind the whole tiles are a second launch of the run-time form; mx_static_form below names the static set).
    if constexpr (MERV_GEMM_ALLVALID && WHOLE == 0 && !REMAP && MX && (EPI == EPI_PLAIN || EPI == EPI_LS || EPI == EPI_FOLD)) {
        if (a.M % 256 == 0 && !a.no_static_form) {
            const int f = 1 | (a.res ? 2 : 0) | (a.stats_out ? 4 : 0) | (a.row_add ? 8 : 0) | (a.mx_out_q ? (a.mx_out_keep_c ? 16 : 32) : 0);
            if constexpr (EPI == EPI_FOLD && ACT == ACT_NONE) {
                if (f == 1) return launch_8phase2<REMAP, ACT, EPI, MX, 1>(a, s);
            } else if constexpr (EPI == EPI_FOLD) {
                if (f == 33) return launch_8phase2<REMAP, ACT, EPI, MX, 33>(a, s);
            } else if constexpr (ACT == ACT_NONE) {
                if (f == 7) return launch_8phase2<REMAP, ACT, EPI, MX, 7>(a, s);
                if (f == 23) return launch_8phase2<REMAP, ACT, EPI, MX, 23>(a, s);
                if constexpr (EPI == EPI_PLAIN) {
                    if (f == 15) return launch_8phase2<REMAP, ACT, EPI, MX, 15>(a, s);
                    if (f == 31) return launch_8phase2<REMAP, ACT, EPI, MX, 31>(a, s);
                }
            }
        }
    }
    constexpr int LDS = 2 * (256 + 256) * ROW_BYTES + (MX ? 4096 : 0);  // 128 KB (+ 4 KB of MX block scales)
    auto kern = gemm_bf16_8phase_kernel<REMAP, ACT, MX, EPI, WHOLE>;
    static bool attr_set[MAX_DEVICES] = {};  // per instantiation, per device
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_set); e != hipSuccess) return e;
    const int tilesM = (a.M + 255) / 256, tilesN = a.N / 256;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(512), LDS, s, a);
    return hipGetLastError();
}
// The epilogue mode of a launch (see EPI_*). Row remapping (patch embedding) and activations never come with a LayerScale in the
// encoder stack: those combinations exist only in generic form.
inline int epi_mode(const GemmArgs& a) {
    if (a.row_stats) return a.lscale ? EPI_GENERIC : EPI_FOLD;
    if (a.lscale) return (a.act == ACT_NONE && a.out_group <= 0 && a.res_row_mod <= 0) ? EPI_LS : EPI_GENERIC;
    return EPI_PLAIN;
}
template <int ACT>
hipError_t launch_8phase(const GemmArgs& a, hipStream_t s) {
    const int epi = epi_mode(a);
    if (a.out_group > 0 || a.res_row_mod > 0) {
        if constexpr (ACT == ACT_NONE) {
            if (epi == EPI_PLAIN) return launch_8phase2<true, ACT, EPI_PLAIN>(a, s);
            if (epi == EPI_GENERIC) return launch_8phase2<true, ACT, EPI_GENERIC>(a, s);
        }
        return hipErrorInvalidValue;
    }
    switch (epi) {
        case EPI_PLAIN: return launch_8phase2<false, ACT, EPI_PLAIN>(a, s);
        case EPI_FOLD: return launch_8phase2<false, ACT, EPI_FOLD>(a, s);
        case EPI_LS:
            if constexpr (ACT == ACT_NONE) return launch_8phase2<false, ACT, EPI_LS>(a, s);
            return hipErrorInvalidValue;
        default: return launch_8phase2<false, ACT, EPI_GENERIC>(a, s);
    }
}
// which MXFP8 launches have a static epilogue form (launch_8phase2): a whole number of 256-row tiles and one of the encoder stacks' combinations
inline bool mx_static_form(const GemmArgs& a) {
    if (!MERV_GEMM_ALLVALID || a.M % 256 != 0 || a.no_static_form) return false;
    const int epi = epi_mode(a);
    const int f = 1 | (a.res ? 2 : 0) | (a.stats_out ? 4 : 0) | (a.row_add ? 8 : 0) | (a.mx_out_q ? (a.mx_out_keep_c ? 16 : 32) : 0);
    if (epi == EPI_FOLD) return a.act == ACT_NONE ? f == 1 : f == 33;
    if (a.act != ACT_NONE) return false;
    if (epi == EPI_PLAIN) return f == 7 || f == 23 || f == 15 || f == 31;
    if (epi == EPI_LS) return f == 7 || f == 23;
    return false;
}
template <int ACT>
hipError_t launch_8phase_mx(const GemmArgs& a, hipStream_t s) {  // the epilogue mode by the launch's arguments, as launch_8phase
    switch (epi_mode(a)) {
        case EPI_PLAIN: return launch_8phase2<false, ACT, EPI_PLAIN, true>(a, s);
        case EPI_FOLD: return launch_8phase2<false, ACT, EPI_FOLD, true>(a, s);
        case EPI_LS:
            if constexpr (ACT == ACT_NONE) return launch_8phase2<false, ACT, EPI_LS, true>(a, s);
            return launch_8phase2<false, ACT, EPI_GENERIC, true>(a, s);
        default: return launch_8phase2<false, ACT, EPI_GENERIC, true>(a, s);
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool STAGGER, bool REMAP, int ACT, int EPI, int WHOLE = 0>
hipError_t launch_cfg2(const GemmArgs& a, hipStream_t s) {
    // the static epilogue forms for whole tiles (as launch_8phase2), on the tile configurations the encoder stacks' remaining rows take
    constexpr bool REST_CFG = (BM == 256 && BN == 128 && STAGGER) || (BM == 128 && BN == 128 && NSTAGE == 4) || (BM == 64 && BN == 128);
    if constexpr (MERV_GEMM_ALLVALID && WHOLE == 0 && REST_CFG && !REMAP && (EPI == EPI_PLAIN || EPI == EPI_LS || EPI == EPI_FOLD)) {
        if (a.M % BM == 0 && !a.mx_out_q) {
            const int f = 1 | (a.res ? 2 : 0) | (a.stats_out ? 4 : 0) | (a.row_add ? 8 : 0);
            if (f == 1) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT, EPI, 1>(a, s);
            if constexpr (EPI != EPI_FOLD && ACT == ACT_NONE) {
                if (f == 3) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT, EPI, 3>(a, s);
                if (f == 7) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT, EPI, 7>(a, s);
                if constexpr (EPI == EPI_PLAIN && !gemm_direct_epilogue<ACT>) {
                    if (f == 15) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT, EPI, 15>(a, s);
                }
            }
        }
    }
    constexpr int LDS = NSTAGE * (BM + BN) * ROW_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(LDS >= WM * WN * (BM / WM) * 128, "epilogue staging must fit in the stage ring");
    static_assert(!STAGGER || WM * WN == 8, "stagger pairs the two waves of each SIMD: 8-wave blocks only");
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT, EPI, WHOLE>;
    static bool attr_set[MAX_DEVICES] = {};  // per instantiation, per device
    if (hipError_t e = ensure_dynamic_lds((const void*)kern, LDS, attr_set); e != hipSuccess) return e;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(WM * WN * 64), LDS, s, a);
    return hipGetLastError();
}
template <int BM, int BN, int WM, int WN, int NSTAGE, bool STAGGER, int ACT>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int epi = epi_mode(a);
    if (a.out_group > 0 || a.res_row_mod > 0) {  // row remapping is only instantiated for the embedding's epilogues
        if constexpr (ACT == ACT_NONE) {
            if (epi == EPI_PLAIN) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, true, ACT, EPI_PLAIN>(a, s);
            if (epi == EPI_GENERIC) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, true, ACT, EPI_GENERIC>(a, s);
        }
        return hipErrorInvalidValue;
    }
    switch (epi) {
        case EPI_PLAIN: return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, false, ACT, EPI_PLAIN>(a, s);
        case EPI_FOLD: return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, false, ACT, EPI_FOLD>(a, s);
        case EPI_LS:
            if constexpr (ACT == ACT_NONE) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, false, ACT, EPI_LS>(a, s);
            return hipErrorInvalidValue;
        default: return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, false, ACT, EPI_GENERIC>(a, s);
    }
}

int g_gemm_variant = 0;
int g_gemm_group_m = 0;  // 0 auto, 1: 128x128, 2: 256x256, 3: 256x128

int num_cus() {
    static int n[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!n[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

// Tile choice (measured on MI355X, tools/gemm_bench.py): the 256x256 eight-phase kernel has by far the best in-round
// rate (~1.2 PFLOP/s at K = 1024, ~1.5 at K = 4096) but one block per CU and 256-row granularity, so a launch whose
// tile count is not a multiple of the CU count loses up to a whole round. plan_split() therefore gives it only the
// leading m-tiles that fill complete rounds of the chip; the remaining rows (incl. the ragged last m-tile) go to the
// 256x128 staggered kernel, or to 128x128 tiles (two blocks per CU) when that is too few blocks to occupy the chip.
int choose_variant(const GemmArgs& a) {
    if (g_gemm_variant) return g_gemm_variant;
    const long tiles = (long)((a.M + 255) / 256) * (a.N / 128);
    if (tiles >= 160) return 4;
    // a few hundred rows (what the eight-phase launch leaves of LanguageBind / DINOv2 at 16 videos: 256 / 1280 rows): the
    // launch is the latency of ONE block's K-loop, so the tile that fills the chip with the most, smallest blocks wins --
    // 64 x 128 with four waves of 32 x 64 (half the MFMAs per K-step of the 128 x 128 block, 4-deep ring) while that is at most
    // one block per CU: 9.0-9.9 us against 12.9-14.2 at K = 1024 and 23.4 against 36.9 at K = 4096 (tools/probes/gemm_remainder.py)
    const long blocks64 = (long)((a.M + 63) / 64) * (a.N / 128);
    if (blocks64 <= num_cus()) return 9;
    // few blocks: at most one 128x128 block per CU, so co-residency cannot hide the DMA latency of a 2-deep ring
    // (a lone block then runs ~1.3 us per K-step); a 4-deep ring keeps three tiles in flight instead
    const long small_tiles = (long)((a.M + 127) / 128) * (a.N / 128);
    if (small_tiles <= num_cus()) return 6;
    // 128-159 tiles of 256 x 128 (what ViViT / SigLIP leave of their N = 768 GEMMs at 16 videos: 6672 rows = 156 tiles): more
    // 128 x 128 blocks than CUs, and the staggered 256 x 128 kernel on half a round beats two co-resident 128 x 128 blocks
    // per CU by 2 x (14.1 vs 27.3 us at K = 768, 35.7 vs 86.1 at K = 3072; tools/probes/gemm_remainder.py). Measured at M = 3328 and
    // 6656 only: a few rows under a very wide N (M <= 1024: up to half of a 256-row tile would be padding) keep the 128 x 128 tile.
    return a.M > 1024 ? 4 : 1;
}

// Round 6, re-swept on the round's kernels (tools/sessions/gpu_r6_s3.sh, ms per step, two passes): one video 32: 9.97 / 9.97 -- 40: 9.79 / 9.81 -- 72: 9.71 / 9.60 --
// 96: 9.57 / 9.63; two videos 32: 15.90 / 15.82 -- 72: 15.97 / 15.87 -- 96: 15.99 / 16.02 (and 16.24 -> 16.42 against the round-5 library on another box:
// ViViT's / SigLIP's 72-tile launches at two videos want the eight-phase kernel); four videos unchanged. A sub-round launch lasts as long as ONE block's K-loop,
// which is MFMA-bound on its CU (0.93 us per K-tile with the chip part empty): the 68-tile launches of LanguageBind / DINOv2 at one video (proj, fc2: 24 / 66 us
// alone) finish in 17 / 45 us as 136 staggered 256 x 128 blocks, the 39-tile ones of ViViT / SigLIP (20 / 52 us) in 12 / 29 us as 150 blocks of 128 x 128.
// 72 = the smallest threshold that takes both.
constexpr long SUBROUND_MIN_TILES = 72;
// ... for the chain that ENDS the step. The other chains of a concurrent step keep the eight-phase kernel from 32 tiles on (round 5's value): their 39 / 68
// blocks then leave the CUs to the critical chain's wide launches -- one video, tools/sessions/gpu_r6_s17.sh, three alternating passes: 72 for every encoder
// 9.43-9.45 ms, for LanguageBind alone 8.84-8.97, for LanguageBind + DINOv2 9.15-9.25, for the three smaller ones alone 9.91-10.07, for none (round 5) 9.8-9.9.
// merv_encoder_set_latency_critical() carries the choice; a lone encoder is its own critical chain.
constexpr long SUBROUND_MIN_TILES_BESIDE = 32;
constexpr long BESIDE_MAX_TILES = 160, BESIDE_WIDE_N = 8;  // width cap of such a chain's wide launches (launch_gemm)
// rows (a multiple of 256 -- or all M rows -- possibly 0) the eight-phase kernel should take from the top of the problem
int plan_split(const GemmArgs& a) {
    if (g_gemm_variant != 0) return 0;
    const int nkt = a.K / BK;
    if (a.N % 256 != 0 || nkt < 4 || (nkt & 1) || a.out_group > 0 || a.res_row_mod > 0) return 0;
    const long tilesN = a.N / 256, full_m = a.M / 256;
    const long rounds = full_m * tilesN / num_cus();
    // less than one round (small batches): one partial round of the eight-phase kernel still beats two rounds of 256x128
    // tiles once it has enough blocks (measured end to end against never doing so: +12 % tokens/s at 1 video per step, +8 % at 2, +1 % at 4; the threshold
    // itself: see SUBROUND_MIN_TILES)
    if (rounds < 1) {
        static const long min_tiles = merv_tuning_env("MERV_SUBROUND_MIN_TILES") ? atol(merv_tuning_env("MERV_SUBROUND_MIN_TILES")) : SUBROUND_MIN_TILES;  // tuning hook
        // (experiment hook: MERV_SUBROUND_ONLY_M = "4112,4176": only launches of these row counts take `min_tiles`, every other one round 5's 32 -- i.e. the
        // chain that ends the step gets the fast wide form, the others keep the form that leaves it the CUs)
        static const char* only_m = merv_tuning_env("MERV_SUBROUND_ONLY_M");
        long mt = merv_tuning_env("MERV_SUBROUND_MIN_TILES") ? min_tiles : (a.subround_min_tiles > 0 ? a.subround_min_tiles : min_tiles);
        if (only_m) {
            bool hit = false;
            for (const char* q = only_m; *q;) { if (atol(q) == a.M) hit = true; while (*q && *q != ',') ++q; if (*q) ++q; }
            mt = hit ? min_tiles : SUBROUND_MIN_TILES_BESIDE;
        }
        return (full_m * tilesN >= mt) ? a.M : 0;
    }
    long k = rounds * num_cus() / tilesN;
    if (k > full_m) k = full_m;
    // (experiment hook, round 6: when what the complete rounds leave over is itself a large part of a round -- two to four videos: a third of a qkv launch --
    // take it in the eight-phase launch as a last, partial round instead of handing it to the small tiles)
    static const long rest8 = merv_tuning_env("MERV_REST8_MIN_TILES") ? atol(merv_tuning_env("MERV_REST8_MIN_TILES")) : 0;
    if (rest8 > 0 && (full_m - k) * tilesN >= rest8) return a.M;
    return (int)(k * 256);
}

template <int ACT>
hipError_t launch_act(const GemmArgs& a, hipStream_t s) {
    switch (choose_variant(a)) {
        case 3: return launch_cfg<256, 128, 4, 2, 3, false, ACT>(a, s);
        case 4: return launch_cfg<256, 128, 4, 2, 3, true, ACT>(a, s);
        case 6: return launch_cfg<128, 128, 2, 2, 4, false, ACT>(a, s);
        case 9: return launch_cfg<64, 128, 2, 2, 4, false, ACT>(a, s);
        case 7:
            if (a.N % 256 == 0 && (a.K / BK) % 2 == 0 && a.K / BK >= 4) return launch_8phase<ACT>(a, s);
            return launch_cfg<256, 128, 4, 2, 3, true, ACT>(a, s);
        default: return launch_cfg<128, 128, 2, 2, 2, false, ACT>(a, s);
    }
}

// GemmArgs::row_add (the next block's temporal embedding, added by fc2) exists in the LDS epilogue only: refused for launches that
// would take the direct one (activation launches; every launch of a -DMERV_GEMM_EPILOGUE=1 A/B build)
inline bool row_add_ok(const GemmArgs& a) {
    if (!a.row_add) return true;
    if (a.act != ACT_NONE || gemm_direct_epilogue<ACT_NONE>) return false;
    return a.row_add_div >= 256 && a.row_add_mod > 0 && a.out_group <= 0;  // (an MXFP8 output quantises the values AFTER the add: fine)
}

}  // namespace

// Round-6 experiment (hooks build only): the remaining-rows launch of a GEMM forked onto a sibling stream, so that it can start in
// its own main launch's tail instead of behind it. set_rest_fork(main, aux) registers the sibling of a stream (the caller picks one that owns an otherwise
// idle hardware queue); launch_gemm then brackets the second launch with two events. Measured: profiles/r06_remainder_fork.json.
namespace {
struct RestFork { hipStream_t main, aux; hipEvent_t fork, join; };
RestFork g_rest_fork[8];
int g_rest_forks = 0;
}
void set_rest_fork(hipStream_t main, hipStream_t aux) {
    if constexpr (MERV_HOOKS) {
        if (!main) { g_rest_forks = 0; return; }
        for (int i = 0; i < g_rest_forks; ++i)
            if (g_rest_fork[i].main == main) { g_rest_fork[i].aux = aux; return; }
        if (g_rest_forks < 8) {
            RestFork& f = g_rest_fork[g_rest_forks];
            f.main = main; f.aux = aux;
            if (hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&f.join, hipEventDisableTiming) == hipSuccess)
                ++g_rest_forks;
        }
    }
}

void set_gemm_variant(int v) {  // (product build: a no-op -- see merv_tuning_env, common.h)
    if constexpr (MERV_HOOKS) { g_gemm_variant = v & 0xff; g_gemm_group_m = (v >> 8) & 0xff; }
}

// rows [r0, r0 + rows) of a problem as a problem of its own (r0 a multiple of 256; plain row-major launches: no row remapping)
static GemmArgs slice_rows(const GemmArgs& a, int r0, int rows) {
    GemmArgs g = a;
    g.M = rows;
    g.A = a.A + (size_t)r0 * a.lda;
    g.C = a.C + (size_t)r0 * a.ldc;
    if (a.res) g.res = a.res + (size_t)r0 * a.ldres;
    if (a.row_stats) g.row_stats = a.row_stats + 2 * (size_t)r0;
    g.row_add_row0 = a.row_add_row0 + r0;
    if (a.stats_out) g.stats_out = a.stats_out + 2 * (size_t)r0;  // same stats_ld: the rows below
    if (a.mx_out_q) {  // whole 64-row scale groups; the K-tile stride (mx_out_groups) is unchanged
        g.mx_out_q = a.mx_out_q + (size_t)r0 * a.N;
        g.mx_out_scales = a.mx_out_scales + (size_t)(r0 / 64) * 256;
    }
    return g;
}

// Host launcher. Requirements (checked): K % 64 == 0, N % 128 == 0, lda/ldw/ldc % 8 == 0.
hipError_t launch_gemm(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0) return hipSuccess;
    // A chain that runs BESIDE the one that ends the step (GemmArgs::subround_min_tiles > 0: merv_encoder_set_latency_critical(0)) gives up some of the
    // chip's WIDTH as well: its wide launches (>= 8 column tiles: qkv, fc1) of more than 160 tiles but no more than a round go out as consecutive launches of
    // at most 160, so that it never holds more than 160 CUs while the critical chain's launches wait (its rows are independent: same bits). One video,
    // tools/sessions/gpu_r6_s22.sh, three alternating passes: 9.01-9.07 -> 8.83-8.90 ms (cap 128: 8.78-8.96, 112: 8.88-8.99); two to four videos within
    // 0.4 %. Narrow launches are never split (a 132-tile launch cut at 128 costs two videos 4 %: gpu_r6_s21.sh). Hooks: MERV_BESIDE_MAX_TILES / _WIDE_N.
    {
        static const long cap = merv_tuning_env("MERV_BESIDE_MAX_TILES") ? atol(merv_tuning_env("MERV_BESIDE_MAX_TILES")) : BESIDE_MAX_TILES;
        static const long wide_n = merv_tuning_env("MERV_BESIDE_WIDE_N") ? atol(merv_tuning_env("MERV_BESIDE_WIDE_N")) : BESIDE_WIDE_N;
        static const long nocap_m = merv_tuning_env("MERV_BESIDE_NOCAP_M") ? atol(merv_tuning_env("MERV_BESIDE_NOCAP_M")) : -1;  // (experiment: this row count keeps its width)
        if (cap > 0 && a_in.subround_min_tiles > 0 && a_in.out_group <= 0 && a_in.res_row_mod <= 0 && a_in.N % 256 == 0 && a_in.M != nocap_m) {
            const long tilesN = a_in.N / 256, tiles = (long)((a_in.M + 255) / 256) * tilesN;
            const long mt = cap / tilesN;
            if (tiles > cap && tiles <= num_cus() + tilesN && mt >= 1 && tilesN >= wide_n) {
                GemmArgs whole = a_in;
                if (whole.stats_out && whole.stats_ld <= 0) whole.stats_ld = whole.M;
                for (int r0 = 0; r0 < a_in.M; r0 += (int)mt * 256) {
                    const int rows = a_in.M - r0 < (int)mt * 256 ? a_in.M - r0 : (int)mt * 256;
                    GemmArgs g = slice_rows(whole, r0, rows);
                    g.subround_min_tiles = -1;  // (not again; the slice keeps the eight-phase form from 32 tiles on)
                    if (hipError_t e = launch_gemm(g, s); e != hipSuccess) return e;
                }
                return hipSuccess;
            }
        }
    }
    GemmArgs a = a_in;
    if (a.subround_min_tiles < 0) a.subround_min_tiles = (int)SUBROUND_MIN_TILES_BESIDE;
    if (g_gemm_group_m > 0) a.group_m = g_gemm_group_m;
    if (a.stats_out && a.stats_ld <= 0) a.stats_ld = a.M;  // rows of the partials array: [N / 64][stats_ld][2]
    if (a.K % BK != 0 || a.N % 128 != 0 || a.K <= 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.ldc) % 8 != 0) return hipErrorInvalidValue;
    if (a.res && a.ldres % 8 != 0) return hipErrorInvalidValue;
    if (!row_add_ok(a)) return hipErrorInvalidValue;
    if ((double)a.lda * 512 >= 4294967296.0 || (double)a.ldw * 512 >= 4294967296.0) return hipErrorInvalidValue;  // a tile's DMA piece offsets are 32-bit
    {   // the epilogue keeps element offsets in 32 bits
        const double rows_out = a.out_group > 0 ? ((double)(a.M / a.out_group) + 1) * a.out_stride + a.out_off : (double)a.M;
        if (rows_out * a.ldc >= 4294967296.0 || (a.res && (double)a.M * a.ldres >= 4294967296.0)) return hipErrorInvalidValue;
    }
    ProfScope ps(PROF_GEMM, s, 2.0 * a.M * a.N * a.K,
                 2.0 * ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N * (a.res ? 2 : 1)));
    auto dispatch = [&](const GemmArgs& g, bool eight_phase) -> hipError_t {
        ProfScope pk(eight_phase ? (g.act == ACT_NONE ? PROF_K_GEMM8_PLAIN : PROF_K_GEMM8_ACT) : PROF_K_GEMM_SMALL, s,
                     2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * (g.res ? 2 : 1)));
        switch (g.act) {
            case ACT_NONE: return eight_phase ? launch_8phase<ACT_NONE>(g, s) : launch_act<ACT_NONE>(g, s);
            case ACT_GELU_ERF: return eight_phase ? launch_8phase<ACT_GELU_ERF>(g, s) : launch_act<ACT_GELU_ERF>(g, s);
            case ACT_GELU_TANH: return eight_phase ? launch_8phase<ACT_GELU_TANH>(g, s) : launch_act<ACT_GELU_TANH>(g, s);
            case ACT_QUICK_GELU: return eight_phase ? launch_8phase<ACT_QUICK_GELU>(g, s) : launch_act<ACT_QUICK_GELU>(g, s);
            default: return hipErrorInvalidValue;
        }
    };
    const int rows1 = plan_split(a);
    if (rows1 == 0) return dispatch(a, false);
    GemmArgs top = a;
    top.M = rows1;
    const RestFork* fork = nullptr;
    if constexpr (MERV_HOOKS) {
        if (rows1 != a.M)
            for (int i = 0; i < g_rest_forks; ++i)
                if (g_rest_fork[i].main == s && g_rest_fork[i].aux) fork = &g_rest_fork[i];
        if (fork && hipEventRecord(fork->fork, s) != hipSuccess) fork = nullptr;  // everything the two launches read is behind this point
    }
    hipError_t e = dispatch(top, true);
    if (e != hipSuccess || rows1 == a.M) return e;
    if constexpr (MERV_PROBE_REST_MODE != 0) return MERV_PROBE_REST_LAUNCH(s, e);
    GemmArgs rest = a;  // independent rows: same stream, no ordering requirement between the two launches
    rest.M = a.M - rows1;
    rest.A = a.A + (size_t)rows1 * a.lda;
    rest.C = a.C + (size_t)rows1 * a.ldc;
    if (a.res) rest.res = a.res + (size_t)rows1 * a.ldres;
    if (a.row_stats) rest.row_stats = a.row_stats + 2 * (size_t)rows1;
    rest.row_add_row0 = a.row_add_row0 + rows1;
    if (a.stats_out) rest.stats_out = a.stats_out + 2 * (size_t)rows1;  // same stats_ld: the rows below the first launch's
    if (a.mx_out_q) {  // rows1 is a multiple of 256: whole 64-row scale groups; the K-tile stride (mx_out_groups) is unchanged
        rest.mx_out_q = a.mx_out_q + (size_t)rows1 * a.N;
        rest.mx_out_scales = a.mx_out_scales + (size_t)(rows1 / 64) * 256;
    }
    if constexpr (MERV_HOOKS) {
        if (fork) {  // the remaining rows on the sibling stream, joined back before anything that follows on s
            if (hipError_t fe = hipStreamWaitEvent(fork->aux, fork->fork, 0); fe != hipSuccess) return fe;
            hipStream_t main = s;
            s = fork->aux;
            e = dispatch(rest, false);
            s = main;
            if (e != hipSuccess) return e;
            if (hipError_t fe = hipEventRecord(fork->join, fork->aux); fe != hipSuccess) return fe;
            return hipStreamWaitEvent(s, fork->join, 0);
        }
    }
    return dispatch(rest, false);
}

// MXFP8 GEMM (A, W: OCP e4m3 bytes, row-major, K contiguous; block scales in the layout mx_quantize writes).
// Requirements: K % 256 == 0 and K >= 512 (an even number >= 4 of 128-element K-tiles), N % 256 == 0, no row remapping.
hipError_t launch_gemm_mx(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0) return hipSuccess;
    GemmArgs a = a_in;
    if (a.stats_out && a.stats_ld <= 0) a.stats_ld = a.M;  // as launch_gemm: rows of the partials array [N / 64][stats_ld][2]
    if (!row_add_ok(a)) return hipErrorInvalidValue;
    if (!a.mx_scale_a || !a.mx_scale_w || a.mx_groups_a < a.mx_group0_a + (a.M + 63) / 64 || a.mx_group0_a < 0 || a.mx_groups_w < a.N / 64) return hipErrorInvalidValue;
    if (a.K % 256 != 0 || a.K < 512 || a.N % 256 != 0 || a.out_group > 0 || a.res_row_mod > 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw) % 16 != 0 || a.ldc % 8 != 0 || (a.res && a.ldres % 8 != 0)) return hipErrorInvalidValue;
    if ((double)a.M * a.ldc >= 4294967296.0 || (a.res && (double)a.M * a.ldres >= 4294967296.0)) return hipErrorInvalidValue;
    ProfScope ps(PROF_GEMM, s, 2.0 * a.M * a.N * a.K,
                 (double)a.M * a.K + (double)a.N * a.K + 2.0 * (double)a.M * a.N * (a.res ? 2 : 1));
    auto dispatch = [&](const GemmArgs& g) -> hipError_t {
        ProfScope pk(g.act == ACT_NONE ? PROF_K_GEMM8_PLAIN : PROF_K_GEMM8_ACT, s, 2.0 * g.M * g.N * g.K,
                     (double)g.M * g.K + (double)g.N * g.K + 2.0 * (double)g.M * g.N * (g.res ? 2 : 1));
        switch (g.act) {
            case ACT_NONE: return launch_8phase_mx<ACT_NONE>(g, s);
            case ACT_GELU_ERF: return launch_8phase_mx<ACT_GELU_ERF>(g, s);
            case ACT_GELU_TANH: return launch_8phase_mx<ACT_GELU_TANH>(g, s);
            case ACT_QUICK_GELU: return launch_8phase_mx<ACT_QUICK_GELU>(g, s);
            default: return hipErrorInvalidValue;
        }
    };
    // A ragged last tile (ViViT: 16 videos x 3137 rows = 196 tiles + 16 rows) would put the WHOLE launch on the run-time epilogue form: the whole
    // tiles go first in their static form, the remaining rows follow as a second launch of the generic one (independent rows, same stream).
    const int rows1 = a.M / 256 * 256;
    GemmArgs top = a;
    top.M = rows1;
    if (rows1 == 0 || rows1 == a.M || !mx_static_form(top)) return dispatch(a);
    if (hipError_t e = dispatch(top); e != hipSuccess) return e;
    GemmArgs rest = a;
    rest.M = a.M - rows1;
    rest.A = (const bf16_t*)((const uint8_t*)a.A + (size_t)rows1 * a.lda);  // (e4m3 bytes: lda is in elements = bytes)
    rest.mx_group0_a = a.mx_group0_a + rows1 / 64;
    rest.C = a.C + (size_t)rows1 * a.ldc;
    if (a.res) rest.res = a.res + (size_t)rows1 * a.ldres;
    if (a.row_stats) rest.row_stats = a.row_stats + 2 * (size_t)rows1;
    rest.row_add_row0 = a.row_add_row0 + rows1;
    if (a.stats_out) rest.stats_out = a.stats_out + 2 * (size_t)rows1;
    if (a.mx_out_q) {
        rest.mx_out_q = a.mx_out_q + (size_t)rows1 * a.N;
        rest.mx_out_scales = a.mx_out_scales + (size_t)(rows1 / 64) * 256;
    }
    return dispatch(rest);
}

}  // namespace merv
