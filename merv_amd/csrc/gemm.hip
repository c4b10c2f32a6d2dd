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
// Structure (cdna_hip_programming.md section 5): 128x128x64 block tile, 4 waves (2x2, 64x64 each),
// v_mfma_f32_16x16x32_bf16, operands staged global->LDS by LDS-DMA (global_load_lds_dwordx4), two LDS
// stages so tile t+1 is in flight while tile t is multiplied, counted vmcnt + raw s_barrier (never a
// __syncthreads() while a DMA is outstanding). The LDS image is lane-linear (DMA constraint), so the bank
// swizzle chunk ^= (row & 7) is applied to the per-lane *source* address and to the ds_read address
// (section 5.4 rule 21). Operands are swapped (D^T = W * A^T) so that every lane ends up with 4 consecutive
// n of one output row: bias / LayerScale / residual / store are then 8-16 byte vector accesses.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include <math.h>
#include <type_traits>

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

// ---- epilogue (shared by all tile configurations) ----
template <int WTM, int WTN, bool REMAP, int ACT>
MERV_DEVICE void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[WTN / 16][WTM / 16], char* smem, int wave, int lane, int m0,
                               int n0, int wr, int wc) {
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
    uint32_t c_off[EP_IT];  // element offsets (the launcher checks they fit 32 bits)
    bool valid[EP_IT];
    u32x4 resv[EP_IT];
    // opaque copy of the lane id: keeps the compiler from hoisting the epilogue's index arithmetic (integer
    // divisions) above the K-loop, where it would sit in registers -- or scratch -- for the whole kernel
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int ec = elane & 7;  // this lane's 16-byte chunk (8 columns) of each row it handles
    uint32_t r_off[EP_IT];
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
        const int r = (elane >> 3) + 8 * it;
        const int m = m0 + wr * WTM + r;
        valid[it] = m < p.M;
        const int mc = valid[it] ? m : p.M - 1;  // clamp instead of branching: loads stay unconditional
        int orow = mc, rr = mc;
        if constexpr (REMAP) {  // patch-embedding launch only: scatter past prefix tokens, position row m % P
            if (p.out_group > 0) orow = (mc / p.out_group) * p.out_stride + p.out_off + (mc % p.out_group);
            if (p.res_row_mod > 0) rr = mc % p.res_row_mod;
        }
        c_off[it] = (uint32_t)orow * (uint32_t)p.ldc + wn0 + ec * 8;
        r_off[it] = (uint32_t)rr * (uint32_t)p.ldres + wn0 + ec * 8;
    }
    // one wave-uniform branch around ALL residual loads (a per-element select would serialise them behind
    // vmcnt(0) waits: cdna_hip_programming.md, "Three .s-level traps" (c))
    if (p.res) {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) resv[it] = *(const u32x4*)(p.res + (size_t)r_off[it]);
    } else {
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) resv[it] = u32x4{0u, 0u, 0u, 0u};
    }
    float4 bias4[NI], ls4[NI];
    if (p.bias) {
#pragma unroll
        for (int i = 0; i < NI; ++i) bias4[i] = *(const float4*)(p.bias + wn0 + i * 16 + fq * 4);
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) bias4[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (p.lscale) {
#pragma unroll
        for (int i = 0; i < NI; ++i) ls4[i] = *(const float4*)(p.lscale + wn0 + i * 16 + fq * 4);
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) ls4[i] = float4{1.f, 1.f, 1.f, 1.f};
    }
    // every wave is done with the stage ring
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    char* stg = smem + wave * (WTM * 128);  // [WTM rows][128 B], 16-byte chunks XOR-swizzled by (row & 7)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const f32x4 v = acc[i][j];
            const f32x2 lo = activate2<ACT>(f32x2{v[0], v[1]} + f32x2{bias4[i].x, bias4[i].y}) * f32x2{ls4[i].x, ls4[i].y};
            const f32x2 hi = activate2<ACT>(f32x2{v[2], v[3]} + f32x2{bias4[i].z, bias4[i].w}) * f32x2{ls4[i].z, ls4[i].w};
            u32x2 o;
            o[0] = pack2bf(lo[0], lo[1]);
            o[1] = pack2bf(hi[0], hi[1]);
            const int row = j * 16 + frow;
            const int chunk = (2 * i + (fq >> 1)) ^ (row & 7);
            *(u32x2*)(stg + row * 128 + chunk * 16 + 8 * (fq & 1)) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: in-wave ordering is enough
#pragma unroll
    for (int it = 0; it < EP_IT; ++it) {
        const int r = (elane >> 3) + 8 * it;
        u32x4 t = *(const u32x4*)(stg + r * 128 + ((ec ^ (r & 7)) * 16));
        if (p.res) {
            // bf16(linear) + bf16(residual), rounded once more: the reference's own order under autocast
#pragma unroll
            for (int q = 0; q < 4; ++q)
                t[q] = pack2bf(bflo(t[q]) + bflo(resv[it][q]), bfhi(t[q]) + bfhi(resv[it][q]));
        }
        if (valid[it]) *(u32x4*)(p.C + (size_t)c_off[it]) = t;
    }
}

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
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool STAGGER, bool REMAP, int ACT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_kernel(GemmArgs p) {
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
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        w_src[i] = p.W + (size_t)(n0 + row) * p.ldw + (((lane & 7) ^ (row & 7)) * 8);
    }
    // piece `pc` (0..DPS-1) of tile kt into stage `buf`
    auto dma_piece = [&](int kt, int buf, int pc) {
        char* a_tile = smem + buf * STAGE_BYTES;
        const bf16_t* src = pc < A_PIECES ? a_src[pc < A_PIECES ? pc : 0] : w_src[pc >= A_PIECES ? pc - A_PIECES : 0];
        char* dst = pc < A_PIECES ? a_tile + (pc * NW + wave) * 1024 : a_tile + A_BYTES + ((pc - A_PIECES) * NW + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kt * BK),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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

    gemm_epilogue<WTM, WTN, REMAP, ACT>(p, acc, smem, wave, lane, m0, n0, wr, wc);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool STAGGER, bool REMAP, int ACT>
hipError_t launch_cfg2(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = NSTAGE * (BM + BN) * ROW_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(LDS >= WM * WN * (BM / WM) * 128, "epilogue staging must fit in the stage ring");
    static_assert(!STAGGER || WM * WN == 8, "stagger pairs the two waves of each SIMD: 8-wave blocks only");
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, STAGGER, REMAP, ACT>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(WM * WN * 64), LDS, s, a);
    return hipGetLastError();
}
template <int BM, int BN, int WM, int WN, int NSTAGE, bool STAGGER, int ACT>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    if (a.out_group > 0 || a.res_row_mod > 0) {
        if constexpr (ACT == ACT_NONE) return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, true, ACT>(a, s);
        else return hipErrorInvalidValue;  // row remapping is only instantiated for the plain (embedding) epilogue
    }
    return launch_cfg2<BM, BN, WM, WN, NSTAGE, STAGGER, false, ACT>(a, s);
}

int g_gemm_variant = 0;
int g_gemm_group_m = 0;  // 0 auto, 1: 128x128, 2: 256x256, 3: 256x128

// Tile choice (measured on MI355X, tools/gemm_bench.py, profiles/r01_gemm_tiles.md): the 256x128 tile with a 3-deep
// LDS ring and staggered half-blocks is the fastest configuration whenever it yields enough blocks to occupy the
// chip; small launches use 128x128 tiles, two blocks per CU. The 256x256 tile has the best bytes/FLOP but pays the
// longest per-tile prologue + epilogue and the worst last-round quantisation (one block per CU).
int choose_variant(const GemmArgs& a) {
    if (g_gemm_variant) return g_gemm_variant;
    const long tiles = (long)((a.M + 255) / 256) * (a.N / 128);
    return tiles >= 160 ? 4 : 1;
}

template <int ACT>
hipError_t launch_act(const GemmArgs& a, hipStream_t s) {
    switch (choose_variant(a)) {
        case 2: return launch_cfg<256, 256, 2, 4, 2, false, ACT>(a, s);
        case 3: return launch_cfg<256, 128, 4, 2, 3, false, ACT>(a, s);
        case 4: return launch_cfg<256, 128, 4, 2, 3, true, ACT>(a, s);
        case 5: return launch_cfg<256, 256, 2, 4, 2, true, ACT>(a, s);
        default: return launch_cfg<128, 128, 2, 2, 2, false, ACT>(a, s);
    }
}

}  // namespace

void set_gemm_variant(int v) { g_gemm_variant = v & 0xff; g_gemm_group_m = (v >> 8) & 0xff; }

// Host launcher. Requirements (checked): K % 64 == 0, N % 128 == 0, lda/ldw/ldc % 8 == 0.
hipError_t launch_gemm(const GemmArgs& a_in, hipStream_t s) {
    if (a_in.M <= 0) return hipSuccess;
    GemmArgs a = a_in;
    if (g_gemm_group_m > 0) a.group_m = g_gemm_group_m;
    if (a.K % BK != 0 || a.N % 128 != 0 || a.K <= 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.ldc) % 8 != 0) return hipErrorInvalidValue;
    if (a.res && a.ldres % 8 != 0) return hipErrorInvalidValue;
    {   // the epilogue keeps element offsets in 32 bits
        const double rows_out = a.out_group > 0 ? ((double)(a.M / a.out_group) + 1) * a.out_stride + a.out_off : (double)a.M;
        if (rows_out * a.ldc >= 4294967296.0 || (a.res && (double)a.M * a.ldres >= 4294967296.0)) return hipErrorInvalidValue;
    }
    if (g_gemm_variant == 2 && a.N % 256 != 0) return hipErrorInvalidValue;
    ProfScope ps(PROF_GEMM, s, 2.0 * a.M * a.N * a.K,
                 2.0 * ((double)a.M * a.K + (double)a.N * a.K + (double)a.M * a.N * (a.res ? 2 : 1)));
    switch (a.act) {
        case ACT_NONE: return launch_act<ACT_NONE>(a, s);
        case ACT_GELU_ERF: return launch_act<ACT_GELU_ERF>(a, s);
        case ACT_GELU_TANH: return launch_act<ACT_GELU_TANH>(a, s);
        case ACT_QUICK_GELU: return launch_act<ACT_QUICK_GELU>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace merv
