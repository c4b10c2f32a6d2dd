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
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int ACT>
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
    constexpr int GM = (BM >= 256) ? 4 : 8;
    const int per_group = GM * tilesN;
    const int grp = id / per_group;
    const int first_m = grp * GM;
    const int gsz = (tilesM - first_m) < GM ? (tilesM - first_m) : GM;
    const int in_grp = id - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // piece `pc` (0..DPS-1) of tile kt into stage `buf`
    auto dma_piece = [&](int kt, int buf, int pc) {
        char* a_tile = smem + buf * STAGE_BYTES;
        if (pc < A_PIECES) dma_rows8(p.A, p.lda, m0, p.M - 1, pc * NW + wave, kt * BK, a_tile, lane);
        else dma_rows8(p.W, p.ldw, n0, p.N - 1, (pc - A_PIECES) * NW + wave, kt * BK, a_tile + A_BYTES, lane);
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

    // one K-step; PREFETCH is a compile-time flag so the steady-state body is one straight-line scheduling region
    auto k_step = [&](int kt, int buf, auto prefetch_tag) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value;
        const int kt_next = kt + NSTAGE - 1;
        int buf_next = buf + NSTAGE - 1;
        if (buf_next >= NSTAGE) buf_next -= NSTAGE;
        const char* a_tile = smem + buf * STAGE_BYTES + (wr * WTM + frow) * ROW_BYTES;
        const char* w_tile = smem + buf * STAGE_BYTES + A_BYTES + (wc * WTN + frow) * ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int coff = (((kk * 4) + fq) ^ sw) * 16;
            bf16x8 af[MI], wf[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) af[j] = *(const bf16x8*)(a_tile + j * 16 * ROW_BYTES + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8*)(w_tile + i * 16 * ROW_BYTES + coff);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                if constexpr (PREFETCH) {
                    // DMA pieces of the prefetched tile, spread over the (BK/32)*NI MFMA rows of this K-step
                    constexpr int ROWS = (BK / 32) * NI;
                    const int r = kk * NI + i;
#pragma unroll
                    for (int pc = 0; pc < DPS; ++pc)
                        if (pc * ROWS / DPS == r) dma_piece(kt_next, buf_next, pc);
                }
            }
        }
    };

    int buf = 0;
    int kt = 0;
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

    // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of each 16x16 fragment ----
    // All loads (bias / LayerScale per n-fragment, residual per fragment) are issued BEFORE the first store: C may
    // alias the residual (in-place x += ...), so a load placed after a store could not be hoisted by the compiler
    // and every fragment would pay a full memory round trip.
    float4 bias4[NI], ls4[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + wc * WTN + i * 16 + fq * 4;
        bias4[i] = p.bias ? *(const float4*)(p.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
        ls4[i] = p.lscale ? *(const float4*)(p.lscale + n) : float4{1.f, 1.f, 1.f, 1.f};
    }
    size_t c_off[MI];
    bool valid[MI];
    u32x2 resv[NI][MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m0 + wr * WTM + j * 16 + frow;
        valid[j] = m < p.M;
        int orow = m;
        if (p.out_group > 0) orow = (m / p.out_group) * p.out_stride + p.out_off + (m % p.out_group);
        c_off[j] = (size_t)orow * p.ldc;
        if (p.res) {
            const int rr = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
            const bf16_t* rrow = p.res + (size_t)rr * p.ldres + n0 + wc * WTN + fq * 4;
#pragma unroll
            for (int i = 0; i < NI; ++i) resv[i][j] = valid[j] ? *(const u32x2*)(rrow + i * 16) : u32x2{0u, 0u};
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) resv[i][j] = u32x2{0u, 0u};
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = n0 + wc * WTN + i * 16 + fq * 4;
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            f32x4 v = acc[i][j];
            v[0] = activate<ACT>(v[0] + bias4[i].x) * ls4[i].x + bflo(resv[i][j][0]);
            v[1] = activate<ACT>(v[1] + bias4[i].y) * ls4[i].y + bfhi(resv[i][j][0]);
            v[2] = activate<ACT>(v[2] + bias4[i].z) * ls4[i].z + bflo(resv[i][j][1]);
            v[3] = activate<ACT>(v[3] + bias4[i].w) * ls4[i].w + bfhi(resv[i][j][1]);
            u32x2 o;
            o[0] = pack2bf(v[0], v[1]);
            o[1] = pack2bf(v[2], v[3]);
            if (valid[j]) *(u32x2*)(p.C + c_off[j] + n) = o;
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int ACT>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = NSTAGE * (BM + BN) * ROW_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, ACT>;
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

int g_gemm_variant = 0;  // 0 auto, 1: 128x128, 2: 256x256, 3: 256x128

// Tile choice (measured on MI355X, tools/gemm_bench.py): the 256x128 tile with a 3-deep LDS ring is the fastest
// configuration whenever it yields enough blocks to occupy the chip; small launches use 128x128 tiles, two blocks per
// CU. The 256x256 tile has the best bytes/FLOP but pays the longest per-tile prologue + epilogue (one block per CU).
int choose_variant(const GemmArgs& a) {
    if (g_gemm_variant) return g_gemm_variant;
    const long tiles3 = (long)((a.M + 255) / 256) * (a.N / 128);
    return tiles3 >= 200 ? 3 : 1;
}

template <int ACT>
hipError_t launch_act(const GemmArgs& a, hipStream_t s) {
    switch (choose_variant(a)) {
        case 2: return launch_cfg<256, 256, 2, 4, 2, ACT>(a, s);
        case 3: return launch_cfg<256, 128, 4, 2, 3, ACT>(a, s);
        case 4: return launch_cfg<128, 128, 2, 2, 3, ACT>(a, s);
        case 5: return launch_cfg<128, 128, 2, 2, 4, ACT>(a, s);
        default: return launch_cfg<128, 128, 2, 2, 2, ACT>(a, s);
    }
}

}  // namespace

void set_gemm_variant(int v) { g_gemm_variant = v; }

// Host launcher. Requirements (checked): K % 64 == 0, N % 128 == 0, lda/ldw/ldc % 8 == 0.
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.K % BK != 0 || a.N % 128 != 0 || a.K <= 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.ldc) % 8 != 0) return hipErrorInvalidValue;
    if (a.res && a.ldres % 4 != 0) return hipErrorInvalidValue;
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
