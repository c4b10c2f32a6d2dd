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

template <int ACT>
MERV_DEVICE void epilogue_store4(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M) return;
    if (p.bias) {
        const float4 b = *(const float4*)(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = activate<ACT>(v[i]);
    if (p.lscale) {
        const float4 s = *(const float4*)(p.lscale + n);
        v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
    }
    if (p.res) {
        const int rr = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
        const u32x2 r = *(const u32x2*)(p.res + (size_t)rr * p.ldres + n);
        v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
    }
    int orow = m;
    if (p.out_group > 0) orow = (m / p.out_group) * p.out_stride + p.out_off + (m % p.out_group);
    u32x2 o;
    o[0] = pack2bf(v[0], v[1]);
    o[1] = pack2bf(v[2], v[3]);
    *(u32x2*)(p.C + (size_t)orow * p.ldc + n) = o;
}

// BM x BN block tile, WAVES_M x WAVES_N waves.
template <int BM, int BN, int WAVES_M, int WAVES_N, int ACT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // per-wave output tile
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int DMA_PER_STAGE = (BM / 8 + BN / 8) / NW;  // LDS-DMA instructions per wave per stage
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over waves");

    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    // ---- tile selection: XCD-contiguous ids, then grouped (GM m-tiles per column sweep) ordering ----
    const int tilesM = (p.M + BM - 1) / BM, tilesN = p.N / BN;
    const int nwg = tilesM * tilesN;
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GM = 8;
    const int per_group = GM * tilesN;
    const int grp = id / per_group;
    const int first_m = grp * GM;
    const int gsz = (tilesM - first_m) < GM ? (tilesM - first_m) : GM;
    const int in_grp = id - grp * per_group;
    const int tm = first_m + in_grp % gsz;
    const int tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    auto stage = [&](int kt, int buf) {
        char* a_tile = smem + buf * STAGE_BYTES;
        char* w_tile = a_tile + A_BYTES;
#pragma unroll
        for (int i = 0; i < BM / 8 / NW; ++i)
            dma_rows8(p.A, p.lda, m0, p.M - 1, i * NW + wave, kt * BK, a_tile, lane);
#pragma unroll
        for (int i = 0; i < BN / 8 / NW; ++i)
            dma_rows8(p.W, p.ldw, n0, p.N - 1, i * NW + wave, kt * BK, w_tile, lane);
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

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) {
            stage(kt + 1, buf ^ 1);
            // all but the youngest DMA_PER_STAGE DMAs (tile kt+1) have landed => tile kt is in LDS
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA_PER_STAGE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
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
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        // every wave's LDS reads of this stage are complete before the next iteration's DMA may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of each 16x16 fragment ----
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int n = n0 + wc * WTN + i * 16 + fq * 4;
            const int m = m0 + wr * WTM + j * 16 + frow;
            epilogue_store4<ACT>(p, m, n, acc[i][j]);
        }
}

template <int ACT>
hipError_t launch_act(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128, BN = 128;
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.N / BN;
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 2, 2, ACT>), dim3(tilesM * tilesN), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace

// Host launcher. Requirements (checked): K % 64 == 0, N % 128 == 0, lda/ldw/ldc/ldres % 8 == 0.
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.K % BK != 0 || a.N % 128 != 0 || a.K <= 0) return hipErrorInvalidValue;
    if ((a.lda | a.ldw | a.ldc) % 8 != 0) return hipErrorInvalidValue;
    if (a.res && a.ldres % 4 != 0) return hipErrorInvalidValue;
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
