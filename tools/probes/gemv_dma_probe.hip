// Weight-streaming GEMV through an LDS-DMA ring (gfx950): is the decode step's GEMV bound by the bytes a CU can keep in flight from
// registers? Product form (decode.hip gemv_kernel): each lane holds UN x ROWS 16-byte weight chunks in VGPRs per trip (8 KB per wave,
// 64 KB per CU) and requests the next trip after it has multiplied the current one. This probe streams the same rows through a
// per-wave ring of R one-KiB LDS slots filled by global_load_lds_dwordx4 in scalar-base form (no VGPRs, no address arithmetic:
// 16 KB per wave, 128 KB per CU in flight at R = 16), reads each piece back with one ds_read_b128 per lane and multiplies in the
// product's per-lane chunk order (chunk lane + 64 p of row r: same bits).
//   gemv_dma_probe -> per shape: us per launch (weights rotated through 8 copies: no Infinity Cache reuse), TB/s, max |err| against fp64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef uint16_t bf16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__device__ inline float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ inline float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NP: 1-KiB pieces per row (K = 512 NP, or 512 NP - 256 with HALF_LAST: the last piece's upper 32 lanes are padding)
template <int ROWS, int NP, bool HALF_LAST, int R, int UNR>
__global__ __launch_bounds__(256) void gemv_dma(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * 4 + wave) * ROWS;
    if (n0 >= N) return;
    constexpr int TOTAL = ROWS * NP;
    static_assert(TOTAL % UNR == 0 && R % UNR == 0 && R <= TOTAL, "ring / batch geometry");
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (R * 1024);
    const char* wbase = (const char*)W + (size_t)n0 * K * 2;  // wave-uniform
    const unsigned voff = lane * 16;
    const unsigned voff_last = (HALF_LAST && lane >= 32) ? (lane - 32) * 16 : lane * 16;  // padding lanes re-read valid bytes (their x is 0)
    auto issue = [&](int i) {  // piece i = (row i / NP, piece i % NP) -> ring slot i % R
        const int r = i / NP, p = i % NP;
        const char* src = wbase + (size_t)r * K * 2 + p * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(p == NP - 1 ? voff_last : voff), "s"(src), "s"(ring + (i % R) * 1024) : "memory");
    };
#pragma unroll
    for (int i = 0; i < R; ++i) issue(i);
    // x: chunk lane + 64 p per piece (behind the ring's first fill in the wave's in-order queue; first used after the first pieces land)
    u32x4 xv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bool pad = HALF_LAST && p == NP - 1 && lane >= 32;
        xv[p] = pad ? u32x4{0u, 0u, 0u, 0u} : *(const u32x4*)(x + (size_t)(lane + 64 * p) * 8);
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    // the BUILTIN wait: hipcc retires the x loads in its own bookkeeping (after an asm wait it would still count them and put
    // vmcnt(7) .. vmcnt(0) in front of their first uses, draining the ring it cannot see)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): x has landed (and with it the first fill)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i0 = 0; i0 < TOTAL; i0 += UNR) {
        // outstanding pieces before this batch: min(R, TOTAL - i0); its UNR oldest must have landed
        constexpr int dummy = 0; (void)dummy;
        const int outstanding = (TOTAL - i0) < R ? (TOTAL - i0) : R;
        const int keep = outstanding - UNR;
        if (i0 > 0) {
            if (keep >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (keep >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (keep >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        u32x4 wv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) wv[u] = *(volatile __attribute__((address_space(3))) u32x4*)((__attribute__((address_space(3))) char*)smem + wave * (R * 1024) + ((i0 + u) % R) * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = (i0 + u) / NP, p = (i0 + u) % NP;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[r] = fmaf(bflo(wv[u][q]), bflo(xv[p][q]), acc[r]);
                acc[r] = fmaf(bfhi(wv[u][q]), bfhi(xv[p][q]), acc[r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (i0 + u + R < TOTAL) issue(i0 + u + R);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum64(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (n0 + r < N) {
                uint32_t u = __float_as_uint(acc[r]);
                y[n0 + r] = (bf16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
            }
    }
}

// the product's register form, reduced to what this probe compares (one matrix, no norm): UN chunks x ROWS rows per lane and trip
template <int ROWS, int UN>
__global__ __launch_bounds__(256) void gemv_reg(const bf16_t* __restrict__ W, const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int K) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * 4 + wave) * ROWS;
    if (n0 >= N) return;
    const int nchunk = K >> 3;
    const bf16_t* wrow[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wrow[r] = W + (size_t)(n0 + r < N ? n0 + r : N - 1) * K;
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    u32x4 wv[ROWS][UN], xv[UN];
    auto issue = [&](int c) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int cu = c + 64 * u < nchunk ? c + 64 * u : c;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) wv[r][u] = __builtin_nontemporal_load((const u32x4*)(wrow[r] + cu * 8));
            xv[u] = *(const u32x4*)(x + cu * 8);
        }
    };
    int c = lane;
    issue(c);
    for (; c < nchunk; c += 64 * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool live = c + 64 * u < nchunk;
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[r] = fmaf(bflo(wv[r][u][q]), live ? bflo(xv[u][q]) : 0.f, acc[r]);
                    acc[r] = fmaf(bfhi(wv[r][u][q]), live ? bfhi(xv[u][q]) : 0.f, acc[r]);
                }
        }
        if (c + 64 * UN < nchunk) issue(c + 64 * UN);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum64(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (n0 + r < N) {
                uint32_t u = __float_as_uint(acc[r]);
                y[n0 + r] = (bf16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
            }
    }
}

static float bf2f_host(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(const bf16_t*, const bf16_t*, bf16_t*, int, int);
struct Cfg { const char* name; kern_t k; int rows; size_t lds; };

int main() {
    struct Shape { const char* name; int N, K; } shapes[] = {
        {"o_proj 4096 x 4096", 4096, 4096}, {"q/k/v as one 12288 x 4096", 12288, 4096}, {"gate+up as one 22016 x 4096", 22016, 4096},
        {"down 4096 x 11008", 4096, 11008}, {"lm_head 32000 x 4096", 32000, 4096}};
    const int COPIES = 8;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto tobf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"shapes\": [\n");
    for (size_t si = 0; si < sizeof(shapes) / sizeof(shapes[0]); ++si) {
        const int N = shapes[si].N, K = shapes[si].K;
        const size_t wn = (size_t)N * K;
        std::vector<uint16_t> hw(wn), hx(K);
        for (auto& v : hw) v = tobf(nd(rng) * 0.02f);
        for (auto& v : hx) v = tobf(nd(rng));
        bf16_t *dW, *dx, *dy;
        CHECK(hipMalloc(&dW, wn * 2 * COPIES + 4096));
        CHECK(hipMalloc(&dx, K * 2));
        CHECK(hipMalloc(&dy, N * 2));
        for (int c = 0; c < COPIES; ++c) CHECK(hipMemcpy(dW + c * wn, hw.data(), wn * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dx, hx.data(), K * 2, hipMemcpyHostToDevice));
        std::vector<double> ref(N);
        for (int n = 0; n < N; n += 97) {  // spot rows
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)bf2f_host(hw[(size_t)n * K + k]) * bf2f_host(hx[k]);
            ref[n] = s;
        }
        std::vector<Cfg> cfgs;
        if (K == 4096) {
            cfgs = {{"registers 2 rows x 4 chunks (product)", gemv_reg<2, 4>, 2, 0}, {"registers 1 row x 8 chunks (product, plain)", gemv_reg<1, 8>, 1, 0},
                    {"DMA ring 2 rows, 16 slots, batches of 4", gemv_dma<2, 8, false, 16, 4>, 2, 4 * 16 * 1024},
                    {"DMA ring 2 rows, 8 slots, batches of 4", gemv_dma<2, 8, false, 8, 4>, 2, 4 * 8 * 1024},
                    {"DMA ring 4 rows, 16 slots, batches of 4", gemv_dma<4, 8, false, 16, 4>, 4, 4 * 16 * 1024},
                    {"DMA ring 1 row, 8 slots, batches of 4", gemv_dma<1, 8, false, 8, 4>, 1, 4 * 8 * 1024}};
        } else {
            cfgs = {{"registers 1 row x 11 chunks (product)", gemv_reg<1, 11>, 1, 0}, {"registers 2 rows x 4 chunks", gemv_reg<2, 4>, 2, 0},
                    {"DMA ring 2 rows, 16 slots, batches of 4", gemv_dma<2, 22, true, 16, 4>, 2, 4 * 16 * 1024},
                    {"DMA ring 1 row, 16 slots, batches of 2", gemv_dma<1, 22, true, 16, 2>, 1, 4 * 16 * 1024}};
        }
        printf(" {\"shape\": \"%s\", \"mb\": %.1f, \"configs\": [\n", shapes[si].name, wn * 2 / 1e6);
        for (auto& c : cfgs)
            if (c.lds) CHECK(hipFuncSetAttribute((const void*)c.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
        std::vector<std::vector<double>> us(cfgs.size());
        std::vector<double> err(cfgs.size(), 0.0);
        for (int rnd = 0; rnd < 4; ++rnd)
            for (size_t ci = 0; ci < cfgs.size(); ++ci) {
                auto& c = cfgs[ci];
                const dim3 grid((N + 4 * c.rows - 1) / (4 * c.rows));
                auto run = [&](int i) { hipLaunchKernelGGL(c.k, grid, dim3(256), c.lds, 0, dW + (size_t)(i % COPIES) * wn, dx, dy, N, K); };
                for (int i = 0; i < 8; ++i) run(i);
                CHECK(hipEventRecord(e0));
                const int L = 64;
                for (int i = 0; i < L; ++i) run(i);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                us[ci].push_back(ms * 1e3 / L);
                if (rnd == 0) {
                    std::vector<uint16_t> hy(N);
                    CHECK(hipMemcpy(hy.data(), dy, N * 2, hipMemcpyDeviceToHost));
                    for (int n = 0; n < N; n += 97) err[ci] = std::max(err[ci], std::fabs((double)bf2f_host(hy[n]) - ref[n]) / (std::fabs(ref[n]) + 1e-2));
                }
            }
        for (size_t ci = 0; ci < cfgs.size(); ++ci) {
            std::sort(us[ci].begin(), us[ci].end());
            const double m = us[ci][us[ci].size() / 2];
            printf("  {\"config\": \"%s\", \"us_median\": %.2f, \"us_min\": %.2f, \"tb_per_s\": %.2f, \"max_rel_err_spot_rows\": %.2e}%s\n", cfgs[ci].name, m, us[ci].front(),
                   wn * 2 / m / 1e6, err[ci], ci + 1 < cfgs.size() ? "," : "");
        }
        printf(" ]}%s\n", si + 1 < sizeof(shapes) / sizeof(shapes[0]) ? "," : "");
        CHECK(hipFree(dW));
        CHECK(hipFree(dx));
        CHECK(hipFree(dy));
    }
    printf("]}\n");
    return 0;
}
