// Energy per MFMA against the ISSUE ORDER of a wave's MFMAs (gfx950): a bare v_mfma_f32_16x16x32_bf16 loop on random operands held in
// registers -- the eight-phase GEMM's wave tile (4 first-operand "X" fragments = W rows, 8 second-operand "Y" fragments = activation
// rows, two k-halves, 32 accumulators; 64 MFMAs per K-tile) -- 8 waves per CU on every CU, nothing else in the loop. Every order
// issues the same 64 MFMAs per iteration (same products, same sums per accumulator: an accumulator takes k-half 0 before k-half 1 and
// returns after >= 8 other MFMAs); what differs is which operand register an MFMA shares with its predecessor. The chip is power-capped
// under this loop, so wall time per MFMA is energy per MFMA (MI355X_MICROARCH.md "DVFS give-back"); the in-kernel clock
// (s_memtime / s_memrealtime) is printed beside it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_hold.hip -o gpurun_out/mfma_hold && gpurun_out/mfma_hold [seconds_warm=2]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

// position p (0..63) of order ORD -> kk << 5 | i << 3 | j   (i: X fragment 0..3, j: Y fragment 0..7)
constexpr int order_at(int ord, int p) {
    int kk = p >> 5, q = p & 31, i = 0, j = 0;
    switch (ord) {
        case 0: i = q >> 3; j = q & 7; break;                                   // X outer: X held 8, Y changes every MFMA
        case 1: j = q >> 2; i = q & 3; break;                                   // Y outer: Y held 4, X changes every MFMA
        case 2: j = q >> 2; i = (j & 1) ? 3 - (q & 3) : (q & 3); break;         // Y outer, X snake: every MFMA shares an operand
        case 3: i = q >> 3; j = (i & 1) ? 7 - (q & 7) : (q & 7); break;         // X outer, Y snake
        case 4: {                                                               // the product's phases: quadrant (4 Y x 2 X), k-half, X, Y
            const int quad = p >> 4, r = p & 15, mh = (quad == 2 || quad == 3), nh = (quad == 1 || quad == 2);
            kk = r >> 3; i = nh * 2 + ((r >> 2) & 1); j = mh * 4 + (r & 3);
            break;
        }
        case 5: {                                                               // the product's phases with k-half, Y, X snake (gemm.hip order 4)
            const int quad = p >> 4, r = p & 15, mh = (quad == 2 || quad == 3), nh = (quad == 1 || quad == 2);
            const int jj = (r >> 1) & 3;
            kk = r >> 3; j = mh * 4 + jj; i = nh * 2 + ((r & 1) ^ (jj & 1));
            break;
        }
        case 6: {                                                               // row-quarter phases (2 Y x 4 X): k-half, Y, X snake -- Y held 4
            const int quad = p >> 4, r = p & 15, jj = (r >> 2) & 1;
            kk = r >> 3; j = quad * 2 + jj; i = jj ? 3 - (r & 3) : (r & 3);
            break;
        }
        case 7: {                                                               // row-quarter phases: k-half, X, Y -- X held 2
            const int quad = p >> 4, r = p & 15;
            kk = r >> 3; i = (r >> 1) & 3; j = quad * 2 + (r & 1);
            break;
        }
        default: i = q & 3; j = ((q & 3) + (q >> 2)) & 7; break;                // 8: diagonal -- no operand shared with the predecessor, none held
    }
    return kk << 5 | i << 3 | j;
}

template <int ORD>
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ ops, float* __restrict__ sink, unsigned long long* stamps, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 X[2][4], Y[2][8];
    // every wave and lane its own random fragments (24 per lane)
    const bf16x8* src = ops + ((size_t)(blockIdx.x * 8 + wave) * 24) * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) X[kk][i] = src[(kk * 12 + i) * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) Y[kk][j] = src[(kk * 12 + 4 + j) * 64];
    }
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            constexpr int dummy = 0; (void)dummy;
            const int c = order_at(ORD, p), kk = c >> 5, i = (c >> 3) & 3, j = c & 7;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(X[kk][i]), "v"(Y[kk][j]));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;  // (order-independent up to nothing: every accumulator sees the same sequence)
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

typedef void (*kern_t)(const bf16x8*, float*, unsigned long long*, int);
static const char* NAMES[] = {
    "0 X outer (X held 8, Y new every MFMA)",
    "1 Y outer (Y held 4, X new every MFMA)",
    "2 Y outer, X snake (Y held 4, every MFMA shares an operand)",
    "3 X outer, Y snake (X held 8, every MFMA shares an operand)",
    "4 product phases: 4Y x 2X quadrants, k-half / X / Y (X held 4)",
    "5 product phases, k-half / Y / X snake (gemm.hip order 4: Y held 2)",
    "6 row-quarter phases 2Y x 4X, k-half / Y / X snake (Y held 4)",
    "7 row-quarter phases 2Y x 4X, k-half / X / Y (X held 2)",
    "8 diagonal (nothing shared, nothing held)",
};

int main(int argc, char** argv) {
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;  // 1: all-zero operands (the clock-unconstrained reference)
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t nfrag = (size_t)cus * 8 * 24 * 64;
    std::vector<uint16_t> h(nfrag * 8);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) {
        float f = zero ? 0.f : nd(rng);
        uint32_t u;
        memcpy(&u, &f, 4);
        v = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    bf16x8* d_ops;
    float* d_sink;
    unsigned long long* d_st;
    CHECK(hipMalloc(&d_ops, nfrag * 16));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 512 * 4));
    CHECK(hipMalloc(&d_st, (size_t)cus * 8 * 2 * 8));
    CHECK(hipMemcpy(d_ops, h.data(), nfrag * 16, hipMemcpyHostToDevice));
    kern_t K[9] = {mfma_loop<0>, mfma_loop<1>, mfma_loop<2>, mfma_loop<3>, mfma_loop<4>, mfma_loop<5>, mfma_loop<6>, mfma_loop<7>, mfma_loop<8>};
    const int NK = 9, iters = 4000;  // 4000 x 64 MFMAs per wave ~ 2 ms per launch
    const double flop = 2.0 * 16 * 16 * 32 * 64.0 * iters * 8.0 * cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // warm: back-to-back launches of the product order until the clock has settled
    {
        CHECK(hipEventRecord(e0));
        double el = 0;
        while (el < warm_s * 1e3) {
            for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(K[4], dim3(cus), dim3(512), 0, 0, d_ops, d_sink, d_st, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            el = ms;
        }
    }
    std::vector<std::vector<double>> tf(NK), ghz(NK);
    std::vector<unsigned long long> st((size_t)cus * 16);
    for (int rnd = 0; rnd < 5; ++rnd)
        for (int k = 0; k < NK; ++k) {
            const int L = 100;  // ~0.2 s per variant and round
            for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(K[k], dim3(cus), dim3(512), 0, 0, d_ops, d_sink, d_st, iters);
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < L; ++r) hipLaunchKernelGGL(K[k], dim3(cus), dim3(512), 0, 0, d_ops, d_sink, d_st, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            tf[k].push_back(flop * L / (ms * 1e-3) / 1e12);
            CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> c;
            for (int w = 0; w < cus * 8; ++w) c.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);  // ticks per 10 ns -> GHz
            std::sort(c.begin(), c.end());
            ghz[k].push_back(c[c.size() / 2]);
        }
    printf("{\"operands\": \"%s\", \"cus\": %d, \"mfma_per_wave_per_launch\": %d, \"orders\": [\n", zero ? "zero" : "random normal", cus, iters * 64);
    double base = 0;
    for (int k = 0; k < NK; ++k) {
        auto t = tf[k], g = ghz[k];
        std::sort(t.begin(), t.end());
        std::sort(g.begin(), g.end());
        if (k == 4) base = t[t.size() / 2];
    }
    for (int k = 0; k < NK; ++k) {
        auto t = tf[k], g = ghz[k];
        std::sort(t.begin(), t.end());
        std::sort(g.begin(), g.end());
        printf(" {\"order\": \"%s\", \"tflops_median\": %.1f, \"tflops_min\": %.1f, \"tflops_max\": %.1f, \"clock_ghz_median\": %.3f, \"vs_product_order_pct\": %.2f}%s\n",
               NAMES[k], t[t.size() / 2], t.front(), t.back(), g[g.size() / 2], (t[t.size() / 2] / base - 1) * 100, k + 1 < NK ? "," : "");
    }
    printf("]}\n");
    return 0;
}
