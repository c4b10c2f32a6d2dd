// Where the eight-phase K-loop's energy goes (gfx950): the loop rebuilt ingredient by ingredient around the same 64 MFMAs per wave and
// K-tile (v_mfma_f32_16x16x32_bf16, product issue order, random operands, 8 waves per CU on every CU, power-capped chip -> wall time per
// MFMA is energy per MFMA; in-kernel clock printed beside it):
//   0  bare: operands stay in registers
//   1  + the wave's 24 ds_read_b128 fragment reads per K-tile (12 / 4 / 8 / 0 per phase, as gemm.hip), waited for before the phase's MFMAs
//   2  + the phase barriers (two per phase), the one-barrier stagger of waves 4-7 and s_setprio around the MFMAs
//   3  + 8 LDS-DMA pieces (1 KiB each) per wave and K-tile, two per phase, from a footprint of `fp_mb` MB (counted vmcnt(8) per K-tile;
//        the pieces land in a region nobody reads): fp 2 = every XCD's L2 holds it, 128 = Infinity Cache, 4096 = HBM
// usage: kloop_energy [warm_seconds=2] -> JSON on stdout
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int OPS_LDS = 128 * 1024, DUMP_LDS = 16 * 1024;

// gemm.hip quad_order 4: position x -> kk << 3 | ii << 2 | jj
constexpr int quad_order4(int x) {
    const int kk = x >> 3, jj = (x >> 1) & 3, ii = (x & 1) ^ (jj & 1);
    return kk << 3 | ii << 2 | jj;
}

#define DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int MODE>
__global__ __launch_bounds__(512) void kloop(const bf16x8* __restrict__ ops, const char* __restrict__ stream, size_t fp_mask, float* __restrict__ sink,
                                             unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2;
    // the block's 128 KB operand image: random fragments (every thread copies 16 x 16 B)
    for (int c = threadIdx.x; c < OPS_LDS / 16; c += 512) *(bf16x8*)(smem + c * 16) = ops[((size_t)blockIdx.x * (OPS_LDS / 16) + c) & 0xFFFFF];
    __syncthreads();
    bf16x8 X[2][2][2], Y[4][2];  // W fragments [nh][ii][kk], A fragments of the current 64-row half [jj][kk]
    // two read bases per wave (fragments 0..11 from the first, 12..23 from the second: offsets < 64 KB)
    const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + ((wave * 24 * 1024) & (OPS_LDS - 1)) + lane * 16;
    const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (((wave * 24 + 12) * 1024) & (OPS_LDS - 1)) + lane * 16;
    auto rd_x = [&](int nh) {  // 4 reads
        if (nh == 0) { DS_READ(X[0][0][0], a0, 0); DS_READ(X[0][0][1], a0, 1024); DS_READ(X[0][1][0], a0, 2048); DS_READ(X[0][1][1], a0, 3072); }
        else { DS_READ(X[1][0][0], a0, 4096); DS_READ(X[1][0][1], a0, 5120); DS_READ(X[1][1][0], a0, 6144); DS_READ(X[1][1][1], a0, 7168); }
    };
    auto rd_y = [&](int mh) {  // 8 reads
        if (mh == 0) {
            DS_READ(Y[0][0], a0, 8192); DS_READ(Y[0][1], a0, 9216); DS_READ(Y[1][0], a0, 10240); DS_READ(Y[1][1], a0, 11264);
            DS_READ(Y[2][0], a1, 0); DS_READ(Y[2][1], a1, 1024); DS_READ(Y[3][0], a1, 2048); DS_READ(Y[3][1], a1, 3072);
        } else {
            DS_READ(Y[0][0], a1, 4096); DS_READ(Y[0][1], a1, 5120); DS_READ(Y[1][0], a1, 6144); DS_READ(Y[1][1], a1, 7168);
            DS_READ(Y[2][0], a1, 8192); DS_READ(Y[2][1], a1, 9216); DS_READ(Y[3][0], a1, 10240); DS_READ(Y[3][1], a1, 11264);
        }
    };
    rd_x(0); rd_x(1); rd_y(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto quadrant = [&](int mh, int nh) {
        if constexpr (MODE >= 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const int c = quad_order4(x), kk = c >> 3, ii = (c >> 2) & 1, jj = c & 3;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nh * 2 + ii][mh * 4 + jj]) : "v"(X[nh][ii][kk]), "v"(Y[jj][kk]));
        }
        if constexpr (MODE >= 2) __builtin_amdgcn_s_setprio(0);
    };
    // the wave's stream of DMA pieces: 1 KiB each, consecutive pieces 8 KB apart (a row of a K = 4096 operand), wrapped into the footprint
    size_t soff = ((size_t)(blockIdx.x * 8 + wave) * 1048576 * 3 + lane * 16) & fp_mask;
    const unsigned dump = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + OPS_LDS + wave * 2048;
    auto dma2 = [&](int ph) {
        if constexpr (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + soff),
                                                 (__attribute__((address_space(3))) void*)(smem + OPS_LDS + wave * 2048 + u * 1024), 16, 0, 0);
                soff = (soff + 8192 + 1024 * 17) & fp_mask;
            }
        }
    };
    (void)dump;
    auto loaded = [&]() {
        if constexpr (MODE >= 2) asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto done = [&]() {
        if constexpr (MODE >= 2) asm volatile("s_barrier" ::: "memory");
    };
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
    if constexpr (MODE >= 2) { if (wr == 1) asm volatile("s_barrier" ::: "memory"); }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        // phase 1
        if constexpr (MODE >= 1) { rd_x(0); rd_y(0); }
        dma2(0);
        loaded(); quadrant(0, 0); done();
        // phase 2
        if constexpr (MODE >= 1) rd_x(1);
        dma2(1);
        loaded(); quadrant(0, 1); done();
        // phase 3
        if constexpr (MODE >= 1) rd_y(1);
        dma2(2);
        loaded(); quadrant(1, 1); done();
        // phase 4
        dma2(3);
        if constexpr (MODE >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        loaded(); quadrant(1, 0); done();
    }
    if constexpr (MODE >= 2) { if (wr == 0) asm volatile("s_barrier" ::: "memory"); }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) {
        stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0;
    }
}

typedef void (*kern_t)(const bf16x8*, const char*, size_t, float*, unsigned long long*, int);
struct Variant { const char* name; kern_t k; size_t fp_mb; };

int main(int argc, char** argv) {
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t nops = 1 << 20;  // 16 MB of random fragments
    std::vector<uint16_t> h(nops * 8);
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) {
        float f = nd(rng);
        uint32_t u;
        memcpy(&u, &f, 4);
        v = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    bf16x8* d_ops;
    char* d_stream;
    float* d_sink;
    unsigned long long* d_st;
    const size_t stream_bytes = (size_t)4096 << 20;
    CHECK(hipMalloc(&d_ops, nops * 16));
    CHECK(hipMalloc(&d_stream, stream_bytes));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 512 * 4));
    CHECK(hipMalloc(&d_st, (size_t)cus * 8 * 2 * 8));
    CHECK(hipMemcpy(d_ops, h.data(), nops * 16, hipMemcpyHostToDevice));
    for (size_t o = 0; o < stream_bytes; o += nops * 16) CHECK(hipMemcpy(d_stream + o, d_ops, nops * 16, hipMemcpyDeviceToDevice));  // random bytes everywhere
    Variant V[] = {
        {"0 bare MFMA loop (operands in registers)", kloop<0>, 0},
        {"1 + 24 ds_read_b128 per K-tile", kloop<1>, 0},
        {"2 + phase barriers, stagger, setprio", kloop<2>, 0},
        {"3 + 8 LDS-DMA pieces per wave and K-tile, 2 MB footprint (L2)", kloop<3>, 2},
        {"3 + DMA pieces, 128 MB footprint (Infinity Cache)", kloop<3>, 128},
        {"3 + DMA pieces, 4096 MB footprint (HBM)", kloop<3>, 4096},
    };
    const int NV = sizeof(V) / sizeof(V[0]), iters = 4000;
    for (auto& v : V) CHECK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, OPS_LDS + DUMP_LDS));
    const double flop = 2.0 * 16 * 16 * 32 * 64.0 * iters * 8.0 * cus;
    auto launch = [&](const Variant& v) {
        const size_t mask = v.fp_mb ? ((v.fp_mb << 20) - 1) & ~(size_t)15 : 0;
        hipLaunchKernelGGL(v.k, dim3(cus), dim3(512), OPS_LDS + DUMP_LDS, 0, d_ops, d_stream, mask, d_sink, d_st, iters);
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    {
        CHECK(hipEventRecord(e0));
        double el = 0;
        while (el < warm_s * 1e3) {
            for (int r = 0; r < 20; ++r) launch(V[2]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            el = ms;
        }
    }
    std::vector<std::vector<double>> tf(NV), ghz(NV), us(NV);
    std::vector<unsigned long long> st((size_t)cus * 16);
    for (int rnd = 0; rnd < 5; ++rnd)
        for (int k = 0; k < NV; ++k) {
            const int L = 60;
            for (int r = 0; r < 10; ++r) launch(V[k]);
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < L; ++r) launch(V[k]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            tf[k].push_back(flop * L / (ms * 1e-3) / 1e12);
            us[k].push_back(ms * 1e3 / L / iters);
            CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> c;
            for (int w = 0; w < cus * 8; ++w) c.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);
            std::sort(c.begin(), c.end());
            ghz[k].push_back(c[c.size() / 2]);
        }
    printf("{\"cus\": %d, \"k_tiles_per_launch\": %d, \"variants\": [\n", cus, iters);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double base = med(tf[0]);
    for (int k = 0; k < NV; ++k) {
        const double t = med(tf[k]), g = med(ghz[k]), u = med(us[k]);
        printf(" {\"variant\": \"%s\", \"tflops_median\": %.1f, \"us_per_k_tile\": %.4f, \"clock_ghz\": %.3f, \"cycles_per_k_tile\": %.0f, \"vs_bare\": %.3f}%s\n", V[k].name, t, u, g,
               u * g * 1e3, t / base, k + 1 < NV ? "," : "");
    }
    printf("]}\n");
    return 0;
}
