// Diagnostic build of the eight-phase GEMM with in-kernel stamps (cdna_hip_programming.md section 7, "In-kernel stamps"): where a
// 256 x 256 tile spends its life -- prologue, K-loop, the epilogue's stages, store acknowledgement -- and how long a CU sits
// between one block's last wave leaving and the next block's first instruction. Read SHARES, not lengths (the stamps' fences
// forbid overlaps the real kernel has; the stamped build also waits for its stores before the last stamp). Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probes/gemm_stamps.hip -o /tmp/gemm_stamps && /tmp/gemm_stamps
#define MERV_GEMM_STAMPS 1
#include "gemm_probe_hooks.h"
#include "../../merv_amd/csrc/gemm.hip"
#include "../../merv_amd/csrc/prof.cpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

extern "C" void merv_set_error(const char*) {}

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536;
    struct Shape { const char* name; int N, K, act; bool res; } shapes[] = {
        {"qkv  N=3072 K=1024 bias", 3072, 1024, merv::ACT_NONE, false}, {"proj N=1024 K=1024 bias+residual", 1024, 1024, merv::ACT_NONE, true},
        {"fc1  N=4096 K=1024 bias+erf-GELU", 4096, 1024, merv::ACT_GELU_ERF, false}, {"fc2  N=1024 K=4096 bias+residual", 1024, 4096, merv::ACT_NONE, true}};
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hA((size_t)M * 4096), hW((size_t)4096 * 4096);
    for (auto& x : hA) x = f2b(nd(rng));
    for (auto& x : hW) x = f2b(nd(rng) * 0.03f);
    merv::bf16_t *A, *W, *C, *R;
    float* bias;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * 4096 * 2); hipMalloc(&R, (size_t)M * 1024 * 2);
    hipMalloc(&bias, 4096 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(R, hA.data(), (size_t)M * 1024 * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 4096 * 4);
    for (const Shape& sh : shapes) {
        merv::GemmArgs g;
        memset(&g, 0, sizeof g);
        g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.C = C; g.ldc = sh.N; g.M = M; g.N = sh.N; g.K = sh.K; g.bias = bias; g.act = sh.act;
        if (sh.res) { g.res = R; g.ldres = sh.N; }
        const size_t nblk = (size_t)(M / 256) * (sh.N / 256);
        unsigned long long* st;
        hipMalloc(&st, nblk * 8 * 16 * 8);
        hipMemset(st, 0, nblk * 8 * 16 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &st, sizeof st);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) merv::launch_gemm(g, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) merv::launch_gemm(g, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hs(nblk * 8 * 16);
        hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        printf("\n%s, M = %d: %zu tiles = %.2f rounds, %.1f us per launch (stamped build)\n", sh.name, M, nblk, nblk / 256.0, ms * 200);
        // per wave: mean cycles between consecutive stamps 1..11
        const char* what[16] = {"", "kernel entry", "prologue: 12 DMA pieces per wave issued", "tile 0 early quarters landed, barrier passed", "K-loop done",
                                "epilogue: bias / statistics / residual part 0 requested", "staging barrier passed", "part 0 scaled, activated, staged (LDS)",
                                "part 0 residual added, stores issued", "part 1 staged (incl. wait for its residual rows)", "part 1 stores issued",
                                "all stores acknowledged (vmcnt 0)"};
        double at[16] = {};
        long nw = 0;
        for (size_t w = 0; w < nblk * 8; ++w) {
            const unsigned long long* s = &hs[w * 16];
            if (!s[1] || !s[11]) continue;
            ++nw;
            for (int k = 2; k <= 11; ++k) at[k] += (double)(s[k] - s[1]);
        }
        double prev = 0, total = at[11] / nw;
        for (int k = 2; k <= 11; ++k) {
            const double v = at[k] / nw;
            printf("  %-62s %8.0f cycles  %5.1f %%\n", what[k], v - prev, 100.0 * (v - prev) / total);
            prev = v;
        }
        // block turnover on a CU: gap between a block's last exit and the next block's first entry (100 MHz real-time ticks)
        struct Blk { unsigned long long in, out; };
        std::map<unsigned long long, std::vector<Blk>> cus;
        for (size_t b = 0; b < nblk; ++b) {
            unsigned long long in = ~0ull, out = 0, id = 0;
            for (int w = 0; w < 8; ++w) {
                const unsigned long long* s = &hs[(b * 8 + w) * 16];
                in = std::min(in, s[0]); out = std::max(out, s[12]);
                if (w == 0) id = ((s[14] >> 32) & 0xf) << 16 | (s[14] & 0xff00);  // xcc | se, sh, cu
            }
            cus[id].push_back({in, out});
        }
        double gap = 0, life = 0; long ng = 0, nl = 0;
        for (auto& kv : cus) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Blk& a, const Blk& b) { return a.in < b.in; });
            for (size_t i = 0; i < v.size(); ++i) {
                life += (double)(v[i].out - v[i].in); ++nl;
                if (i + 1 < v.size() && v[i + 1].in >= v[i].out) { gap += (double)(v[i + 1].in - v[i].out); ++ng; }
            }
        }
        printf("  CUs seen %zu; block life %.2f us; gap between a block's last exit and the next block's first instruction on the same CU %.2f us (%ld gaps)\n",
               cus.size(), life / nl / 100.0, ng ? gap / ng / 100.0 : 0.0, ng);
        printf("  wave lifetime %.0f cycles = %.2f us of block life => shader clock ~%.2f GHz\n", total, life / nl / 100.0, total / (life / nl / 100.0) / 1e3);
        hipFree(st);
    }
    return 0;
}
