// Diagnostic build of the attention kernel with in-kernel s_memtime stamps (cdna_hip_programming.md section 7,
// "In-kernel stamps"): where a block of the short-sequence shapes spends its lifetime. Read SHARES, not lengths
// (the stamps' fences forbid overlaps the real kernel has). Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/probes/attn_stamps.hip -o /tmp/attn_stamps && /tmp/attn_stamps 128 257 16
#define MERV_ATTN_STAMPS 1
#include "../../merv_amd/csrc/attention.hip"
#include "../../merv_amd/csrc/prof.cpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" void merv_set_error(const char*) {}

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 257, heads = argc > 3 ? atoi(argv[3]) : 16;
    const int D = heads * 64;
    const size_t rows = (size_t)nseq * L;
    std::vector<uint16_t> h(rows * 3 * D);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.5f);
    for (auto& x : h) { float f = nd(rng); uint32_t u; memcpy(&u, &f, 4); x = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    merv::bf16_t *qkv, *out;
    hipMalloc(&qkv, h.size() * 2); hipMalloc(&out, rows * D * 2);
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int t32 = (L + 31) / 32;
    const int nw = 4;  // upper bound on waves per block for the buffer
    const size_t nblk = (size_t)nseq * heads * 16;
    unsigned long long* st;
    hipMalloc(&st, nblk * nw * 32 * 8);
    hipMemset(st, 0, nblk * nw * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &st, sizeof st);
    merv::AttnArgs a{};
    a.qkv = qkv; a.out = out; a.nseq = nseq; a.L = L; a.heads = heads; a.D = D; a.scale = 0.125f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) merv::launch_attention(a, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) merv::launch_attention(a, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("nseq %d L %d heads %d t32 %d: %.1f us per launch (stamped build)\n", nseq, L, heads, t32, ms * 100);
    std::vector<unsigned long long> hs(nblk * nw * 32);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    // per wave: mean offset of every stamp from stamp 0 (stamps a kernel form does not execute stay 0 and are skipped)
    std::vector<double> at(32, 0.0);
    std::vector<long> cnt(32, 0);
    long nwaves = 0;
    for (size_t w = 0; w < nblk * nw; ++w) {
        const unsigned long long* s = &hs[w * 32];
        if (!s[0] || !s[31]) continue;
        ++nwaves;
        for (int k = 1; k < 32; ++k)
            if (s[k]) { at[k] += (double)(s[k] - s[0]); ++cnt[k]; }
    }
    printf("waves %ld; stamp: mean cycles since kernel entry (delta to the previous executed stamp)\n", nwaves);
    const char* what[32] = {};
    what[1] = "loads issued (streamed) / all K,V landed + barrier (resident)";
    what[30] = "key-tile loop done"; what[31] = "outputs stored";
    double prev = 0;
    for (int k = 1; k < 32; ++k) {
        if (!cnt[k]) continue;
        const double v = at[k] / cnt[k];
        char buf[64] = "";
        if (k >= 2 && k < 30) snprintf(buf, sizeof buf, "tile %d: %s", (k - 2) / 4, ((k - 2) % 4 == 0) ? "barrier A passed" : ((k - 2) % 4 == 1) ? "tile written to LDS" : ((k - 2) % 4 == 2) ? "barrier B passed" : "computed");
        printf("  stamp %2d  %9.0f  (+%7.0f)  %s\n", k, v, v - prev, what[k] ? what[k] : buf);
        prev = v;
    }
    return 0;
}
