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
    // per wave: stamp differences; s_memtime ticks at 100 MHz? (MI355X: shader clock) -- report raw ticks and shares
    const int ntiles = (L + 63) / 64;
    std::vector<double> seg(40, 0.0);
    long nwaves = 0; double life = 0;
    for (size_t w = 0; w < nblk * nw; ++w) {
        const unsigned long long* s = &hs[w * 32];
        if (!s[0] || !s[31]) continue;
        ++nwaves; life += (double)(s[31] - s[0]);
        seg[0] += (double)(s[1] - s[0]);                    // Q loads + first tile loads ISSUED
        for (int t = 0; t < ntiles; ++t) {
            const unsigned long long prev = t == 0 ? s[1] : s[5 + 4 * (t - 1)];
            seg[1] += (double)(s[2 + 4 * t] - prev);        // barrier 1 (previous compute of other waves)
            seg[2] += (double)(s[3 + 4 * t] - s[2 + 4 * t]);  // wait for the tile's global loads + LDS write
            seg[3] += (double)(s[4 + 4 * t] - s[3 + 4 * t]);  // barrier 2
            seg[4] += (double)(s[5 + 4 * t] - s[4 + 4 * t]);  // next-tile load issue + compute
        }
        seg[5] += (double)(s[30] - s[5 + 4 * (ntiles - 1)]);
        seg[6] += (double)(s[31] - s[30]);                  // epilogue
    }
    const char* names[] = {"prologue issue", "barrier A (wait other waves)", "wait loads + LDS write", "barrier B", "compute (+ next load issue)", "-", "epilogue"};
    printf("waves %ld, mean lifetime %.0f ticks\n", nwaves, life / nwaves);
    for (int i = 0; i < 7; ++i) printf("  %-32s %8.0f ticks  %5.1f %%\n", names[i], seg[i] / nwaves, 100.0 * seg[i] / life);
    return 0;
}
