// Bounding probe for the "operands served from L2" energy lever and the in-kernel clock per GEMM class (1c).
//
// The eight-phase GEMM is power-bound (EXPERIMENTS.md section 1: zero operands +22..28 %, half the CUs 1.31 x per CU), and the guide
// ranks "streamed data served from L2 rather than from beyond it" as the largest clock lever (cdna_hip_programming.md 5.4 rule 28).
// The product kernel moves 1.66 x its algorithmic bytes across the fabric. This probe measures the CEILING of removing that: the
// real encoder shapes on random data, with the operand ROWS of the DMA pieces wrapped into an L2-resident footprint
// (MERV_ABL_WRAPOPS: g_probe_wrap_a / g_probe_wrap_w bytes; the same wrapped rows on every XCD, so each L2 keeps its own copy) --
// same MFMA count, same LDS traffic, same outputs / residual streams, operand fabric traffic -> ~0 -- against the product mapping,
// interleaved in ONE process after >= 2 s of back-to-back launches. Reported per (shape, mode): wall time per launch (HIP events) and
// the in-kernel clock of the eight-phase blocks (delta s_memtime / delta s_memrealtime x 100 MHz, entry / exit stamps only: the
// kernel's own overlaps stay intact; MI355X_MICROARCH.md "DVFS give-back" item 6), median over the blocks of the last launch.
// Results of the wrapped modes are WRONG by design. Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imerv_amd/csrc tools/probes/gemm_energy_bound.hip -o /tmp/geb && /tmp/geb > out.json
#define MERV_GEMM_STAMPS_LIGHT 1
#define MERV_ABL_WRAPOPS 1
#define MERV_ABL_ABLOCKED 1
#include "gemm_probe_hooks.h"
#include "../../merv_amd/csrc/gemm.hip"
#include "../../merv_amd/csrc/prof.cpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

extern "C" void merv_set_error(const char*) {}

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

struct Shape { const char* name; const char* cls; int M, N, K, act; bool res, fold, stats; };
struct Mode { const char* name; int wrap_a, wrap_w, a_blocked; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const double chunk_s = argc > 2 ? atof(argv[2]) : 0.3;
    // rows of 16 videos: LanguageBind 16 x 16 x 257, ViViT 16 x 3137 (what bench.py's default step launches)
    const int MLB = 65792, MVV = 50192;
    const Shape shapes[] = {
        {"languagebind qkv  N=3072 K=1024 folded LayerNorm", "gemm eight-phase, no activation", MLB, 3072, 1024, merv::ACT_NONE, false, true, false},
        {"languagebind proj N=1024 K=1024 bias+residual+statistics", "gemm eight-phase, no activation", MLB, 1024, 1024, merv::ACT_NONE, true, false, true},
        {"languagebind fc1  N=4096 K=1024 folded LayerNorm + quick-GELU", "gemm eight-phase + activation epilogue", MLB, 4096, 1024, merv::ACT_QUICK_GELU, false, true, false},
        {"languagebind fc2  N=1024 K=4096 bias+residual+statistics", "gemm eight-phase, no activation", MLB, 1024, 4096, merv::ACT_NONE, true, false, true},
        {"vivit qkv  N=2304 K=768 folded LayerNorm", "gemm eight-phase, no activation", MVV, 2304, 768, merv::ACT_NONE, false, true, false},
        {"vivit fc1  N=3072 K=768 folded LayerNorm + tanh-GELU", "gemm eight-phase + activation epilogue", MVV, 3072, 768, merv::ACT_GELU_TANH, false, true, false},
        {"vivit fc2  N=768 K=3072 bias+residual+statistics", "gemm eight-phase, no activation", MVV, 768, 3072, merv::ACT_NONE, true, false, true},
    };
    const Mode modes[] = {{"product mapping", 0, 0, 0}, {"W rows wrapped into 1 MB", 0, 1 << 20, 0}, {"A rows wrapped into 1.5 MB", 3 << 19, 0, 0},
                          {"A and W wrapped (1.5 + 1 MB per L2)", 3 << 19, 1 << 20, 0},
                          {"A read K-tile-blocked ([m-tile][K-tile][256 rows x 128 B]: 32 KB contiguous per piece set)", 0, 0, 1}};
    const int NM = sizeof(modes) / sizeof(modes[0]);
    const size_t MMAX = MLB;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hA(MMAX * 4096), hW((size_t)4096 * 4096);
    for (auto& x : hA) x = f2b(nd(rng));
    for (auto& x : hW) x = f2b(nd(rng) * 0.03f);
    merv::bf16_t *A, *W, *C, *R;
    float *bias, *colsum, *rstats, *stats_out;
    hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, MMAX * 4096 * 2); hipMalloc(&R, MMAX * 1024 * 2);
    hipMalloc(&bias, 4096 * 4); hipMalloc(&colsum, 4096 * 4); hipMalloc(&rstats, MMAX * 2 * 4); hipMalloc(&stats_out, (size_t)16 * MMAX * 2 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(R, hA.data(), MMAX * 1024 * 2, hipMemcpyHostToDevice);
    {
        std::vector<float> b(4096), rs(MMAX * 2);
        for (auto& x : b) x = nd(rng) * 0.1f;
        hipMemcpy(bias, b.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (auto& x : b) x = nd(rng);
        hipMemcpy(colsum, b.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (size_t i = 0; i < MMAX; ++i) { rs[2 * i] = 1.f + 0.1f * nd(rng); rs[2 * i + 1] = 0.05f * nd(rng); }
        hipMemcpy(rstats, rs.data(), rs.size() * 4, hipMemcpyHostToDevice);
    }
    const size_t max_blk = (MMAX / 256) * 16;
    unsigned long long* st;
    hipMalloc(&st, max_blk * 8 * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &st, sizeof st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto set_mode = [&](const Mode& m) {
        hipDeviceSynchronize();
        hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wrap_a), &m.wrap_a, sizeof(int));
        hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wrap_w), &m.wrap_w, sizeof(int));
        hipMemcpyToSymbol(HIP_SYMBOL(g_probe_a_blocked), &m.a_blocked, sizeof(int));
        hipDeviceSynchronize();
    };
    printf("{\"what\": \"eight-phase GEMM, operand rows wrapped into an L2-resident footprint against the product mapping: wall time per launch and in-kernel clock\",\n");
    printf(" \"method\": \"tools/probes/gemm_energy_bound.hip: one process, %d interleaved rounds of >= %.2f s of back-to-back launches per (shape, mode) after a 2 s warm-up; "
           "wall = HIP events over the chunk / launches (includes the remainder-row launch of the product's plan); clock = delta s_memtime / delta s_memrealtime x 100 MHz per "
           "eight-phase block of the chunk's last launch (entry / exit stamps only), median over blocks; medians over rounds; random normal operands\",\n \"shapes\": [\n", rounds, chunk_s);
    bool first_shape = true;
    for (const Shape& sh : shapes) {
        merv::GemmArgs g;
        memset(&g, 0, sizeof g);
        g.A = A; g.lda = sh.K; g.W = W; g.ldw = sh.K; g.C = C; g.ldc = sh.N; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.bias = bias; g.act = sh.act;
        if (sh.res) { g.res = R; g.ldres = sh.N; }
        if (sh.fold) { g.row_stats = rstats; g.ln_colsum = colsum; }
        if (sh.stats) g.stats_out = stats_out;
        const int rows8 = merv::plan_split(g);
        const size_t nblk = (size_t)(rows8 / 256) * (sh.N / 256);
        // launches per chunk from a first timing
        set_mode(modes[0]);
        if (hipError_t e = merv::launch_gemm(g, 0); e != hipSuccess) { fprintf(stderr, "%s: launch_gemm failed: %s\n", sh.name, hipGetErrorString(e)); return 1; }
        for (int i = 0; i < 5; ++i) merv::launch_gemm(g, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) merv::launch_gemm(g, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const int L = std::max(20, (int)(chunk_s * 1e3 / (ms / 20)));
        for (int i = 0; i < (int)(2.0e3 / (ms / 20)); ++i) merv::launch_gemm(g, 0);  // >= 2 s of back-to-back launches on random data
        hipDeviceSynchronize();
        std::vector<double> wall[NM], clk[NM], life[NM], clk_k[NM], clk_e[NM], kshare[NM];
        std::vector<unsigned long long> hs(nblk * 8 * 16);
        for (int r = 0; r < rounds; ++r)
            for (int mi = 0; mi < NM; ++mi) {
                const int m = (mi + r) % NM;  // rotate the order from round to round
                set_mode(modes[m]);
                for (int i = 0; i < 10; ++i) merv::launch_gemm(g, 0);
                hipEventRecord(e0, 0);
                for (int i = 0; i < L; ++i) merv::launch_gemm(g, 0);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                wall[m].push_back(ms * 1e3 / L);
                hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> c, lf, ck, ce, ks;
                for (size_t b = 0; b < nblk; ++b) {
                    unsigned long long c0 = ~0ull, c1 = 0, r0 = ~0ull, r1 = 0;
                    for (int w = 0; w < 8; ++w) {
                        const unsigned long long* s = &hs[(b * 8 + w) * 16];
                        c0 = std::min(c0, s[1]); c1 = std::max(c1, s[11]); r0 = std::min(r0, s[0]); r1 = std::max(r1, s[12]);
                    }
                    if (r1 > r0 && c1 > c0) { c.push_back((double)(c1 - c0) / (double)(r1 - r0) * 0.1); lf.push_back((double)(r1 - r0) / 100.0); }
                    // per phase, wave 0 of the block: K-loop (entry .. K-loop end) and epilogue (K-loop end .. exit)
                    const unsigned long long* s = &hs[(b * 8) * 16];
                    if (s[13] > s[0] && s[12] > s[13] && s[4] > s[1] && s[11] > s[4]) {
                        ck.push_back((double)(s[4] - s[1]) / (double)(s[13] - s[0]) * 0.1);
                        ce.push_back((double)(s[11] - s[4]) / (double)(s[12] - s[13]) * 0.1);
                        ks.push_back((double)(s[13] - s[0]) / (double)(s[12] - s[0]));
                    }
                }
                clk[m].push_back(median(c));
                life[m].push_back(median(lf));
                clk_k[m].push_back(median(ck)); clk_e[m].push_back(median(ce)); kshare[m].push_back(median(ks));
            }
        const double flop = 2.0 * sh.M * sh.N * sh.K;
        printf("%s  {\"shape\": \"%s\", \"class\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"eight_phase_rows\": %d, \"eight_phase_tiles\": %zu, \"launches_per_chunk\": %d, \"modes\": [\n",
               first_shape ? "" : ",\n", sh.name, sh.cls, sh.M, sh.N, sh.K, rows8, nblk, L);
        first_shape = false;
        const double base = median(wall[0]);
        for (int m = 0; m < NM; ++m) {
            const double w = median(wall[m]);
            printf("    {\"mode\": \"%s\", \"wrap_a_bytes\": %d, \"wrap_w_bytes\": %d, \"a_blocked\": %d, \"us_per_launch_median\": %.2f, \"us_per_launch_min\": %.2f, \"tflops\": %.1f, "
                   "\"clock_ghz\": %.3f, \"clock_ghz_kloop\": %.3f, \"clock_ghz_epilogue\": %.3f, \"kloop_share_of_block_life\": %.3f, \"block_life_us\": %.2f, \"speedup_vs_product\": %.4f}%s\n",
                   modes[m].name, modes[m].wrap_a, modes[m].wrap_w, modes[m].a_blocked, w, *std::min_element(wall[m].begin(), wall[m].end()), flop / w / 1e6, median(clk[m]),
                   median(clk_k[m]), median(clk_e[m]), median(kshare[m]), median(life[m]), base / w, m + 1 < NM ? "," : "");
        }
        printf("  ]}");
        fflush(stdout);
    }
    printf("\n]}\n");
    return 0;
}
