// Probe: which physical CUs (XCC, SE, CU) run the blocks of a kernel launched on a stream created with a CU mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void who(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    int ncu = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    const int NB = 4096;
    unsigned* d; CK(hipMalloc(&d, NB * 8));
    std::vector<unsigned> h(NB * 2);
    // masks to try: argv[1] = hex words comma separated, else a set of presets
    std::vector<std::vector<uint32_t>> masks;
    std::vector<const char*> names;
    auto words = (ncu + 31) / 32;
    { std::vector<uint32_t> m(words, 0xffffffffu); masks.push_back(m); names.push_back("all"); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xffffffffu; masks.push_back(m); names.push_back("bits 0-31"); }
    { std::vector<uint32_t> m(words, 0); m[1] = 0xffffffffu; masks.push_back(m); names.push_back("bits 32-63"); }
    { std::vector<uint32_t> m(words, 0); for (int w = 0; w < words / 2; ++w) m[w] = 0xffffffffu; masks.push_back(m); names.push_back("lower half of the bits"); }
    { std::vector<uint32_t> m(words, 0x0000ffffu); masks.push_back(m); names.push_back("low 16 of every 32"); }
    { std::vector<uint32_t> m(words, 0x55555555u); masks.push_back(m); names.push_back("even bits"); }
    { std::vector<uint32_t> m(words, 0x000000ffu); masks.push_back(m); names.push_back("low 8 of every 32"); }
    for (size_t mi = 0; mi < masks.size(); ++mi) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)masks[mi].size(), masks[mi].data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", names[mi], hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(d, 0xff, NB * 8, s));
        hipLaunchKernelGGL(who, dim3(NB), dim3(64), 0, s, d, 20000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int b = 0; b < NB; ++b) {
            unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        size_t tot = 0;
        printf("%-26s:", names[mi]);
        for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
        printf("  total distinct CUs %zu\n", tot);
        if (mi == 1 || mi == 4) {
            for (auto& kv : per_xcc) { printf("   xcc%u:", kv.first); for (auto c : kv.second) printf(" se%u.sh%u.cu%u", c >> 8, (c >> 4) & 1, c & 0xf); printf("\n"); }
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
