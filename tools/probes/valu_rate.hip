// VALU issue-rate probe (gfx950): cycles per wave64 instruction for the instruction kinds the GEMM / attention epilogues are
// made of, at one and two waves per SIMD (256 / 512-thread blocks, one block per CU). Each wave runs REP x 16 independent
// instructions of one kind between two s_memtime stamps.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define K16(INSTR)                                                                                                     \
    asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) INSTR(8) INSTR(9) INSTR(10)   \
                     INSTR(11) INSTR(12) INSTR(13) INSTR(14) INSTR(15)                                                 \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),     \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(b), "v"(c))
// operand %N = a[N] (64-bit pairs), %16 = b, %17 = c
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %16\n\t"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %16\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %16\n\t"
#define I_CVT(n) "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %16\n\t"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n\t"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define I_PERM16(n) "v_permlane16_swap_b32 %" #n ", %16\n\t"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 16, %" #n "\n\t"
#define I_AND(n) "v_and_b32 %" #n ", 0xffff0000, %" #n "\n\t"
template <int KIND>
__global__ void probe(unsigned long long* out, float seed) {
    // 64-bit registers for the packed forms; the scalar forms use the low halves
    f32x2 a2[16], b2 = {seed, seed * 0.5f}, c2 = {0.25f, 0.125f};
    float a1[16], b1 = seed, c1 = 0.25f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a2[i] = f32x2{seed + i, seed - i}; a1[i] = seed + i; }
    unsigned long long t0, t1;
    __syncthreads();
    STAMP(t0);
    for (int r = 0; r < REP; ++r) {
        if constexpr (KIND == 0) { auto& a = a2; auto b = b2, c = c2; K16(I_PKFMA); }
        if constexpr (KIND == 1) { auto& a = a2; auto b = b2, c = c2; K16(I_PKMUL); }
        if constexpr (KIND == 2) { auto& a = a2; auto b = b2, c = c2; K16(I_PKADD); }
        if constexpr (KIND == 3) { auto& a = a1; auto b = b1, c = c1; K16(I_FMA); }
        if constexpr (KIND == 4) { auto& a = a1; auto b = b1, c = c1; K16(I_MUL); }
        if constexpr (KIND == 5) { auto& a = a1; auto b = b1, c = c1; K16(I_CVT); }
        if constexpr (KIND == 6) { auto& a = a1; auto b = b1, c = c1; K16(I_EXP); }
        if constexpr (KIND == 7) { auto& a = a1; auto b = b1, c = c1; K16(I_MOVDPP); }
        if constexpr (KIND == 8) { auto& a = a1; auto b = b1, c = c1; K16(I_MAX3); }
        if constexpr (KIND == 9) { auto& a = a1; auto b = b1, c = c1; K16(I_LSHL); }
        if constexpr (KIND == 10) { auto& a = a1; auto b = b1, c = c1; K16(I_AND); }
    }
    STAMP(t1);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a2[i][0] + a2[i][1] + a1[i];
    if (acc == 12345.678f) out[0] = 1;  // keep the chains alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND>
void run(const char* name, unsigned long long* d) {
    for (int threads : {256, 512}) {
        const int waves = 256 * threads / 64;
        probe<KIND><<<256, threads>>>(d, 1.0f);
        probe<KIND><<<256, threads>>>(d, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1 + waves);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin() + 1, h.end());
        printf("%-22s %d waves/SIMD: median %.2f  min %.2f cycles per instruction per wave\n", name, threads / 256,
               h[1 + waves / 2] / (double)(REP * 16), h[1] / (double)(REP * 16));
    }
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * (1 + 256 * 8));
    run<0>("v_pk_fma_f32", d);
    run<1>("v_pk_mul_f32", d);
    run<2>("v_pk_add_f32", d);
    run<3>("v_fma_f32", d);
    run<4>("v_mul_f32", d);
    run<5>("v_cvt_pk_bf16_f32", d);
    run<6>("v_exp_f32", d);
    run<7>("v_mov_b32_dpp row_ror", d);
    run<8>("v_max3_f32", d);
    run<9>("v_lshlrev_b32", d);
    run<10>("v_and_b32 (literal)", d);
    return 0;
}
