// MXFP8 quantisation for the fp8 encoder-GEMM mode (BASELINE.json configs[4]: "fp8 MFMA encoder GEMMs"). OCP
// Microscaling: blocks of 32 consecutive k share one E8M0 scale 2^e, e = floor(log2(max|v|)) - 8 (8 = emax of e4m3),
// +1 when max|v| / 2^e > 448 (common.h, mx_shared_exponent); elements are v / 2^e rounded to nearest even into OCP e4m3. HBM-bound: reads 64 B, writes
// 33 B per block; one lane per block, 16-byte accesses.
#include "common.h"
#include "kernels.h"

namespace merv {
namespace {

__global__ __launch_bounds__(256) void mx_quantize_kernel(MxQuantArgs p) {
    const int kblocks = p.K >> 5;
    const long long total = (long long)p.rows * kblocks;
    const int groups = (p.rows + 63) >> 6;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int row = (int)(i / kblocks), kb = (int)(i - (long long)row * kblocks);
        const bf16_t* src = p.x + (size_t)row * p.ld + kb * 32;
        u32x4 raw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[c] = *(const u32x4*)(src + c * 8);
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                v[c * 8 + 2 * w] = bflo(raw[c][w]);
                v[c * 8 + 2 * w + 1] = bfhi(raw[c][w]);
                amax = fmaxf(amax, fmaxf(fabsf(v[c * 8 + 2 * w]), fabsf(v[c * 8 + 2 * w + 1])));
            }
        const int e = mx_shared_exponent(amax);  // amax == 0 (or subnormal) -> smallest scale
        const float inv = __uint_as_float((uint32_t)(127 - e) << 23);  // 2^-e, exponent field in [1, 254]
        uint32_t outw[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) outw[w] = mx_pack4(v[4 * w], v[4 * w + 1], v[4 * w + 2], v[4 * w + 3], inv);
        uint8_t* dst = p.q + (size_t)row * p.K + kb * 32;
        *(u32x4*)dst = u32x4{outw[0], outw[1], outw[2], outw[3]};
        *(u32x4*)(dst + 16) = u32x4{outw[4], outw[5], outw[6], outw[7]};
        p.scales[mx_scale_offset(row, kb, groups)] = (uint8_t)(e + 127);
    }
}

// one wave per row: lane l takes the 32-element blocks l, l + 64, ...
__global__ __launch_bounds__(256) void mx_colsum_kernel(const uint8_t* q, const uint8_t* scales, float* colsum, int N, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const int groups = (N + 63) >> 6;
    float acc = 0.f;
    for (int kb = lane; kb < (K >> 5); kb += 64) {
        const float sc = __uint_as_float((uint32_t)scales[mx_scale_offset(row, kb, groups)] << 23);  // 2^(byte - 127); byte 0 = 2^-127 (a zero block)
        const uint32_t* src = (const uint32_t*)(q + (size_t)row * K + kb * 32);
        float b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int v = (int)src[w];
            b += (__builtin_amdgcn_cvt_f32_fp8(v, 0) + __builtin_amdgcn_cvt_f32_fp8(v, 1)) + (__builtin_amdgcn_cvt_f32_fp8(v, 2) + __builtin_amdgcn_cvt_f32_fp8(v, 3));
        }
        acc = fmaf(b, sc, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) colsum[row] = acc;
}

}  // namespace

hipError_t launch_mx_colsum(const uint8_t* q, const uint8_t* scales, float* colsum, int N, int K, hipStream_t s) {
    if (N <= 0 || K <= 0 || K % 128 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mx_colsum_kernel, dim3((N + 3) / 4), dim3(256), 0, s, q, scales, colsum, N, K);
    return hipGetLastError();
}

size_t mx_scale_bytes(int rows, int K) { return (size_t)(K / 128) * ((rows + 63) / 64) * 256; }

hipError_t launch_mx_quantize(const MxQuantArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.K <= 0 || a.K % 128 != 0 || a.ld % 8 != 0 || a.ld < a.K) return hipErrorInvalidValue;
    const long long total = (long long)a.rows * (a.K / 32);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mx_quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace merv
