// Backward of the trainable tail of the visual path (projector + fusion gradients; the
// encoders stay forward-only, merv.py:562). What the reference gets from torch autograd through
// AveragePooling3DProjector.forward (nn_utils.py:320-330) and CrossAttentionAdapterLearnableQuery.forward
// (nn_utils.py:487-521) under loss.backward() (training/strategies/base_strategy.py), written out:
//
//   fusion      out = sum_e w_e V_e,  w = softmax_e(s),  s_e = mean_t(V_e) . u
//               dw_e   = sum_{t,c} g . V_e                                   (reduce kernel, with vbar_e = mean_t V_e)
//               ds_e   = w_e (dw_e - sum_j w_j dw_j),  du = sum_{b,e} ds_e vbar_e    (B x E scalars: host side)
//               dV_e   = w_e g + ds_e u / T                                  (mix kernel)
//   projector   Y = pool(X) W^T + b  (X from a frozen encoder: no dX)
//               dW = dY^T pool(X)  -> two LDS-tiled transposes + the forward GEMM kernel with K = rows
//               db = column sums of dY (two-pass, fixed order)
//
// All kernels here are HBM-bound streams (16 bytes per lane where the layout allows) with fp32 accumulation and
// fixed-order two-pass reductions: no float atomics, results are run-to-run identical.
#include "common.h"
#include "kernels.h"

namespace merv {
namespace {

MERV_DEVICE void unpack8(u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(v[i]); f[2 * i + 1] = bfhi(v[i]); }
}
MERV_DEVICE u32x4 pack8f(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

constexpr int BWD_ROWS = 16;  // tokens per pass-1 block (same chunking as the forward score kernel)

// pass 1: per (chunk, e, b): partial_dw = sum_{t in chunk, c} g v ; partial_vs[c] = sum_{t in chunk} v
__global__ __launch_bounds__(256) void fusion_bwd_reduce_kernel(FusionBwdArgs p) {
    const int nchunk = (p.T + BWD_ROWS - 1) / BWD_ROWS;
    const int chunk = blockIdx.x, e = blockIdx.y, b = blockIdx.z;
    const bf16_t* v = p.v[e] + (size_t)b * p.T * p.C;
    const bf16_t* g = p.grad_out + (size_t)b * p.T * p.C;
    float* vs_out = p.partial_vs + (((size_t)b * p.E + e) * nchunk + chunk) * p.C;
    const int dc = p.C >> 3;
    const int t0 = chunk * BWD_ROWS, t1 = min(p.T, t0 + BWD_ROWS);
    float dw = 0.f;
    for (int c = threadIdx.x; c < dc; c += 256) {
        float vs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = t0; t < t1; ++t) {
            float fv[8], fg[8];
            unpack8(*(const u32x4*)(v + (size_t)t * p.C + c * 8), fv);
            unpack8(*(const u32x4*)(g + (size_t)t * p.C + c * 8), fg);
#pragma unroll
            for (int j = 0; j < 8; ++j) { vs[j] += fv[j]; dw += fv[j] * fg[j]; }
        }
        *(float4*)(vs_out + c * 8) = float4{vs[0], vs[1], vs[2], vs[3]};
        *(float4*)(vs_out + c * 8 + 4) = float4{vs[4], vs[5], vs[6], vs[7]};
    }
    __shared__ float red[4];
    dw = wave_sum(dw);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dw;
    __syncthreads();
    if (threadIdx.x == 0) p.partial_dw[((size_t)b * p.E + e) * nchunk + chunk] = red[0] + red[1] + red[2] + red[3];
}

// pass 2: vbar[b][e][c] = (sum_chunk partial_vs) / T ; dw[b][e] = sum_chunk partial_dw   (chunks in index order)
__global__ __launch_bounds__(256) void fusion_bwd_finish_kernel(FusionBwdArgs p) {
    const int nchunk = (p.T + BWD_ROWS - 1) / BWD_ROWS;
    const int e = blockIdx.y, b = blockIdx.z;
    const size_t be = (size_t)b * p.E + e;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < p.C) {
        float a = 0.f;
        for (int k = 0; k < nchunk; ++k) a += p.partial_vs[(be * nchunk + k) * p.C + c];
        p.vbar[be * p.C + c] = a / (float)p.T;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float a = 0.f;
        for (int k = threadIdx.x; k < nchunk; k += 64) a += p.partial_dw[be * nchunk + k];
        a = wave_sum(a);
        if (threadIdx.x == 0) p.dw[be] = a;
    }
}

// dV_e = w_e g + (ds_e / T) u
__global__ __launch_bounds__(256) void fusion_bwd_mix_kernel(FusionBwdMixArgs p) {
    const int b = blockIdx.y;
    __shared__ float w_s[8], d_s[8];
    if (threadIdx.x < p.E) {
        w_s[threadIdx.x] = p.w[b * p.E + threadIdx.x];
        d_s[threadIdx.x] = p.ds[b * p.E + threadIdx.x] / (float)p.T;
    }
    __syncthreads();
    const int dc = p.C >> 3;
    const long long per_b = (long long)p.T * dc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_b; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % dc);
        const size_t off = (size_t)b * p.T * p.C + (size_t)i * 8;
        float fg[8];
        unpack8(*(const u32x4*)(p.grad_out + off), fg);
        const float4 u0 = *(const float4*)(p.u + c * 8), u1 = *(const float4*)(p.u + c * 8 + 4);
        const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        for (int e = 0; e < p.E; ++e) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = w_s[e] * fg[j] + d_s[e] * uu[j];
            *(u32x4*)(p.dv[e] + off) = pack8f(o);
        }
    }
}

// out[c][r] = in[r][c] for r < R, 0 for R <= r < Rpad. 64 x 64 tiles through LDS, two bf16 per lane on both sides.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(TransposeArgs p) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const unsigned short* in = (const unsigned short*)p.in;
    unsigned short* out = (unsigned short*)p.out;
    for (int i = ty; i < 64; i += 8) {
        const int r = r0 + i, c = c0 + 2 * tx;
        unsigned int v = 0;
        if (r < p.R) {
            if (c + 1 < p.C) v = *(const unsigned int*)(in + (size_t)r * p.ldi + c);
            else if (c < p.C) v = in[(size_t)r * p.ldi + c];
        }
        tile[i][2 * tx] = (unsigned short)(v & 0xffffu);
        tile[i][2 * tx + 1] = (unsigned short)(v >> 16);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 8) {
        const int c = c0 + i, r = r0 + 2 * tx;  // output row c, output columns r, r+1
        if (c < p.C && r < p.Rpad) {
            const unsigned int v = (unsigned int)tile[2 * tx][i] | ((unsigned int)tile[2 * tx + 1][i] << 16);
            *(unsigned int*)(out + (size_t)c * p.ldo + r) = v;  // Rpad and ldo are even: r + 1 < Rpad
        }
    }
}

constexpr int CS_CHUNKS = 64;
// pass 1: partial[k][n] = sum of rows of chunk k ; pass 2: out[n] = sum_k partial[k][n]
__global__ __launch_bounds__(256) void colsum_partial_kernel(ColsumArgs p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.N) return;
    const int rows_per = (p.M + CS_CHUNKS - 1) / CS_CHUNKS;
    const int m0 = blockIdx.y * rows_per, m1 = min(p.M, m0 + rows_per);
    float a = 0.f;
    const unsigned short* x = (const unsigned short*)p.x;
    for (int m = m0; m < m1; ++m) a += __uint_as_float((unsigned int)x[(size_t)m * p.ld + n] << 16);
    p.partial[(size_t)blockIdx.y * p.N + n] = a;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(ColsumArgs p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.N) return;
    float a = 0.f;
    for (int k = 0; k < CS_CHUNKS; ++k) a += p.partial[(size_t)k * p.N + n];
    p.out[n] = a;
}

}  // namespace

size_t fusion_bwd_workspace_floats(int B, int E, int T, int C) {
    const size_t nchunk = (size_t)(T + BWD_ROWS - 1) / BWD_ROWS;
    return (size_t)B * E * nchunk * ((size_t)C + 1);
}

hipError_t launch_fusion_bwd_reduce(FusionBwdArgs a, float* ws, hipStream_t s) {
    if (a.E < 1 || a.E > 8 || a.C % 8 != 0) return hipErrorInvalidValue;
    if (a.B <= 0 || a.T <= 0) return hipSuccess;
    const int nchunk = (a.T + BWD_ROWS - 1) / BWD_ROWS;
    a.partial_vs = ws;
    a.partial_dw = ws + (size_t)a.B * a.E * nchunk * a.C;
    hipLaunchKernelGGL(fusion_bwd_reduce_kernel, dim3(nchunk, a.E, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(fusion_bwd_finish_kernel, dim3((a.C + 255) / 256, a.E, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fusion_bwd_mix(const FusionBwdMixArgs& a, hipStream_t s) {
    if (a.E < 1 || a.E > 8 || a.C % 8 != 0) return hipErrorInvalidValue;
    if (a.B <= 0 || a.T <= 0) return hipSuccess;
    const long long per_b = (long long)a.T * (a.C / 8);
    int gx = (int)((per_b + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(fusion_bwd_mix_kernel, dim3(gx, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_transpose(const TransposeArgs& a, hipStream_t s) {
    if (a.R <= 0 || a.C <= 0) return hipSuccess;
    if (a.Rpad < a.R || (a.Rpad & 1) || (a.ldo & 1) || (a.ldi & 1) || a.ldo < a.Rpad || a.ldi < a.C) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((a.C + 63) / 64, (a.Rpad + 63) / 64), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t colsum_workspace_floats(int N) { return (size_t)CS_CHUNKS * N; }

hipError_t launch_colsum(const ColsumArgs& a, hipStream_t s) {
    if (a.N <= 0) return hipSuccess;
    if (a.M <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((a.N + 255) / 256, CS_CHUNKS), dim3(256), 0, s, a);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((a.N + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace merv
