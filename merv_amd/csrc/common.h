// Common device/host helpers for the MERV multi-encoder video forward path on gfx950 (MI355X).
// Everything here is written for CDNA4 only: 64-lane wavefronts, bf16 MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace merv {

typedef uint16_t bf16_t;  // raw bf16 bits; all activations / GEMM weights are stored this way
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MERV_DEVICE __device__ __forceinline__

// f32 -> bf16, round-to-nearest-even. A plain cast lowers to v_cvt_pk_bf16_f32 at -O3 and keeps
// NaNs NaN (MI355X_MICROARCH "Correctness boundaries").
MERV_DEVICE bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(bf16_t, b);
}
MERV_DEVICE float bf2f(bf16_t x) { return __builtin_bit_cast(float, ((uint32_t)x) << 16); }
// two f32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving)
MERV_DEVICE uint32_t pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
MERV_DEVICE float bflo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
MERV_DEVICE float bfhi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// Tuning hooks. The product library reads NO environment variable and keeps no process-global switch (no hidden global
// state): kernel selection is a function of a call's arguments alone. The probe / test build of the same sources (-DMERV_TUNING_HOOKS:
// `make hooks` -> merv_amd/lib/libmerv_hip_hooks.so, loaded only under MERV_TUNING_HOOKS=1; tools/probes/build_ab*.sh) reads the MERV_* tuning
// variables through this function and honours merv_debug_set_gemm_variant / merv_debug_set_attn_rescale_thr (no-ops in the product).
#ifdef MERV_TUNING_HOOKS
inline const char* merv_tuning_env(const char* name) { return getenv(name); }
constexpr bool MERV_HOOKS = true;
#else
inline const char* merv_tuning_env(const char*) { return nullptr; }
constexpr bool MERV_HOOKS = false;
#endif

// Activation kinds (ABI values, see include/merv_hip.h).
enum : int { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_TANH = 2, ACT_QUICK_GELU = 3 };

template <int ACT>
MERV_DEVICE float activate(float x) {
    if constexpr (ACT == ACT_GELU_ERF) {
        // timm nn.GELU (exact erf form): 0.5 x (1 + erf(x / sqrt 2)). erfc(|z|) by Abramowitz-Stegun 7.1.26
        // (|abs err| <= 1.5e-7, far below the bf16 output rounding); 1 + erf(z) = 2 - erfc(z) for z >= 0 and
        // erfc(-z) for z < 0, so the negative tail keeps its relative accuracy. One v_rcp + one v_exp + 6 FMAs.
        const float z = x * 0.70710678118654752440f;
        const float az = fabsf(z);
        const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
        float poly = 1.061405429f;
        poly = poly * t - 1.453152027f;
        poly = poly * t + 1.421413741f;
        poly = poly * t - 0.284496736f;
        poly = poly * t + 0.254829592f;
        const float e = poly * t * __expf(-az * az);  // erfc(|z|)
        const float one_plus_erf = z >= 0.f ? 2.0f - e : e;
        return 0.5f * x * one_plus_erf;
    } else if constexpr (ACT == ACT_GELU_TANH) {
        // HF "gelu_fast": 0.5x(1+tanh(0.7978845608 x (1+0.044715 x^2)))
        float u = 0.7978845608f * x * (1.0f + 0.044715f * x * x);
        // tanh(u) = 1 - 2/(exp(2u)+1); saturates cleanly at +-1 for |u| large
        float e = __expf(2.0f * u);
        float t = 1.0f - 2.0f / (e + 1.0f);
        return 0.5f * x * (1.0f + t);
    } else if constexpr (ACT == ACT_QUICK_GELU) {
        // CLIP quick_gelu: x * sigmoid(1.702 x)
        return x / (1.0f + __expf(-1.702f * x));
    } else {
        return x;
    }
}

// Pair form of the activations for the GEMM epilogue: the fma / mul / add chains are written on 2-vectors so hipcc emits
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two values per issue slot); only the transcendentals are per element.
// With all eight waves of a block in the epilogue at once the activation is pure VALU time with the matrix pipe idle:
// measured 22-30 % of an fc1 launch before this form (tools/fc1_probe.py).
typedef __attribute__((ext_vector_type(2))) float f32x2;

// a * b and fma(a, b, 0.5) clamped to [0, 1] by the VOP3 output modifier of the instruction itself (written as fmed3(.., 0, 1) hipcc
// SLP-packs the product into a v_pk_mul_f32 and then spends a v_max_f32 .. clamp per element on the clamp)
MERV_DEVICE float mul_clamp01(float a, float b) {
    float r;
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
MERV_DEVICE float fma_half_clamp01(float a, float b) {  // clamp(a * b + 0.5): 0.5 is an inline constant of the encoding
    float r;
    asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Phi(x) - 0.5 = (x/c) Q((x/c)^2) on |x| <= c = 4.25: degree-8 minimax for the absolute error of x Phi(x) (see activate2)
__device__ constexpr float GELU_Q[9] = {1.6949809279f, -5.0811122567f, 13.475584255f, -27.024260417f, 40.026775381f,
                                        -42.054740051f, 29.308115773f, -12.034233795f, 2.1888832456f};

template <int ACT>
MERV_DEVICE f32x2 activate2(f32x2 x) {
    if constexpr (ACT == ACT_GELU_ERF) {
        // gelu(x) = x Phi(x), Phi(x) = 0.5 + 0.5 erf(x / sqrt 2), with NO transcendental instruction (round 3): on [-c, c],
        // c = 4.25, Phi(x) - 0.5 = (x/c) Q((x/c)^2) with Q the degree-8 minimax polynomial for the absolute error of x Phi(x)
        // (LP fit on 6000 Chebyshev nodes: 1.6e-5; evaluated in fp32 incl. the cut-off 4.6e-5 -- bf16 resolves 6e-5 at |y| = 0.016
        // and 3.9e-3 at |y| = 1); outside, t = clamp((x/c)^2) = 1 makes 0.5 + (x/c) Q(1) overshoot [0, 1] and the second clamp
        // returns exactly 0 or 1 (true Phi(-4.25) = 1.07e-5). Both clamps are the VOP3 output modifier of a v_mul / v_fma, i.e.
        // free. 11 packed + 4 plain VALU per pair against v_rcp + v_exp + 11 packed + 2 v_max of the erfc form this replaces
        // (A&S 7.1.26, still the scalar activate<> above): in the epilogue all eight waves of a block are in VALU code with the
        // matrix pipe idle, and the erfc form cost a LanguageBind fc1 launch 90 us of 541 (tools/probes/fc1_epilogue_cost.py).
        const f32x2 xs = x * (1.0f / 4.25f);
        const f32x2 t = {mul_clamp01(xs[0], xs[0]), mul_clamp01(xs[1], xs[1])};
        f32x2 q = t * GELU_Q[8] + GELU_Q[7];
#pragma unroll
        for (int k = 6; k >= 0; --k) q = q * t + GELU_Q[k];
        const f32x2 phi = {fma_half_clamp01(xs[0], q[0]), fma_half_clamp01(xs[1], q[1])};
        return x * phi;
    } else if constexpr (ACT == ACT_GELU_TANH) {
        // 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + exp(-2u)),  u = 0.7978845608 x (1 + 0.044715 x^2)
        const f32x2 q = (x * x) * 0.044715f + 1.0f;
        const f32x2 a = (x * q) * (-2.0f * 0.7978845608f * 1.4426950408889634f);  // -2u log2(e)
        const f32x2 d = {__builtin_amdgcn_exp2f(a[0]) + 1.0f, __builtin_amdgcn_exp2f(a[1]) + 1.0f};
        const f32x2 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return x * rc;
    } else if constexpr (ACT == ACT_QUICK_GELU) {
        const f32x2 a = x * (-1.702f * 1.4426950408889634f);
        const f32x2 d = {__builtin_amdgcn_exp2f(a[0]) + 1.0f, __builtin_amdgcn_exp2f(a[1]) + 1.0f};
        const f32x2 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return x * rc;
    } else {
        return x;
    }
}

// Two pairs at once. For the erf GELU the two Horner chains are written interleaved: a v_pk_fma_f32 whose operand was written by
// the instruction right before it needs a wait state (hipcc pads every step of a lone chain with s_nop 0, and under the register
// pressure of the GEMM epilogue it does not interleave two activate2 calls by itself).
template <int ACT>
MERV_DEVICE void activate4(f32x2& a, f32x2& b) {
    if constexpr (ACT == ACT_GELU_ERF) {
        const f32x2 xa = a * (1.0f / 4.25f), xb = b * (1.0f / 4.25f);
        const f32x2 ta = {mul_clamp01(xa[0], xa[0]), mul_clamp01(xa[1], xa[1])};
        const f32x2 tb = {mul_clamp01(xb[0], xb[0]), mul_clamp01(xb[1], xb[1])};
        f32x2 qa = ta * GELU_Q[8] + GELU_Q[7];
        f32x2 qb = tb * GELU_Q[8] + GELU_Q[7];
#pragma unroll
        for (int k = 6; k >= 0; --k) {
            qa = qa * ta + GELU_Q[k];
            qb = qb * tb + GELU_Q[k];
        }
        const f32x2 pa = {fma_half_clamp01(xa[0], qa[0]), fma_half_clamp01(xa[1], qa[1])};
        const f32x2 pb = {fma_half_clamp01(xb[0], qb[0]), fma_half_clamp01(xb[1], qb[1])};
        a = a * pa;
        b = b * pb;
    } else {
        a = activate2<ACT>(a);
        b = activate2<ACT>(b);
    }
}

MERV_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MERV_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a 1-D block id (cdna_hip_programming.md section 5, "XCD swizzle must be
// bijective"): blocks that share an XCD (b % 8 equal) get a contiguous chunk of the logical grid, so
// neighbouring tiles share operand panels in one XCD's L2. Speed only, never correctness.
MERV_DEVICE int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}


// ---- MXFP8 (OCP Microscaling, e4m3 elements, E8M0 scale per 32 consecutive k) helpers shared by the quantiser, the
// LayerNorm kernel and the GEMM epilogue. A 32-element block is held by 4 consecutive lanes, 8 elements each. ----
// Shared exponent of a block: floor(log2(amax)) - 8 (8 = emax of e4m3), plus one when amax / 2^e would exceed the largest
// e4m3 value 448 = 1.75 * 2^8 -- i.e. the smallest power-of-two scale under which no element saturates. (The OCP
// reference conversion stops at the floor and saturates the top eighth of the block maximum's binade; the extra step
// costs the small elements one bit and lowers the product error from 4.2 % to 3.8 % on Gaussian data, 4.9 % to 3.8 %
// with outlier channels.) Clamped to the E8M0 range.
MERV_DEVICE int mx_shared_exponent(float amax) {
    const uint32_t bits = __float_as_uint(amax);
    int e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    return e < -127 ? -127 : e;
}
MERV_DEVICE uint32_t mx_pack4(float a0, float a1, float a2, float a3, float inv) {
    a0 = fminf(fmaxf(a0 * inv, -448.f), 448.f);
    a1 = fminf(fmaxf(a1 * inv, -448.f), 448.f);
    a2 = fminf(fmaxf(a2 * inv, -448.f), 448.f);
    a3 = fminf(fmaxf(a3 * inv, -448.f), 448.f);
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, packed, true);
    return (uint32_t)packed;
}
// 8 values of this lane (one quarter of a block; lanes l, l^1, l^2, l^3 hold the block): returns the 8 e4m3 bytes and
// the block's biased exponent byte. Every lane of the 4-lane group must call it (cross-lane max).
MERV_DEVICE u32x2 mx_quantize8(const float* v, int& scale_byte) {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    // maximum over the quad by two DPP moves (quad_perm [1,0,3,2], [2,3,0,1]): no LDS crossbar (__shfl_xor lowers to ds_bpermute, ~100 cycles
    // each, two per 8 elements in every MXFP8 epilogue); a maximum is exact, the bits do not change (round 6)
    amax = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0xB1, 0xf, 0xf, true)));
    amax = fmaxf(amax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, amax), 0x4E, 0xf, 0xf, true)));
    const int e = mx_shared_exponent(amax);
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
    scale_byte = e + 127;
    return u32x2{mx_pack4(v[0], v[1], v[2], v[3], inv), mx_pack4(v[4], v[5], v[6], v[7], inv)};
}
// byte offset of the scale of (row, kblock) in the GEMM's lane-order layout [kblock/4][row/64][(kblock%4)*16 + row%16][(row%64)/16]
MERV_DEVICE size_t mx_scale_offset(int row, int kb, int groups) {
    return ((((size_t)(kb >> 2) * groups + (row >> 6)) * 64 + (kb & 3) * 16 + (row & 15)) << 2) + ((row & 63) >> 4);
}

}  // namespace merv
