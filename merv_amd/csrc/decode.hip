// Batch-1 token decode kernels for the LLM hand-off (merv/models/vidlms/merv.py:818-825 ->
// HF GenerationMixin's per-token forward of LlamaForCausalLM / MistralForCausalLM, transformers modeling_llama).
//
// The north_star keeps the LLM on PyTorch-ROCm; the e2e half of the metric moved two things here. (1) A decode step is ~1100 tiny PyTorch kernels per token
// (RMSNorm = 8 launches, rotary embedding = 10, ...): measured on MI355X, 6.1 of the 10.4 ms of a graph-replayed
// Llama-2-7B step are those launches, 4.7 ms the library's M = 1 GEMMs at 2.4-3.4 TB/s (tools/probes/decode_breakdown.py).
// These kernels are the step as 5 launches per layer (7 through the separate rotary / attention / merge entry points), every one
// a pure HBM stream:
//
//   rmsnorm_kernel         y = w * bf16(x * rsqrt(mean(x^2) + eps))                    (LlamaRMSNorm.forward)
//   gemv_kernel            y = bf16(W x) [+ residual]      W [N, K] bf16 streamed once (nn.Linear, M = 1)
//   gemv_silu_mul_kernel   y = bf16(silu(bf16(Wg x))) * bf16(Wu x)                      (LlamaMLP: act_fn(gate) * up)
//   gemv_xlds_kernel       the same two, one row per wave: x once per block in LDS, the whole row in flight (round 5; plain launches + gate-up pair)
//   rope_cache_kernel      q, k <- rotary(pos); K / V cache[pos] <- k, v              (apply_rotary_pos_emb + cache update)
//   decode_attn_kernel     one query per head against cache[0 .. pos], split over positions, (m, l, o) partials
//   decode_attn_merge_kernel  merges the partials                                      (flash-decoding)
//   decode_attn_split_kernel + oproj_merge_kernel   the default pair since round 4: rotary + cache + split attention ending at the
//                          partials, and the o-projection that merges them under its first weight trip
//   greedy_advance_kernel  argmax -> next token, token log, position + 1 inside the captured step (greedy search)
//   sample_advance_kernel  the same hand-over with the token drawn from softmax(logits / T) (Gumbel-max on a Philox stream)
// (2) The prompt prefill keeps its GEMMs on the library and takes everything between them from here and attention.hip:
//   rmsnorm_kernel (rows = S), add_rmsnorm_kernel, prefill_rope_cache_kernel, silu_mul_kernel, prefill_attn_kernel
//
// Rounding points follow the bf16 module: every nn.Linear output, every elementwise result is rounded to bf16 where the
// PyTorch graph materialises a bf16 tensor; accumulation and softmax are fp32. The position is read from device memory
// (one int64), so a step captured in a hipGraph replays for every position.
#include "common.h"
#include "kernels.h"

namespace merv {
namespace {

MERV_DEVICE float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MERV_DEVICE void unpack8f(const u32x4& p, float (&f)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { f[2 * q] = bflo(p[q]); f[2 * q + 1] = bfhi(p[q]); }
}
MERV_DEVICE float round_bf(float x) { return bf2f(f2bf(x)); }

// sum of squares of the 16-byte chunks c with (c / 64) % 4 == part of a row (one wave; all 64 lanes get the result), in two
// steps so that a caller can put other loads between the request of the first four chunks per lane (K <= 8192: all of them) and
// their use. The row's mean(x^2) is ALWAYS formed as ((part 0 + part 1) + part 2) + part 3, whether one wave computes the four
// parts (rmsnorm_kernel) or the four waves of a GEMV block one each (fused norm): the fused launches stay bit-identical to
// RMSNorm + GEMV.
MERV_DEVICE void sumsq_request(const bf16_t* x, int nchunk, int lane, int part, u32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * part + 256 * i;
        v[i] = c < nchunk ? *(const u32x4*)(x + c * 8) : u32x4{0u, 0u, 0u, 0u};  // (a zero chunk adds exactly nothing)
    }
}
MERV_DEVICE float sumsq_finish(const bf16_t* x, int nchunk, int lane, int part, const u32x4 (&v)[4]) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8f(v[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
    }
    for (int c = lane + 64 * part + 1024; c < nchunk; c += 256) {
        float f[8];
        unpack8f(*(const u32x4*)(x + c * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
    }
    return wave_sum64(ss);
}
MERV_DEVICE float sumsq_part(const bf16_t* x, int nchunk, int lane, int part) {
    u32x4 v[4];
    sumsq_request(x, nchunk, lane, part, v);
    return sumsq_finish(x, nchunk, lane, part, v);
}

// ---- RMSNorm: one wave per row ----
__global__ __launch_bounds__(256) void rmsnorm_kernel(DecodeRmsArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const bf16_t* x = p.x + (size_t)row * p.D;
    const int nchunk = p.D >> 3;
    const float ss = ((sumsq_part(x, nchunk, lane, 0) + sumsq_part(x, nchunk, lane, 1)) + sumsq_part(x, nchunk, lane, 2)) +
                     sumsq_part(x, nchunk, lane, 3);
    const float rstd = rsqrtf(ss / (float)p.D + p.eps);
    bf16_t* y = p.y + (size_t)row * p.D;
    for (int c = lane; c < nchunk; c += 64) {
        float f[8], w[8];
        unpack8f(*(const u32x4*)(x + c * 8), f);  // second read: L1 / L2 resident
        unpack8f(*(const u32x4*)(p.w + c * 8), w);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)  // weight * hidden.to(bf16): two roundings, as the module
            o[q] = pack2bf(w[2 * q] * round_bf(f[2 * q] * rstd), w[2 * q + 1] * round_bf(f[2 * q + 1] * rstd));
        *(u32x4*)(y + c * 8) = o;
    }
}

// ---- prompt prefill: residual add + RMSNorm of the sum in one pass (x <- bf16(x + delta); y = RMSNorm(x)): one BLOCK per row ----
// Wave w takes part w of the row (chunks lane + 64 w + 256 i, as sumsq_part) and the four parts meet in LDS, so the mean of squares is
// formed exactly as rmsnorm_kernel forms it on the updated row -- from registers instead of a second read. D <= 8192.
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(DecodeRmsArgs p, const bf16_t* delta) {
    __shared__ float part_ss[4];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int row = blockIdx.x;
    bf16_t* x = const_cast<bf16_t*>(p.x) + (size_t)row * p.D;
    const bf16_t* d = delta + (size_t)row * p.D;
    const int nchunk = p.D >> 3;
    u32x4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * part + 256 * i;
        v[i] = u32x4{0u, 0u, 0u, 0u};
        if (c < nchunk) {
            float a[8], b[8];
            unpack8f(*(const u32x4*)(x + c * 8), a);
            unpack8f(*(const u32x4*)(d + c * 8), b);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i][q] = pack2bf(a[2 * q] + b[2 * q], a[2 * q + 1] + b[2 * q + 1]);
            *(u32x4*)(x + c * 8) = v[i];
        }
        float f[8];
        unpack8f(v[i], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) ss = fmaf(f[q], f[q], ss);  // (a zero chunk adds exactly nothing)
    }
    ss = wave_sum64(ss);
    if (lane == 0) part_ss[part] = ss;
    __syncthreads();
    const float tot = ((part_ss[0] + part_ss[1]) + part_ss[2]) + part_ss[3];
    const float rstd = rsqrtf(tot / (float)p.D + p.eps);
    bf16_t* y = p.y + (size_t)row * p.D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * part + 256 * i;
        if (c < nchunk) {
            float f[8], w[8];
            unpack8f(v[i], f);
            unpack8f(*(const u32x4*)(p.w + c * 8), w);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack2bf(w[2 * q] * round_bf(f[2 * q] * rstd), w[2 * q + 1] * round_bf(f[2 * q + 1] * rstd));
            *(u32x4*)(y + c * 8) = o;
        }
    }
}

// ---- GEMV: W [N, K] bf16 row-major streamed once; each wave owns ROWS output rows, lanes stride K in 16-byte chunks ----
constexpr int GEMV_WAVES = 4;  // waves per block
constexpr int GEMV_XN_CHUNKS = 2048;  // fused norm: the block's normalised input as an LDS image of 16-byte chunks (32 KB: K <= 16384)
struct GemvLds {
    float norm_part[GEMV_WAVES];
};

// NW_MATS 1: y = W x (+ res); 2: y = silu(Wg x) * (Wu x); NORM: RMSNorm of x fused in; GEMV_ROWS: output rows per wave (the x chunk
// is reused across them). `block`: this workgroup's index.
// Fused norm (round 5): the BLOCK normalises x once -- thread t the chunks t + 256 i it already holds for its share of mean(x^2) --
// into an LDS image, and every wave multiplies from that image. Until round 4 each wave normalised every chunk it met (two multiplies and
// two bf16 roundings per element: ~80 VALU instructions per chunk against 16-32 of multiply-add for the wave's rows, and two more global
// loads per chunk): the q / k / v launch ran 20.3 us against 17.0 for the same bytes without a norm. Same values (the formula and its
// rounding points are unchanged), same accumulation order: same bits.
// EXACT (round 5): K / 8 is a multiple of the trip (64 lanes x UN chunks) and N of the block's rows -- no chunk is ever clamped or zeroed, no row
// past N: the per-chunk compares and selects (as many VALU instructions as the multiply-adds of a one-row wave) are compiled out
template <int NW_MATS, int UN, bool NORM, int GEMV_ROWS, bool EXACT = false>
MERV_DEVICE void gemv_body(DecodeGemvArgs p, const int block, GemvLds& lds, u32x4* xn_lds) {  // xn_lds (NORM): K / 8 chunks of LDS
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: row pointers stay in SGPRs
    int n0 = (block * GEMV_WAVES + wave) * GEMV_ROWS;
    // several matrices in one launch (q / k / v): a wave's rows lie in ONE of them (row counts are multiples of GEMV_ROWS)
    if constexpr (NW_MATS == 1) {
        if (n0 >= p.N && p.Nb > 0) {
            n0 -= p.N;
            if (n0 < p.Nb || p.Nc <= 0) { p.W = p.Wb; p.y = p.yb; p.N = p.Nb; p.bias = p.bias_b; }  // (two matrices: rows past the end idle on the second)
            else { n0 -= p.Nb; p.W = p.Wc; p.y = p.yc; p.N = p.Nc; p.bias = p.bias_c; }
        }
    }
    static_assert(GEMV_WAVES == 4, "the fused norm splits the row's sum of squares over the block's four waves");
    // a wave past the last row still takes part in the fused norm's block-wide work (and then leaves)
    const bool idle = n0 >= p.N;
    if (idle && !NORM) return;
    const int nchunk = p.K >> 3;
    float acc[NW_MATS][GEMV_ROWS];
#pragma unroll
    for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) acc[m][r] = 0.f;
    const bf16_t* wrow[NW_MATS][GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
        const int n = EXACT || n0 + r < p.N ? n0 + r : p.N - 1;
        wrow[0][r] = p.W + (size_t)n * p.K;
        if constexpr (NW_MATS == 2) wrow[1][r] = p.W2 + (size_t)n * p.K;
    }
    // bias / residual of the wave's rows: requested before the weight trips (read at the end they were one more memory round trip at the
    // tail of every wave, i.e. of the launch)
    // (unconditional, raw: an absent vector reads x[0] -- a line the launch reads anyway -- and is not used -- behind a branch, or converted here, hipcc
    // waits for the value on the spot, a round trip BEFORE the first weight request)
    // Only in the launches without a fused norm (o / down projection, lm_head): the q / k / v launch has neither vector in the Llama
    // family, and two more registers took it from six waves per SIMD to five (80 -> 83 VGPRs: +0.5 us).
    constexpr bool PRE_ADD = NW_MATS == 1 && !NORM;
    bf16_t biasv[GEMV_ROWS], resv[GEMV_ROWS];
    if constexpr (PRE_ADD) {
        const bf16_t* bias_p = p.bias ? p.bias : p.x;
        const bf16_t* res_p = p.res ? p.res : p.x;
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) {
            const int n = n0 + r < p.N ? n0 + r : p.N - 1;
            biasv[r] = bias_p[p.bias ? n : 0];
            resv[r] = res_p[p.res ? n : 0];
        }
    }
    // UN chunks per lane per trip: UN x ROWS x NW_MATS weight loads of 16 B (and, without a norm, the x chunks they meet) are in
    // flight per lane before the first use. Every trip is a full batch: chunks past the row's end are clamped to a valid
    // address and meet x = 0 (a remainder loop of single loads costs one memory round trip per iteration: 3 us of the
    // K = 11008 launch).
    u32x4 wv[NW_MATS][GEMV_ROWS][UN], xv[UN];
    auto issue_w = [&](int c) {
        if constexpr (!EXACT) c = c < nchunk ? c : 0;  // (rows shorter than 64 chunks: the lanes past the end load a valid chunk and multiply nothing)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int cu = EXACT || c + 64 * u < nchunk ? c + 64 * u : c;
#pragma unroll
            for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
                for (int r = 0; r < GEMV_ROWS; ++r) wv[m][r][u] = __builtin_nontemporal_load((const u32x4*)(wrow[m][r] + cu * 8));
        }
    };
    auto issue_x = [&](int c) {  // (NORM: the chunks come from the LDS image at their use)
        if constexpr (!NORM) {
            if constexpr (!EXACT) c = c < nchunk ? c : 0;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int cu = EXACT || c + 64 * u < nchunk ? c + 64 * u : c;
                xv[u] = *(const u32x4*)(p.x + cu * 8);
            }
        }
    };
    // fused RMSNorm: the BLOCK reduces mean(x^2) over the whole input once (8 KB, L2-resident; a quarter per wave) and normalises it
    // once. The norm's chunks are requested FIRST (vmcnt is in order: they return first) and the whole first weight trip before
    // anything waits for them.
    int c = lane;
    u32x4 nx[4], nw[4];
    if constexpr (NORM) {
        sumsq_request(p.x, nchunk, lane, wave, nx);
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // the norm weight of the same chunks (thread t: chunks t + 256 i)
            const int cc = lane + 64 * wave + 256 * i;
            nw[i] = cc < nchunk ? *(const u32x4*)(p.norm_w + cc * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // unconditional (idle waves of the last block re-read a valid row, lanes past a short row's end a valid chunk): behind a
    // branch hipcc's waitcnt pass joins the two paths and waits vmcnt(0) for the norm's chunks, i.e. for the whole weight trip
    issue_w(c);
    issue_x(c);
    __builtin_amdgcn_sched_barrier(0);  // the whole first trip is requested before anything waits for the norm's chunks
    if constexpr (NORM) {
        const float part = sumsq_finish(p.x, nchunk, lane, wave, nx);  // (K <= 8192: no load inside)
        if (lane == 0) lds.norm_part[wave] = part;
        __syncthreads();
        const float rstd = rsqrtf((((lds.norm_part[0] + lds.norm_part[1]) + lds.norm_part[2]) + lds.norm_part[3]) / (float)p.K + p.norm_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = lane + 64 * wave + 256 * i;
            if (cc < nchunk) {
                float xf[8], wn[8];
                unpack8f(nx[i], xf);
                unpack8f(nw[i], wn);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)  // weight * hidden.to(bf16): two roundings, as the module (exact as bf16: the image keeps the values)
                    o[q] = pack2bf(wn[2 * q] * round_bf(xf[2 * q] * rstd), wn[2 * q + 1] * round_bf(xf[2 * q + 1] * rstd));
                xn_lds[cc] = o;
            }
        }
        for (int cc = threadIdx.x + 1024; cc < nchunk; cc += 256) {  // K > 8192: the chunks sumsq_finish re-read, once more
            float xf[8], wn[8];
            unpack8f(*(const u32x4*)(p.x + cc * 8), xf);
            unpack8f(*(const u32x4*)(p.norm_w + cc * 8), wn);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack2bf(wn[2 * q] * round_bf(xf[2 * q] * rstd), wn[2 * q + 1] * round_bf(xf[2 * q + 1] * rstd));
            xn_lds[cc] = o;
        }
        __syncthreads();
        if (idle) return;
    }
    for (; c < nchunk; c += 64 * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float xf[8];
            if constexpr (NORM) {
                const int cu = EXACT || c + 64 * u < nchunk ? c + 64 * u : c;
                unpack8f(xn_lds[cu], xf);
            } else {
                unpack8f(xv[u], xf);
            }
            if (!EXACT && c + 64 * u >= nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
                for (int r = 0; r < GEMV_ROWS; ++r) {
                    float wf[8];
                    unpack8f(wv[m][r][u], wf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[m][r] = fmaf(wf[j], xf[j], acc[m][r]);
                }
        }
        if (c + 64 * UN < nchunk) { issue_w(c + 64 * UN); issue_x(c + 64 * UN); }
    }
#pragma unroll
    for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) acc[m][r] = wave_sum64(acc[m][r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) {
            const int n = n0 + r;
            if (n >= p.N) break;
            float v = round_bf(NW_MATS == 1 && p.bias ? acc[0][r] + bf2f(PRE_ADD ? biasv[r] : p.bias[n]) : acc[0][r]);  // the nn.Linear output as a bf16 tensor
            if constexpr (NW_MATS == 2) {
                const float g = v;
                const float s = round_bf(g / (1.f + __expf(-g)));  // F.silu on a bf16 tensor
                v = s * round_bf(acc[1][r]);
            } else if (p.res) {
                v = v + bf2f(PRE_ADD ? resv[r] : p.res[n]);  // x + linear(...), rounded once more below
            }
            if (p.y32) p.y32[n] = v;  // logits: .float() of the bf16 linear output
            else p.y[n] = f2bf(v);
        }
    }
}

template <int NW_MATS, int UN, bool NORM, int GEMV_ROWS = 2, bool EXACT = false>
__global__ __launch_bounds__(GEMV_WAVES * 64) void gemv_kernel(DecodeGemvArgs p) {
    // NORM: the normalised input's image, dynamic LDS sized by the launcher (2 K bytes: 8 KB at K = 4096 -- a fixed 32 KB image halved
    // the resident blocks and cost the q / k / v launch 1.5 us)
    extern __shared__ __attribute__((aligned(16))) char gemv_dyn_lds[];
    __shared__ GemvLds lds;
    gemv_body<NW_MATS, UN, NORM, GEMV_ROWS, EXACT>(p, blockIdx.x, lds, (u32x4*)gemv_dyn_lds);
}

// GEMV, one row per wave, with (a) the block's x -- copied, or RMS-normalised as in gemv_body -- ONCE in an LDS image: in gemv_body's plain
// launches the x chunks are half of a lane's global loads (L2 hits, but the same queue and as many registers per trip as the weights) --
// and (b) the wave's WHOLE row(s) requested up front (TRIPS x UN chunks per lane and matrix, no loop) and multiplied chunk by chunk: with
// one trip in flight at a time a wave of the K = 11008 projection had nothing outstanding while it multiplied (8 waves per CU: 5.3 TB/s
// against 5.9-6.8 for the K = 4096 launches), and left to itself hipcc converts every chunk to fp32 ahead of the serial multiply-add
// chain (330 registers, one wave per SIMD). A lane multiplies its chunks in gemv_body's order (lane, lane + 64, ...), the norm and the
// epilogue are gemv_body's: same bits. Rows of 64 (UN TRIPS - 1) < K / 8 <= 64 UN TRIPS chunks (only a lane's last chunk can lie past the
// end: every other load is base + immediate offset); NORM: K <= 8192.
// NXI (fused norm): chunks per thread of the norm's pass over x -- 2 for K <= 4096 (the chunks 512.. of sumsq_request's four are zeros there
// and add exactly nothing: skipping them leaves the bits and frees 24 registers), 4 up to K = 8192
template <int UN, int TRIPS, int NW_MATS, bool NORM, int NXI = 4>
__global__ __launch_bounds__(GEMV_WAVES * 64) void gemv_xlds_kernel(DecodeGemvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char gemv_dyn_lds[];
    __shared__ GemvLds lds;
    u32x4* x_lds = (u32x4*)gemv_dyn_lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int n0 = blockIdx.x * GEMV_WAVES + wave;
    if constexpr (NW_MATS == 1) {  // several matrices in one launch (q / k / v), as gemv_body
        if (n0 >= p.N && p.Nb > 0) {
            n0 -= p.N;
            if (n0 < p.Nb || p.Nc <= 0) { p.W = p.Wb; p.y = p.yb; p.N = p.Nb; p.bias = p.bias_b; }  // (two matrices: rows past the end idle on the second)
            else { n0 -= p.Nb; p.W = p.Wc; p.y = p.yc; p.N = p.Nc; p.bias = p.bias_c; }
        }
    }
    const bool idle = n0 >= p.N;  // (takes part in the block's image, then leaves)
    const int n = idle ? p.N - 1 : n0;
    const int nchunk = p.K >> 3;
    const bf16_t* wrow[NW_MATS];
    wrow[0] = p.W + (size_t)n * p.K;
    if constexpr (NW_MATS == 2) wrow[1] = p.W2 + (size_t)n * p.K;
    // bias / residual: raw, unconditional, before everything else (gemv_body)
    const bf16_t biasv = (p.bias ? p.bias : p.x)[p.bias ? n : 0], resv = (p.res ? p.res : p.x)[p.res ? n : 0];
    constexpr int XI = NORM ? NXI : (64 * UN * TRIPS + GEMV_WAVES * 64 - 1) / (GEMV_WAVES * 64);
    u32x4 xi[XI], nw[NORM ? NXI : 1];
    if constexpr (NORM) {
#pragma unroll
        for (int i = 0; i < NXI; ++i) {  // sumsq_request's chunks (thread t: t + 256 i) and their norm weights
            const int cc = lane + 64 * wave + 256 * i;
            xi[i] = cc < nchunk ? *(const u32x4*)(p.x + cc * 8) : u32x4{0u, 0u, 0u, 0u};  // (a zero chunk adds exactly nothing)
            nw[i] = cc < nchunk ? *(const u32x4*)(p.norm_w + cc * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int cc = (int)threadIdx.x + GEMV_WAVES * 64 * i;
            xi[i] = *(const u32x4*)(p.x + (cc < nchunk ? cc : 0) * 8);
        }
    }
    u32x4 wv[NW_MATS][TRIPS][UN];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int cu = lane + 64 * (t * UN + u);
            const int cv = t * UN + u < UN * TRIPS - 1 || cu < nchunk ? cu : lane;
#pragma unroll
            for (int m = 0; m < NW_MATS; ++m) wv[m][t][u] = __builtin_nontemporal_load((const u32x4*)(wrow[m] + cv * 8));
        }
    __builtin_amdgcn_sched_barrier(0);  // everything is requested before anything waits
    if constexpr (NORM) {
        float ss = 0.f;  // sumsq_finish for K <= 8192 (no load inside)
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            float f[8];
            unpack8f(xi[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
            asm volatile("" : "+v"(ss));
        }
        ss = wave_sum64(ss);
        if (lane == 0) lds.norm_part[wave] = ss;
        __syncthreads();
        const float rstd = rsqrtf((((lds.norm_part[0] + lds.norm_part[1]) + lds.norm_part[2]) + lds.norm_part[3]) / (float)p.K + p.norm_eps);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int cc = lane + 64 * wave + 256 * i;
            if (cc < nchunk) {
                float xf[8], wn[8];
                unpack8f(xi[i], xf);
                unpack8f(nw[i], wn);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)  // weight * hidden.to(bf16): two roundings, as the module
                    o[q] = pack2bf(wn[2 * q] * round_bf(xf[2 * q] * rstd), wn[2 * q + 1] * round_bf(xf[2 * q + 1] * rstd));
                x_lds[cc] = o;
            }
            asm volatile("" ::: "memory");  // chunk by chunk (registers: see the multiply below)
        }
    } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int cc = (int)threadIdx.x + GEMV_WAVES * 64 * i;
            if (cc < nchunk) x_lds[cc] = xi[i];
        }
    }
    __syncthreads();
    if (idle) return;
    float acc[NW_MATS];
#pragma unroll
    for (int m = 0; m < NW_MATS; ++m) acc[m] = 0.f;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int cu = lane + 64 * (t * UN + u);
            const bool past = t * UN + u == UN * TRIPS - 1 && cu >= nchunk;
            float xf[8];
            unpack8f(x_lds[past ? lane : cu], xf);
            if (past) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < NW_MATS; ++m) {
                float wf[8];
                unpack8f(wv[m][t][u], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[m] = fmaf(wf[j], xf[j], acc[m]);
            }
            // (the next chunk's LDS read and conversions stay behind this chunk's sums)
            if constexpr (NW_MATS == 2) asm volatile("" : "+v"(acc[0]), "+v"(acc[1])::"memory");
            else asm volatile("" : "+v"(acc[0])::"memory");
        }
#pragma unroll
    for (int m = 0; m < NW_MATS; ++m) acc[m] = wave_sum64(acc[m]);
    if (lane == 0) {
        float v = round_bf(NW_MATS == 1 && p.bias ? acc[0] + bf2f(biasv) : acc[0]);  // the nn.Linear output as a bf16 tensor
        if constexpr (NW_MATS == 2) {
            const float g = v;
            const float sg = round_bf(g / (1.f + __expf(-g)));  // F.silu on a bf16 tensor
            v = sg * round_bf(acc[1]);
        } else if (p.res) {
            v = v + bf2f(resv);  // x + linear(...), rounded once more below
        }
        if (p.y32) p.y32[n] = v;
        else p.y[n] = f2bf(v);
    }
}

// ---- rotary embedding of q and k at the current position + cache update ----
// q [H * hd], k [Hkv * hd], v [Hkv * hd] bf16; cos / sin tables [max_len, hd] bf16 (HF: emb = cat(freqs, freqs));
// caches [Hkv, max_len, hd]. q_embed = q * cos + rotate_half(q) * sin with bf16 rounding of each product and of the sum.
__global__ __launch_bounds__(256) void rope_cache_kernel(DecodeRopeArgs p) {
    const long pos = *p.pos;
    const int hd = p.hd, half = hd >> 1;
    const int total = (p.H + 2 * p.Hkv) * hd;
    const bf16_t* cs = p.cos + pos * hd;
    const bf16_t* sn = p.sin + pos * hd;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int head = i / hd, d = i - head * hd;
        if (head < p.H + p.Hkv) {
            const bf16_t* vec = head < p.H ? p.q + head * hd : p.k + (head - p.H) * hd;
            const float a = bf2f(vec[d]);
            const float b = d < half ? -bf2f(vec[d + half]) : bf2f(vec[d - half]);  // rotate_half
            const float r = round_bf(round_bf(a * bf2f(cs[d])) + round_bf(b * bf2f(sn[d])));
            // q is rotated in place AFTER every thread of its head has read: write to the separate output instead
            if (head < p.H) p.q_out[i] = f2bf(r);
            else p.k_cache[((size_t)(head - p.H) * p.max_len + pos) * hd + d] = f2bf(r);
        } else {
            const int hv = head - p.H - p.Hkv;
            p.v_cache[((size_t)hv * p.max_len + pos) * hd + d] = p.v[hv * hd + d];
        }
    }
}

// ---- prompt prefill: rotary embedding of S positions + cache fill, 16 bytes per lane ----
// One item = 8 elements d0 .. d0 + 7 of the first half of a (position, head) vector together with its partners d0 + hd / 2 ..
// (rotate_half pairs them), so q can be rotated in place; the v items are plain 16-byte copies into the cache.
__global__ __launch_bounds__(256) void prefill_rope_cache_kernel(PrefillRopeArgs p) {
    const int hc = p.hd >> 4;                       // 8-element chunks per half vector
    const int per_row = (p.H + p.Hkv) * hc + p.Hkv * 2 * hc;
    const long total = (long)p.S * per_row;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int s = (int)(g / per_row);
        int it = (int)(g - (long)s * per_row);
        const long pos = p.pos0 + s;
        if (it < (p.H + p.Hkv) * hc) {
            const int head = it / hc, d0 = (it - head * hc) * 8, half = p.hd >> 1;
            const bool is_q = head < p.H;
            const bf16_t* src = is_q ? p.q + (size_t)s * p.ldq + head * p.hd : p.k + (size_t)s * p.ldk + (head - p.H) * p.hd;
            float lo[8], hi[8], cl[8], ch[8], sl[8], sh[8];
            unpack8f(*(const u32x4*)(src + d0), lo);
            unpack8f(*(const u32x4*)(src + d0 + half), hi);
            unpack8f(*(const u32x4*)(p.cos + pos * p.hd + d0), cl);
            unpack8f(*(const u32x4*)(p.cos + pos * p.hd + d0 + half), ch);
            unpack8f(*(const u32x4*)(p.sin + pos * p.hd + d0), sl);
            unpack8f(*(const u32x4*)(p.sin + pos * p.hd + d0 + half), sh);
            u32x4 olo, ohi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // x * cos + rotate_half(x) * sin, every product and the sum rounded to bf16 (rope_cache_kernel)
                const int a = 2 * j, b = 2 * j + 1;
                olo[j] = pack2bf(round_bf(lo[a] * cl[a]) + round_bf(-hi[a] * sl[a]), round_bf(lo[b] * cl[b]) + round_bf(-hi[b] * sl[b]));
                ohi[j] = pack2bf(round_bf(hi[a] * ch[a]) + round_bf(lo[a] * sh[a]), round_bf(hi[b] * ch[b]) + round_bf(lo[b] * sh[b]));
            }
            bf16_t* dst = is_q ? p.q + (size_t)s * p.ldq + head * p.hd : p.k_cache + ((size_t)(head - p.H) * p.max_len + pos) * p.hd;
            *(u32x4*)(dst + d0) = olo;
            *(u32x4*)(dst + d0 + half) = ohi;
        } else {
            it -= (p.H + p.Hkv) * hc;
            const int hv = it / (2 * hc), d0 = (it - hv * 2 * hc) * 8;
            *(u32x4*)(p.v_cache + ((size_t)hv * p.max_len + pos) * p.hd + d0) = *(const u32x4*)(p.v + (size_t)s * p.ldk + hv * p.hd + d0);
        }
    }
}

// ---- prompt prefill: silu(gate) * up on materialised tensors (the decode step has it inside its GEMV pair) ----
__global__ __launch_bounds__(256) void silu_mul_kernel(SiluMulArgs p) {
    const long nchunk = p.n >> 3;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += (long)gridDim.x * 256) {
        float g[8], u[8];
        unpack8f(*(const u32x4*)(p.gate + c * 8), g);
        unpack8f(*(const u32x4*)(p.up + c * 8), u);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s0 = round_bf(g[2 * j] / (1.f + __expf(-g[2 * j])));  // F.silu on a bf16 tensor
            const float s1 = round_bf(g[2 * j + 1] / (1.f + __expf(-g[2 * j + 1])));
            o[j] = pack2bf(s0 * u[2 * j], s1 * u[2 * j + 1]);
        }
        *(u32x4*)(p.out + c * 8) = o;
    }
}

// ---- decode attention: grid (H, nsplit); 16 lanes per cache position (16 B of a 128-wide K / V row each; hd == 128) ----
// Each block takes positions [s * chunk, (s + 1) * chunk) of [0, pos]; every 16-lane group keeps a running (m, l, o[8]);
// groups and waves are merged through LDS; the block writes (m, l, o[hd]) to the workspace.
constexpr int DA_THREADS = 256;
constexpr int DA_SPLIT_STRIDE = 132;  // floats per (head, range) partial of the split-only attention launch: [o 128][m][l][2 pad]
constexpr int DA_UN = 4;  // cache positions per 16-lane group per trip: 2 x DA_UN 16-byte loads in flight per lane (8 / 12: 14.0 / 13.6 us against 14.0)

struct DaState {  // running (max, sum, out[8 dims of this lane]) of one 16-lane group
    float m, l, o[8];
};
MERV_DEVICE float da_dot16(const float (&qf)[8], const float (&kf)[8]) {  // q . k over the group's 128 dims
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = fmaf(qf[i], kf[i], d);
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
    return d;
}
// one position, from registers
MERV_DEVICE void da_one(DaState& st, const float (&qf)[8], const float (&kf)[8], const float (&vf)[8], float sc) {
    const float d = da_dot16(qf, kf) * sc;
    const float mn = fmaxf(st.m, d);
    const float a = __builtin_amdgcn_exp2f(st.m - mn), e = __builtin_amdgcn_exp2f(d - mn);
    st.l = st.l * a + e;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.o[i] = fmaf(st.o[i], a, e * vf[i]);
    st.m = mn;
}
// One trip of a 16-lane group: cache positions j, j + 16, ... (DA_UN of them) < jc. All K / V rows of a trip are requested before the
// first is used, and the trip's scores share one running-max update (a trip per position left the kernel waiting for one HBM round
// trip per position: 15 us per layer at 1050 positions for 17 MB of cache). da_request only issues the loads (past the range: a valid
// row, masked in da_consume), so that a caller can have the next trip in flight while it multiplies this one.
MERV_DEVICE void da_request(u32x4 (&kr)[DA_UN], u32x4 (&vr)[DA_UN], const bf16_t* Kc, const bf16_t* Vc, int j, int jc, int sub) {
    const int jv = j < jc ? j : 0;
#pragma unroll
    for (int u = 0; u < DA_UN; ++u) {
        const int ju = j + 16 * u < jc ? j + 16 * u : jv;
        kr[u] = *(const u32x4*)(Kc + (size_t)ju * 128 + sub * 8);
        vr[u] = *(const u32x4*)(Vc + (size_t)ju * 128 + sub * 8);
    }
}
MERV_DEVICE void da_consume(DaState& st, const float (&qf)[8], const u32x4 (&kr)[DA_UN], const u32x4 (&vr)[DA_UN], int j, int jc, float sc) {  // j < jc
    float d[DA_UN], mn = st.m;
#pragma unroll
    for (int u = 0; u < DA_UN; ++u) {
        float kf[8];
        unpack8f(kr[u], kf);
        const float dd = da_dot16(qf, kf) * sc;  // (unconditional: the shuffles stay out of a branch; a masked row is a valid row)
        d[u] = j + 16 * u < jc ? dd : -INFINITY;
        mn = fmaxf(mn, d[u]);
    }
    const float a = __builtin_amdgcn_exp2f(st.m - mn);  // position j itself is in range: mn is finite
    st.l *= a;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.o[i] *= a;
#pragma unroll
    for (int u = 0; u < DA_UN; ++u) {
        const float e = __builtin_amdgcn_exp2f(d[u] - mn);
        float vf[8];
        unpack8f(vr[u], vf);
        st.l += e;
#pragma unroll
        for (int i = 0; i < 8; ++i) st.o[i] = fmaf(e, vf[i], st.o[i]);
    }
    st.m = mn;
}
// cache positions j0 + g, j0 + g + 16, ... < jc
MERV_DEVICE void da_range(DaState& st, const float (&qf)[8], const bf16_t* Kc, const bf16_t* Vc, int j0, int jc, int g, int sub, float sc) {
    for (int j = j0 + g; j < jc; j += 16 * DA_UN) {
        u32x4 kr[DA_UN], vr[DA_UN];
        da_request(kr, vr, Kc, Vc, j, jc, sub);
        da_consume(st, qf, kr, vr, j, jc, sc);
    }
}

__global__ __launch_bounds__(DA_THREADS) void decode_attn_kernel(DecodeAttnArgs p) {
    __shared__ float sm_m[16], sm_l[16];
    __shared__ __attribute__((aligned(16))) float sm_o[16][128];
    const int h = blockIdx.x, s = blockIdx.y;
    const int hkv = h / (p.H / p.Hkv);
    const long npos = *p.pos + 1;  // positions 0 .. pos hold keys (the current token's k / v were just written)
    const int chunk = (int)((npos + p.nsplit - 1) / p.nsplit);
    const int j0 = s * chunk, j1 = (long)(j0 + chunk) < npos ? j0 + chunk : (int)npos;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;  // 16 lanes per position, 8 dims per lane
    const int g = wave * 4 + grp;                // 16 groups per block
    float qf[8];
    unpack8f(*(const u32x4*)(p.q + h * 128 + sub * 8), qf);
    const bf16_t* Kc = p.k_cache + (size_t)hkv * p.max_len * 128;
    const bf16_t* Vc = p.v_cache + (size_t)hkv * p.max_len * 128;
    DaState st;
    st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.o[i] = 0.f;
    const float sc = p.scale * 1.4426950408889634f;
    const long pos = npos - 1;
    // the token being decoded is taken last, by the group whose turn it is (the order the fused launch uses: same bits)
    da_range(st, qf, Kc, Vc, j0, j1 < pos ? j1 : (int)pos, g, sub, sc);
    if (pos >= j0 && pos < j1 && g == (int)((pos - j0) & 15)) {
        float kf[8], vf[8];
        unpack8f(*(const u32x4*)(Kc + (size_t)pos * 128 + sub * 8), kf);
        unpack8f(*(const u32x4*)(Vc + (size_t)pos * 128 + sub * 8), vf);
        da_one(st, qf, kf, vf, sc);
    }
    if (sub == 0) { sm_m[g] = st.m; sm_l[g] = st.l; }
#pragma unroll
    for (int i = 0; i < 8; ++i) sm_o[g][sub * 8 + i] = st.o[i];
    __syncthreads();
    if (threadIdx.x < 128) {
        const int d = threadIdx.x;
        float M = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) M = fmaxf(M, sm_m[q]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float w = sm_m[q] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(sm_m[q] - M);
            L = fmaf(w, sm_l[q], L);
            O = fmaf(w, sm_o[q][d], O);
        }
        float* ws = p.ws + ((size_t)h * p.nsplit + s) * (128 + 2);
        ws[d] = O;
        if (d == 0) { ws[128] = M; ws[129] = L; }
    }
}

__global__ __launch_bounds__(128) void decode_attn_merge_kernel(DecodeAttnArgs p) {
    const int h = blockIdx.x, d = threadIdx.x;
    const float* ws = p.ws + (size_t)h * p.nsplit * (128 + 2);
    float M = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, ws[s * 130 + 128]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float ms = ws[s * 130 + 128];
        const float w = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - M);
        L = fmaf(w, ws[s * 130 + 129], L);
        O = fmaf(w, ws[s * 130 + d], O);
    }
    p.out[h * 128 + d] = f2bf(O / L);
}

// ---- rotary + cache update + split attention + merge in one launch: grid (H, nsplit), 256 threads ----
// What the three launches above do, without the two kernel boundaries between them (each ~5 us of drain + ramp in a step that
// is one dependent chain): every block rotates its head's query itself (128 values); positions 0 .. pos-1 come from the cache;
// the token being decoded is rotated from the raw k / v by the one 16-lane group whose turn it is and used from registers --
// no block reads cache[pos] -- and the first head of each kv group stores it for the steps to come. The partials meet through
// the workspace: a block publishes (o, m, l), fences, and takes a ticket on its head's counter; the block that draws the last
// ticket merges all splits and resets the counter (so a captured graph replays). Same rounding points as the separate kernels.
// Measured (tools/probes/decode_kernels.py, 1050 positions): 13.7 us per layer with 8 splits of 256 threads; 17.9 / 24.0 with
// 16 / 32 splits, 16.5 with one 1024-thread block per head and no workspace round trip, 12.5-13.8 with 1024-thread blocks
// and 2-8 splits -- the launch is a chain of dependent memory round trips (position -> tables / q / cache rows -> partials ->
// ticket -> partials), not a bandwidth problem (17 MB of cache).
struct AttnLds {
    float m[16], l[16];
    __attribute__((aligned(16))) float o[16][128];
    unsigned ticket;
};
// h: head, s: position range.
// SPLIT_ONLY (round 4, stand-alone launch): the launch ends at the partials -- written with plain stores, visible at the kernel
// boundary -- and the o-projection launch merges them while its first weight trip is in flight (oproj_merge_kernel): no write-through,
// no vmcnt(0) + barrier + ticket round trip, no second pass over the partials inside this launch's dependency chain (9.0 against 13.8 us).
template <bool SPLIT_ONLY>
MERV_DEVICE void attn_fused_body(const DecodeAttnFusedArgs& p, const int h, const int s, AttnLds& lds) {
    const int grp_heads = p.H / p.Hkv;
    const int hkv = h / grp_heads;
    const long pos = *p.pos;
    const long npos = pos + 1;
    const int chunk = (int)((npos + p.nsplit - 1) / p.nsplit);
    const int j0 = s * chunk, j1 = (long)(j0 + chunk) < npos ? j0 + chunk : (int)npos;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int g = wave * 4 + grp;
    // Everything the block reads hangs on *pos alone, so all of it is requested at once (round 5; until then the launch was a chain of
    // round trips: position -> tables / q -> first trip of cache rows -> ... -> the new token's k / v in the last range's block):
    // the rotary tables' row, the head's raw q, the raw k / v of the token being decoded (used by ONE group of the last range, requested
    // by all: 32 B per lane) and then the group's first trip of cache rows; the rotation of q runs while the cache rows are in flight.
    bf16_t* Kc = p.k_cache + (size_t)hkv * p.max_len * 128;
    bf16_t* Vc = p.v_cache + (size_t)hkv * p.max_len * 128;
    const int jc = j1 < pos ? j1 : (int)pos;  // cached positions of this range end before the current token
    const int ntrips = jc > j0 ? (jc - j0 + 16 * DA_UN - 1) / (16 * DA_UN) : 0;  // block-uniform (a group past its last row skips the multiply)
    const u32x4 cosv = *(const u32x4*)(p.cos + pos * 128 + sub * 8);
    const u32x4 sinv = *(const u32x4*)(p.sin + pos * 128 + sub * 8);
    const u32x4 q_own = *(const u32x4*)(p.q + (h * 128 + sub * 8));
    const u32x4 k_own = *(const u32x4*)(p.k + (hkv * 128 + sub * 8));
    const u32x4 vraw = *(const u32x4*)(p.v + (hkv * 128 + sub * 8));
    int j = j0 + g;
    u32x4 kr[DA_UN], vr[DA_UN];
    da_request(kr, vr, Kc, Vc, j, jc, sub);
    __builtin_amdgcn_sched_barrier(0);
    // rotary at *pos: lane `sub` owns dims 8 sub .. 8 sub + 7, their rotate_half partners sit in lane sub ^ 8 of the group
    float cf[8], sf[8];
    unpack8f(cosv, cf);
    unpack8f(sinv, sf);
    auto rotary = [&](const u32x4& own, float (&r)[8]) {
        u32x4 oth;
#pragma unroll
        for (int q = 0; q < 4; ++q) oth[q] = __shfl_xor(own[q], 8, 64);
        float a[8], b[8];
        unpack8f(own, a);
        unpack8f(oth, b);
        const float sgn = sub < 8 ? -1.f : 1.f;  // rotate_half: (-x[d + half], x[d - half])
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = round_bf(round_bf(a[i] * cf[i]) + round_bf(sgn * b[i] * sf[i]));
    };
    float qf[8];
    rotary(q_own, qf);
    DaState st;
    st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.o[i] = 0.f;
    const float sc = p.scale * 1.4426950408889634f;
    // trip t + 1 is requested (unconditionally: a request behind a branch would make every wait of the multiply a vmcnt(0)) before trip t
    // is multiplied, into the other of two register sets (a copy would wait for the rows just requested); the last trip is multiplied
    // without a request behind it, so that no wave ends with loads nobody uses
    if (ntrips > 0) {
        u32x4 kn[DA_UN], vn[DA_UN];
        constexpr int TRIP = 16 * DA_UN;
        for (int t = 0;; t += 2, j += 2 * TRIP) {
            if (t + 1 >= ntrips) {
                if (j < jc) da_consume(st, qf, kr, vr, j, jc, sc);
                break;
            }
            da_request(kn, vn, Kc, Vc, j + TRIP, jc, sub);
            if (j < jc) da_consume(st, qf, kr, vr, j, jc, sc);
            if (t + 2 >= ntrips) {
                if (j + TRIP < jc) da_consume(st, qf, kn, vn, j + TRIP, jc, sc);
                break;
            }
            da_request(kr, vr, Kc, Vc, j + 2 * TRIP, jc, sub);
            if (j + TRIP < jc) da_consume(st, qf, kn, vn, j + TRIP, jc, sc);
        }
    }
    if (pos >= j0 && pos < j1 && g == (int)((pos - j0) & 15)) {  // uniform per 16-lane group: the shuffles inside stay in the group
        float kf[8], vf[8];
        rotary(k_own, kf);
        unpack8f(vraw, vf);
        da_one(st, qf, kf, vf, sc);
        if (h % grp_heads == 0) {
            u32x4 kw;
#pragma unroll
            for (int q = 0; q < 4; ++q) kw[q] = pack2bf(kf[2 * q], kf[2 * q + 1]);
            *(u32x4*)(Kc + (size_t)pos * 128 + sub * 8) = kw;
            *(u32x4*)(Vc + (size_t)pos * 128 + sub * 8) = vraw;
        }
    }
    if (sub == 0) { lds.m[g] = st.m; lds.l[g] = st.l; }
#pragma unroll
    for (int i = 0; i < 8; ++i) lds.o[g][sub * 8 + i] = st.o[i];
    __syncthreads();
    // partial layout: [o 128][m][l] per (head, range); the split-only launch pads the record to 132 floats so that the merging
    // o-projection reads 16-byte aligned rows
    constexpr int PS = SPLIT_ONLY ? DA_SPLIT_STRIDE : 130;
    float* ws_h = p.ws + (size_t)h * p.nsplit * PS;
    if (threadIdx.x < 128) {
        const int d = threadIdx.x;
        float M = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) M = fmaxf(M, lds.m[q]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float w = lds.m[q] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(lds.m[q] - M);
            L = fmaf(w, lds.l[q], L);
            O = fmaf(w, lds.o[q][d], O);
        }
        // the partials travel between blocks (possibly between XCDs, each with its own L2) inside one launch: device-scope
        // relaxed atomics write through / read past the non-coherent cache levels, which costs nothing beside an ordinary
        // store here, whereas a device-scope FENCE writes back and invalidates the whole L2 (measured: +15 us per layer)
        float* ws = ws_h + (size_t)s * PS;
        if constexpr (SPLIT_ONLY) {
            ws[d] = O;
            if (d == 0) { ws[128] = M; ws[129] = L; }
        } else {
            __hip_atomic_store(ws + d, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == 0) {
                __hip_atomic_store(ws + 128, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ws + 129, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if constexpr (SPLIT_ONLY) return;
    // publish (every store of this block acknowledged), then take a ticket; the last arrival of this head merges.
    // This is the write-through form of the in-launch hand-off (cdna_hip_programming.md Guideline 16, R1 in its counter form, and
    // MI355X_MICROARCH.md "Valid forms", first table row): every partial is stored sc1 (relaxed agent-scope atomic stores ARE
    // global_store ... sc1), every storing wave waits vmcnt(0) -- as ASM: hipcc can drop a builtin wait it believes redundant
    // (Guideline 16, Pitfall 12) -- the workgroup meets at its barrier, ONE lane adds to the head's counter (agent-scope atomic),
    // and the merging workgroup -- the one whose add returned nsplit - 1, told to its other waves through LDS behind a barrier --
    // reads every partial with sc1 loads (relaxed agent-scope atomic loads). No release / acquire fence is needed in this form and
    // none is paid for (a fence pair cost +15 us per layer). Invariants: (1) every partial is written and read ONLY through these
    // atomics, (2) one workspace serves one stream -- launches that share it must not overlap (HipDecoder owns one per decoder and
    // zeroes it in prefill()). tools/probes/decode_attention_stress.py hammers it under uneven load.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* counter = (unsigned*)(p.ws + (size_t)p.H * p.nsplit * (128 + 2)) + h * 32;  // one 128-byte line per head
    if (threadIdx.x == 0) lds.ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (lds.ticket != (unsigned)(p.nsplit - 1)) return;
    if (threadIdx.x < 128) {
        const int d = threadIdx.x;
        auto ld = [&](int i) { return __hip_atomic_load(ws_h + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        float M = -INFINITY;
        for (int q = 0; q < p.nsplit; ++q) M = fmaxf(M, ld(q * 130 + 128));
        float L = 0.f, O = 0.f;
        for (int q = 0; q < p.nsplit; ++q) {
            const float ms = ld(q * 130 + 128);
            const float w = ms == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ms - M);
            L = fmaf(w, ld(q * 130 + 129), L);
            O = fmaf(w, ld(q * 130 + d), O);
        }
        p.out[h * 128 + d] = f2bf(O / L);
    }
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(DA_THREADS) void decode_attn_fused_kernel(DecodeAttnFusedArgs p) {
    __shared__ AttnLds lds;
    attn_fused_body<false>(p, blockIdx.x, blockIdx.y, lds);
}
__global__ __launch_bounds__(DA_THREADS) void decode_attn_split_kernel(DecodeAttnFusedArgs p) {
    __shared__ AttnLds lds;
    if (blockIdx.y >= (unsigned)p.nsplit) {
        // prefetch role (blocks behind the H x nsplit attention blocks; their linear id is a multiple of 8 further on, so prefetch block b shares
        // its XCD -- its L2 -- with block b of the next launch): read-only pass over that block's slice of the next launch's weights
        const int b = (blockIdx.y - p.nsplit) * p.H + blockIdx.x;
        if (b >= p.prefetch_blocks) return;
        const char* base = (const char*)p.prefetch + (size_t)b * p.prefetch_block_bytes;
        u32x4 acc = {0u, 0u, 0u, 0u};
        for (int off = threadIdx.x * 16; off < p.prefetch_block_bytes; off += DA_THREADS * 16 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = off + u * DA_THREADS * 16;
                v[u] = o < p.prefetch_block_bytes ? *(const u32x4*)(base + o) : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
        // (the loads must not be dead code: a store that never happens -- bf16 weights do not XOR to this pattern in all four words)
        if (acc[0] == 0x7fc07fc1u && acc[1] == 0x7fc17fc0u && acc[2] == 0x7fc27fc3u && acc[3] == 0x7fc37fc2u) p.ws[0] = 0.f;
        return;
    }
    attn_fused_body<true>(p, blockIdx.x, blockIdx.y, lds);
}

// ---- greedy decoding inside the captured step: token = argmax(logits) (torch.argmax's rule: the first maximum; a NaN wins), ----
// ---- tok[0] <- token, out_tokens[*pos - pos0] <- token, *pos += 1: what the host loop did with three PyTorch kernels per token ----
__global__ __launch_bounds__(1024) void greedy_advance_kernel(const float* logits, int V, long* tok, long* pos, long* out_tokens, long pos0) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    bool nan = false;
    auto take = [&](float v, int i) {  // ascending i per thread: a strict > keeps the first maximum
        if (v != v) { if (!nan) { nan = true; idx = i; } }
        else if (!nan && (v > best || idx == 0x7fffffff)) { best = v; idx = i; }
    };
    // 16 bytes per lane, eight loads in flight before the first compare (one block reads 128 KB of logits: latency, not bandwidth)
    const int nvec = ((uintptr_t)logits & 15) == 0 ? V >> 2 : 0;
    for (int b = threadIdx.x; b < nvec; b += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = b + u * 1024;
            v[u] = c < nvec ? *(const float4*)(logits + 4 * c) : float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = b + u * 1024;
            if (c < nvec) { take(v[u].x, 4 * c); take(v[u].y, 4 * c + 1); take(v[u].z, 4 * c + 2); take(v[u].w, 4 * c + 3); }
        }
    }
    for (int i = 4 * nvec + threadIdx.x; i < V; i += 1024) take(logits[i], i);
    auto better = [](bool na, float va, int ia, bool nb, float vb, int ib) {  // is (b) ahead of (a)?
        if (na != nb) return nb;
        if (na) return ib < ia;
        return vb > va || (vb == va && ib < ia);
    };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        const bool on = __shfl_xor((int)nan, o, 64) != 0;
        if (better(nan, best, idx, on, ov, oi)) { best = ov; idx = oi; nan = on; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[wave] = nan ? NAN : best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = bv[0]; int ix = bi[0]; bool n = v != v;
        for (int w = 1; w < 16; ++w) {
            const bool wn = bv[w] != bv[w];
            if (better(n, v, ix, wn, bv[w], bi[w])) { v = bv[w]; ix = bi[w]; n = wn; }
        }
        const long p = *pos;
        tok[0] = ix;
        if (out_tokens) out_tokens[p - pos0] = ix;
        *pos = p + 1;
    }
}

// ---- temperature sampling inside the captured step (round 6): token ~ softmax(logits / T), drawn as argmax_i(logits_i / T + g_i) with ----
// ---- independent standard Gumbel g_i = -log(-log(u_i)) (the Gumbel-max trick: exactly the categorical distribution torch.multinomial ----
// ---- draws from, without a normalising pass or a prefix sum). u_i comes from Philox4x32-10 keyed by the call's seed with counter ----
// ---- (i / 4, position): a counter-based stream, so the captured step replays for every position with fresh numbers and a (seed, ----
// ---- position) pair always gives the same token. Temperature, seed and the end-of-sequence rule live in DEVICE memory (`params`), ----
// ---- written by the host before a generation: nothing about them is baked into the graph. Replaces the host loop's ----
// ---- divide / softmax / multinomial / copy kernels and its one host round trip per token (merv.py:818-825 -> HF GenerationMixin.sample). ----
struct SampleParams {     // 32 bytes, device memory
    float inv_temperature;
    int eos_id;           // < 0: no end-of-sequence token
    unsigned long seed;
    long min_new_tokens;  // the end-of-sequence token cannot be drawn as new token number < min_new_tokens ...
    long first_pos;       // ... where the step at position first_pos draws token number 1 (number 0 was drawn from the prompt's logits)
};
MERV_DEVICE void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c[0], hi0 = __umulhi(0xD2511F53u, c[0]);
        const uint32_t lo1 = 0xCD9E8D57u * c[2], hi1 = __umulhi(0xCD9E8D57u, c[2]);
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
MERV_DEVICE float gumbel_from_bits(uint32_t x) {
    const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1): 24 bits, never 0 or 1
    return -__logf(-__logf(u));
}
__global__ __launch_bounds__(1024) void sample_advance_kernel(const float* logits, int V, const SampleParams* prm, long* tok, long* pos,
                                                              long* out_tokens, long pos0) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const long p = *pos;
    const float inv_t = prm->inv_temperature;
    const int eos = (prm->eos_id >= 0 && p - prm->first_pos + 1 < prm->min_new_tokens) ? prm->eos_id : -1;  // barred this step (HF MinNewTokensLength)
    const uint32_t k0 = (uint32_t)prm->seed, k1 = (uint32_t)(prm->seed >> 32);
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto take = [&](float v, int i) {  // ascending i per thread; a NaN logit never wins (torch.multinomial would raise on it)
        if (i != eos && (v > best || idx == 0x7fffffff)) { best = v; idx = i; }
    };
    // 16 bytes per lane, eight loads in flight before the first Philox block (as greedy_advance_kernel: one block reads 128 KB of logits -- latency, not
    // bandwidth; one load per round of the loop made the launch a chain of eight round trips: 18.1 us against 10.8 for the argmax)
    const int nvec = ((uintptr_t)logits & 15) == 0 ? V >> 2 : 0;
    for (int b = threadIdx.x; b < nvec; b += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = b + u * 1024;
            v[u] = q < nvec ? *(const float4*)(logits + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = b + u * 1024;
            if (q < nvec) {
                uint32_t c[4] = {(uint32_t)q, (uint32_t)p, (uint32_t)((unsigned long)p >> 32), 0x4D455256u};
                philox4x32_10(c, k0, k1);
                const float l[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) take(l[j] == l[j] ? fmaf(l[j], inv_t, gumbel_from_bits(c[j])) : -INFINITY, 4 * q + j);
            }
        }
    }
    for (int q = nvec + threadIdx.x; 4 * q < V; q += 1024) {  // the tail (and everything, for an unaligned buffer): the same numbers for the same (q, j)
        uint32_t c[4] = {(uint32_t)q, (uint32_t)p, (uint32_t)((unsigned long)p >> 32), 0x4D455256u};
        philox4x32_10(c, k0, k1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * q + j;
            if (i < V) {
                const float l = logits[i];
                take(l == l ? fmaf(l, inv_t, gumbel_from_bits(c[j])) : -INFINITY, i);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = bv[0]; int ix = bi[0];
        for (int w = 1; w < 16; ++w)
            if (bv[w] > v || (bv[w] == v && bi[w] < ix)) { v = bv[w]; ix = bi[w]; }
        if (ix == 0x7fffffff) ix = 0;  // (every logit NaN or barred: cannot happen with V > 1)
        tok[0] = ix;
        if (out_tokens) out_tokens[p - pos0] = ix;
        *pos = p + 1;
    }
}

// ---- o-projection that merges the attention's split partials on the way in: y = res + W_o . merge(ws) ----
// One workgroup of 8 waves per 16 output rows (256 workgroups at D = 4096: one per CU, two rows per wave). Every lane first requests
// the partials of its chunk and then its whole first weight trip (ROWS x 8 chunks of 16 B); the workgroup merges the H x nsplit partials
// once -- thread t the 8 values 8 t .. 8 t + 7 of the attention output, with the arithmetic of attn_fused_body's merging block, rounded
// to bf16 -- into LDS (waiting for the partials alone: the merge runs while the weights are in flight), and after one barrier the
// GEMV reads x from LDS. Per row the products are accumulated in the order of gemv_body<1, 8, false, 1> (the plain o-projection's
// configuration): the result is bit-identical to merv_decode_attention_fused + merv_decode_gemv. Workgroup 0 also stores the
// merged vector (the separate path's attention output).
constexpr int OM_WAVES = 8, OM_ROWS = 2, OM_UN = 8;
template <bool EXACT>  // whole trips (K / 8 a multiple of 512 chunks) and whole blocks: no clamps, no zeroed chunks (as gemv_body)
__global__ __launch_bounds__(OM_WAVES * 64) void oproj_merge_kernel(DecodeOprojMergeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char om_smem[];
    bf16_t* x_lds = (bf16_t*)om_smem;  // [K]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * OM_WAVES + wave) * OM_ROWS;
    const int K = p.H * 128, nchunk = K >> 3;
    const bf16_t* wrow[OM_ROWS];
#pragma unroll
    for (int r = 0; r < OM_ROWS; ++r) wrow[r] = p.W + (size_t)(EXACT || n0 + r < p.N ? n0 + r : p.N - 1) * K;
    u32x4 wv[OM_ROWS][OM_UN];
    auto issue_w = [&](int c) {
        if constexpr (!EXACT) c = c < nchunk ? c : 0;
#pragma unroll
        for (int u = 0; u < OM_UN; ++u) {
            const int cu = EXACT || c + 64 * u < nchunk ? c + 64 * u : c;
#pragma unroll
            for (int r = 0; r < OM_ROWS; ++r) wv[r][u] = __builtin_nontemporal_load((const u32x4*)(wrow[r] + cu * 8));
        }
    };
    bf16_t resv[OM_ROWS];  // requested before the weights (as gemv_body: raw, unconditional; not one more round trip at the tail)
    const bf16_t* res_p = p.res ? p.res : p.W;  // (absent: any valid address; the decoder always passes the residual)
#pragma unroll
    for (int r = 0; r < OM_ROWS; ++r) resv[r] = res_p[p.res ? (EXACT || n0 + r < p.N ? n0 + r : p.N - 1) : 0];
    int c = lane;
    // merge: 8 consecutive values per thread, all of head (8 t) / 128. The decoder's geometry (8 ranges; K = 4096: one chunk per thread)
    // requests its partials BEFORE the weight trip and waits for them alone: vmcnt counts in order, so partials requested behind the
    // weights (until round 5) were merged only after the whole trip had arrived -- the merge, the LDS image and the barrier then stood
    // between the last weight byte and the first multiply (10.0 us against 7.9 for the plain o-projection).
    constexpr int PS = DA_SPLIT_STRIDE;
    const bool eight = p.nsplit == 8;
    auto merge8 = [&](int t, const float2 (&ml)[8], const float4 (&oa)[8], const float4 (&ob)[8]) {
        float M = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) M = fmaxf(M, ml[q].x);
        float L = 0.f, O[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float w = ml[q].x == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ml[q].x - M);
            L = fmaf(w, ml[q].y, L);
            O[0] = fmaf(w, oa[q].x, O[0]); O[1] = fmaf(w, oa[q].y, O[1]); O[2] = fmaf(w, oa[q].z, O[2]); O[3] = fmaf(w, oa[q].w, O[3]);
            O[4] = fmaf(w, ob[q].x, O[4]); O[5] = fmaf(w, ob[q].y, O[5]); O[6] = fmaf(w, ob[q].z, O[6]); O[7] = fmaf(w, ob[q].w, O[7]);
        }
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack2bf(O[2 * j] / L, O[2 * j + 1] / L);
        *(u32x4*)(x_lds + t * 8) = pk;
        if (blockIdx.x == 0 && p.attn_out) *(u32x4*)(p.attn_out + t * 8) = pk;
    };
    auto request8 = [&](int t, float2 (&ml)[8], float4 (&oa)[8], float4 (&ob)[8]) {
        const int h = t >> 4, d0 = (t & 15) * 8;
        const float* ws_h = p.ws + (size_t)h * 8 * PS;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ml[q] = *(const float2*)(ws_h + q * PS + 128);
            oa[q] = *(const float4*)(ws_h + q * PS + d0);
            ob[q] = *(const float4*)(ws_h + q * PS + d0 + 4);
        }
    };
    {
        float2 ml[8];
        float4 oa[8], ob[8];
        if (eight) request8(threadIdx.x < nchunk ? threadIdx.x : 0, ml, oa, ob);
        issue_w(c);
        __builtin_amdgcn_sched_barrier(0);
        if (eight && (int)threadIdx.x < nchunk) merge8(threadIdx.x, ml, oa, ob);
    }
    for (int t = threadIdx.x + (eight ? OM_WAVES * 64 : 0); t < nchunk; t += OM_WAVES * 64) {
        if (eight) {
            float2 ml[8];
            float4 oa[8], ob[8];
            request8(t, ml, oa, ob);
            merge8(t, ml, oa, ob);
            continue;
        }
        const int h = t >> 4, d0 = (t & 15) * 8;
        const float* ws_h = p.ws + (size_t)h * p.nsplit * PS;
        float L = 0.f, O[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float M = -INFINITY;
        for (int q = 0; q < p.nsplit; ++q) M = fmaxf(M, ws_h[q * PS + 128]);
        for (int q = 0; q < p.nsplit; ++q) {
            const float2 ml = *(const float2*)(ws_h + q * PS + 128);
            const float w = ml.x == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ml.x - M);
            L = fmaf(w, ml.y, L);
            const float4 a = *(const float4*)(ws_h + q * PS + d0), b = *(const float4*)(ws_h + q * PS + d0 + 4);
            O[0] = fmaf(w, a.x, O[0]); O[1] = fmaf(w, a.y, O[1]); O[2] = fmaf(w, a.z, O[2]); O[3] = fmaf(w, a.w, O[3]);
            O[4] = fmaf(w, b.x, O[4]); O[5] = fmaf(w, b.y, O[5]); O[6] = fmaf(w, b.z, O[6]); O[7] = fmaf(w, b.w, O[7]);
        }
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack2bf(O[2 * j] / L, O[2 * j + 1] / L);
        *(u32x4*)(x_lds + t * 8) = pk;
        if (blockIdx.x == 0 && p.attn_out) *(u32x4*)(p.attn_out + t * 8) = pk;
    }
    __syncthreads();
    float acc[OM_ROWS];
#pragma unroll
    for (int r = 0; r < OM_ROWS; ++r) acc[r] = 0.f;
    for (; c < nchunk; c += 64 * OM_UN) {
#pragma unroll
        for (int u = 0; u < OM_UN; ++u) {
            float xf[8];
            const int cu = c + 64 * u;
            unpack8f(*(const u32x4*)(x_lds + (EXACT || cu < nchunk ? cu : 0) * 8), xf);
            if (!EXACT && cu >= nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < OM_ROWS; ++r) {
                float wf[8];
                unpack8f(wv[r][u], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r] = fmaf(wf[j], xf[j], acc[r]);
            }
        }
        if (c + 64 * OM_UN < nchunk) issue_w(c + 64 * OM_UN);
    }
#pragma unroll
    for (int r = 0; r < OM_ROWS; ++r) acc[r] = wave_sum64(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < OM_ROWS; ++r) {
            const int n = n0 + r;
            if (n >= p.N) break;
            float v = round_bf(acc[r]);                 // the nn.Linear output as a bf16 tensor
            if (p.res) v = v + bf2f(resv[r]);           // x + o_proj(...), rounded once more below
            p.y[n] = f2bf(v);
        }
    }
}

}  // namespace

hipError_t launch_decode_rmsnorm(const DecodeRmsArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.D % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_add_rmsnorm(const DecodeRmsArgs& a, const bf16_t* delta, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.D % 8 != 0 || a.D > 8192) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(a.rows), dim3(256), 0, s, a, delta);
    return hipGetLastError();
}

template <int ROWS, int UN, int UN_PAIR = (UN > 4 ? 4 : UN)>
static hipError_t launch_gemv_cfg(const DecodeGemvArgs& a, hipStream_t s) {
    const int rows_per_block = ROWS * GEMV_WAVES;
    dim3 grid((a.N + a.Nb + a.Nc + rows_per_block - 1) / rows_per_block);
    const dim3 blk(GEMV_WAVES * 64);
    // EXACT: whole trips and whole blocks (every matrix of a q / k / v launch a multiple of the block's rows)
    const int nchunk = a.K >> 3;
    const bool rows_whole = a.N % rows_per_block == 0 && a.Nb % rows_per_block == 0 && a.Nc % rows_per_block == 0;
    if (a.W2) {
        const bool exact = rows_whole && nchunk % (64 * UN_PAIR) == 0;
        if (a.norm_w) {
            if (exact) hipLaunchKernelGGL((gemv_kernel<2, UN_PAIR, true, ROWS, true>), grid, blk, 2 * a.K, s, a);
            else hipLaunchKernelGGL((gemv_kernel<2, UN_PAIR, true, ROWS>), grid, blk, 2 * a.K, s, a);
        } else {
            if (exact) hipLaunchKernelGGL((gemv_kernel<2, UN_PAIR, false, ROWS, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((gemv_kernel<2, UN_PAIR, false, ROWS>), grid, blk, 0, s, a);
        }
    } else {
        const bool exact = rows_whole && nchunk % (64 * UN) == 0;
        if (a.norm_w) {
            if (exact) hipLaunchKernelGGL((gemv_kernel<1, UN, true, ROWS, true>), grid, blk, 2 * a.K, s, a);
            else hipLaunchKernelGGL((gemv_kernel<1, UN, true, ROWS>), grid, blk, 2 * a.K, s, a);
        } else {
            if (exact) hipLaunchKernelGGL((gemv_kernel<1, UN, false, ROWS, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((gemv_kernel<1, UN, false, ROWS>), grid, blk, 0, s, a);
        }
    }
    return hipGetLastError();
}

hipError_t launch_decode_gemv(const DecodeGemvArgs& a, hipStream_t s) {
    if (a.N <= 0 || a.K <= 0 || a.K % 8 != 0) return hipErrorInvalidValue;
    if (a.norm_w && a.K > 8 * GEMV_XN_CHUNKS) return hipErrorInvalidValue;  // the fused norm keeps the normalised input in LDS (32 KB)
    if (a.Nb > 0 || a.Nc > 0) {
        if (a.W2 || a.res || a.y32 || a.N % 2 || a.Nb % 2 || a.Nc % 2 || (a.Nb > 0 && (!a.Wb || !a.yb)) ||
            (a.Nc > 0 && (!a.Wc || !a.yc || a.Nb <= 0)))
            return hipErrorInvalidValue;
    }
    // Rows per wave x 16-byte chunks per lane and trip (every configuration multiplies a lane's chunks in the same order: same bits).
    // Rounds 1-4 (tools/probes/decode_kernels.py, MERV_GEMV_CFG sweeps): 2 x 4 everywhere, then 1 x 8 for the plain projections of <= 16384
    // rows (o_proj 7.9 -> 7.8 us, down_proj 20.0 -> 18.6) -- with the RMSNorm fused in, one row per wave was slower while EVERY WAVE
    // reduced mean(x^2) itself (22.1 -> 29.8 us). Since round 4 the block reduces it once (a quarter per wave), and a stand-alone
    // sweep (tools/probes/gemv_dma_probe.hip, round 5) has one row x 8 chunks ahead of two rows x 4 on every plain shape, lm_head
    // included (38.3 against 41.2 us). The hooks below select per class: MERV_GEMV_CFG (all launches, "<rows><chunks>"), or
    // MERV_GEMV_CFG_NORM / _PAIR / _BIG for the launches with a fused norm / the gate-up pair / more than 16384 rows.
    static const char* cfg = merv_tuning_env("MERV_GEMV_CFG");
    static const char* cfg_norm = merv_tuning_env("MERV_GEMV_CFG_NORM");
    static const char* cfg_pair = merv_tuning_env("MERV_GEMV_CFG_PAIR");
    static const char* cfg_big = merv_tuning_env("MERV_GEMV_CFG_BIG");
    const long rows_total = (long)a.N + a.Nb + a.Nc;
    // Defaults (round 5, with the block-level norm; q / k / v 20.3 -> 18.5 us, gate / up 32.3 -> 30.7, step 2.99 -> 2.86 ms): one row x 4 chunks
    // for the launches with a fused norm and for the gate-up pair (1 x 8 there: 20.6 / 42.5 us -- 200 registers), 1 x 8 for the plain
    // projections of <= 16384 rows, 2 x 4 otherwise.
    int rows = (a.W2 || a.norm_w || rows_total <= 16384) ? 1 : 2, un = (rows == 1 && !a.W2 && !a.norm_w) ? 8 : 4;
    const char* c = a.W2 ? (cfg_pair ? cfg_pair : cfg) : a.norm_w ? (cfg_norm ? cfg_norm : cfg) : rows_total > 16384 ? (cfg_big ? cfg_big : cfg) : cfg;
    const bool hooked = c && c[0] && c[1] && !c[2];
    if (hooked) {
        const int r = c[0] - '0', u = c[1] - '0';
        if ((r == 2 && u == 4) || (r == 1 && (u == 4 || u == 8))) { rows = r; un = u; }
    }
    // Launches whose row is one trip of 8 chunks per lane (K = 4096) or two of 11 (K = 11008: down projection) take the x-in-LDS,
    // whole-row-in-flight form (round 5, tools/sessions/gpu_r5_gemv_xlds.sh): the plain launches of <= 16384 rows (down 18.5 -> 16.9 us,
    // plain o 7.8 -> 7.3) and the gate-up pair (30.3 -> 29.4); q / k / v with the fused norm measured 18.3 -> 18.6 and stays on gemv_body,
    // lm_head is a tie. MERV_GEMV_XLDS (probe hook): "0" never, "1" the plain launches only, "2" plain launches of any size, "3" the
    // default, "4" every launch the form fits.
    static const char* xlds = merv_tuning_env("MERV_GEMV_XLDS");
    const char xmode = xlds && xlds[0] ? xlds[0] : '3';
    const bool xplain = !a.W2 && !a.norm_w;
    const bool xsize = rows_total <= 16384 || xmode == '4' || (xmode == '2' && xplain);
    const bool xclass = xplain ? xmode >= '1' : a.W2 ? xmode >= '3' : xmode >= '4';
    if (!hooked && xsize && xclass && (!a.norm_w || a.K <= 8192)) {
        const dim3 grid((unsigned)((rows_total + GEMV_WAVES - 1) / GEMV_WAVES)), blk(GEMV_WAVES * 64);
        const int nchunk = a.K >> 3;
        const bool one = nchunk > 448 && nchunk <= 512, two = nchunk > 1344 && nchunk <= 1408 && xplain;
        if (one || two) {
            if (two) hipLaunchKernelGGL((gemv_xlds_kernel<11, 2, 1, false>), grid, blk, 2 * a.K, s, a);
            else if (a.W2 && a.norm_w) hipLaunchKernelGGL((gemv_xlds_kernel<8, 1, 2, true, 2>), grid, blk, 2 * a.K, s, a);  // (K <= 4096 here)
            else if (a.W2) hipLaunchKernelGGL((gemv_xlds_kernel<8, 1, 2, false>), grid, blk, 2 * a.K, s, a);
            else if (a.norm_w) hipLaunchKernelGGL((gemv_xlds_kernel<8, 1, 1, true, 2>), grid, blk, 2 * a.K, s, a);
            else hipLaunchKernelGGL((gemv_xlds_kernel<8, 1, 1, false>), grid, blk, 2 * a.K, s, a);
            return hipGetLastError();
        }
    }
    if (rows == 1 && un == 8 && !hooked && !a.W2 && !a.norm_w) {
        // a trip is 64 lanes x UN chunks and the last one is padded with clamped (wasted) loads: K = 11008 is 1376 chunks = 3 trips
        // of 512 with 160 wasted, or 2 trips of 704 (UN = 11) with 32 -- same per-lane chunk order, so the same bits
        const int nchunk = a.K >> 3;
        const int waste8 = (nchunk + 511) / 512 * 512 - nchunk, waste11 = (nchunk + 703) / 704 * 704 - nchunk;
        if (waste11 + 64 < waste8) {
            hipLaunchKernelGGL((gemv_kernel<1, 11, false, 1>), dim3((a.N + GEMV_WAVES - 1) / GEMV_WAVES), dim3(GEMV_WAVES * 64), 0, s, a);
            return hipGetLastError();
        }
    }
    if (rows == 1 && un == 8) return launch_gemv_cfg<1, 8, 8>(a, s);  // (the pair: 2 matrices x 8 chunks = 16 loads per lane)
    if (rows == 1) return launch_gemv_cfg<1, 4>(a, s);
    return launch_gemv_cfg<2, 4>(a, s);
}

hipError_t launch_decode_rope_cache(const DecodeRopeArgs& a, hipStream_t s) {
    if (a.hd % 2 != 0 || a.H <= 0 || a.Hkv <= 0) return hipErrorInvalidValue;
    const int total = (a.H + 2 * a.Hkv) * a.hd;
    hipLaunchKernelGGL(rope_cache_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_prefill_rope_cache(const PrefillRopeArgs& a, hipStream_t s) {
    if (a.S <= 0) return hipSuccess;
    if (a.hd % 16 != 0 || a.H <= 0 || a.Hkv <= 0 || a.pos0 < 0 || a.pos0 + a.S > a.max_len) return hipErrorInvalidValue;
    if (a.ldq < a.H * a.hd || a.ldk < a.Hkv * a.hd || (a.ldq | a.ldk) % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)a.S * ((a.H + a.Hkv) * (a.hd / 16) + a.Hkv * (a.hd / 8));
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(prefill_rope_cache_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_silu_mul(const SiluMulArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (a.n % 8 != 0) return hipErrorInvalidValue;
    const long blocks = ((a.n >> 3) + 255) / 256;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decode_attention(const DecodeAttnArgs& a, hipStream_t s) {
    if (a.hd != 128 || a.H <= 0 || a.Hkv <= 0 || a.H % a.Hkv != 0 || a.nsplit <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_attn_kernel, dim3(a.H, a.nsplit), dim3(DA_THREADS), 0, s, a);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    hipLaunchKernelGGL(decode_attn_merge_kernel, dim3(a.H), dim3(128), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decode_attention_fused(const DecodeAttnFusedArgs& a, hipStream_t s) {
    if (a.hd != 128 || a.H <= 0 || a.Hkv <= 0 || a.H % a.Hkv != 0 || a.nsplit <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_attn_fused_kernel, dim3(a.H, a.nsplit), dim3(DA_THREADS), 0, s, a);
    return hipGetLastError();
}

size_t decode_attention_split_workspace_floats(int H, int nsplit) { return (size_t)H * nsplit * DA_SPLIT_STRIDE; }
hipError_t launch_decode_greedy_advance(const float* logits, int V, long* tok, long* pos, long* out_tokens, long pos0, hipStream_t s) {
    if (V <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(greedy_advance_kernel, dim3(1), dim3(1024), 0, s, logits, V, tok, pos, out_tokens, pos0);
    return hipGetLastError();
}
hipError_t launch_decode_sample_advance(const float* logits, int V, const void* params, long* tok, long* pos, long* out_tokens, long pos0, hipStream_t s) {
    if (V <= 0 || !params) return hipErrorInvalidValue;
    static_assert(sizeof(SampleParams) == 32, "merv_decode_sample_advance's params block is 32 bytes (include/merv_hip.h)");
    hipLaunchKernelGGL(sample_advance_kernel, dim3(1), dim3(1024), 0, s, logits, V, (const SampleParams*)params, tok, pos, out_tokens, pos0);
    return hipGetLastError();
}
hipError_t launch_decode_attention_split(const DecodeAttnFusedArgs& a, hipStream_t s) {
    if (a.hd != 128 || a.H <= 0 || a.Hkv <= 0 || a.H % a.Hkv != 0 || a.nsplit <= 0) return hipErrorInvalidValue;
    int extra = 0;
    if (a.prefetch) {
        if (a.prefetch_blocks <= 0 || a.prefetch_block_bytes <= 0 || a.prefetch_block_bytes % 16 != 0 || (a.H * a.nsplit) % 8 != 0) return hipErrorInvalidValue;
        extra = (a.prefetch_blocks + a.H - 1) / a.H;
    }
    hipLaunchKernelGGL(decode_attn_split_kernel, dim3(a.H, a.nsplit + extra), dim3(DA_THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_decode_oproj_merge(const DecodeOprojMergeArgs& a, hipStream_t s) {
    if (a.H <= 0 || a.nsplit <= 0 || a.N <= 0 || a.H * 256 > 64 * 1024) return hipErrorInvalidValue;  // x image in LDS: 256 B per head
    const int rows_per_block = OM_WAVES * OM_ROWS;
    const dim3 grid((a.N + rows_per_block - 1) / rows_per_block);
    if (a.N % rows_per_block == 0 && (a.H * 16) % (64 * OM_UN) == 0) hipLaunchKernelGGL(oproj_merge_kernel<true>, grid, dim3(OM_WAVES * 64), a.H * 256, s, a);
    else hipLaunchKernelGGL(oproj_merge_kernel<false>, grid, dim3(OM_WAVES * 64), a.H * 256, s, a);
    return hipGetLastError();
}

}  // namespace merv
