// Batch-1 token decode kernels for the LLM hand-off (merv/models/vidlms/merv.py:818-825 ->
// HF GenerationMixin's per-token forward of LlamaForCausalLM / MistralForCausalLM, transformers modeling_llama).
//
// The north_star keeps the LLM *prefill* on PyTorch-ROCm. A decode step, though, is ~1100 tiny PyTorch kernels per token
// (RMSNorm = 8 launches, rotary embedding = 10, ...): measured on MI355X, 6.1 of the 10.4 ms of a graph-replayed
// Llama-2-7B step are those launches, 4.7 ms the library's M = 1 GEMMs at 2.4-3.4 TB/s (tools/probes/decode_breakdown.py).
// These kernels are the step as 5 launches per layer (7 through the separate rotary / attention / merge entry points), every one
// a pure HBM stream:
//
//   rmsnorm_kernel         y = w * bf16(x * rsqrt(mean(x^2) + eps))                    (LlamaRMSNorm.forward)
//   gemv_kernel            y = bf16(W x) [+ residual]      W [N, K] bf16 streamed once (nn.Linear, M = 1)
//   gemv_silu_mul_kernel   y = bf16(silu(bf16(Wg x))) * bf16(Wu x)                      (LlamaMLP: act_fn(gate) * up)
//   rope_cache_kernel      q, k <- rotary(pos); K / V cache[pos] <- k, v              (apply_rotary_pos_emb + cache update)
//   decode_attn_kernel     one query per head against cache[0 .. pos], split over positions, (m, l, o) partials
//   decode_attn_merge_kernel  merges the partials                                      (flash-decoding)
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

// ---- RMSNorm: one wave per row ----
__global__ __launch_bounds__(256) void rmsnorm_kernel(DecodeRmsArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const bf16_t* x = p.x + (size_t)row * p.D;
    const int nchunk = p.D >> 3;
    float ss = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        float f[8];
        unpack8f(*(const u32x4*)(x + c * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
    }
    const float rstd = rsqrtf(wave_sum64(ss) / (float)p.D + p.eps);
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

// ---- GEMV: W [N, K] bf16 row-major streamed once; each wave owns ROWS output rows, lanes stride K in 16-byte chunks ----
constexpr int GEMV_WAVES = 4;  // waves per block

// NW_MATS 1: y = W x (+ res); 2: y = silu(Wg x) * (Wu x); NORM: RMSNorm of x fused in; GEMV_ROWS: output rows per wave (the x chunk
// is reused across them)
template <int NW_MATS, int UN, bool NORM, int GEMV_ROWS = 2>
__global__ __launch_bounds__(GEMV_WAVES * 64) void gemv_kernel(DecodeGemvArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: row pointers stay in SGPRs
    int n0 = (blockIdx.x * GEMV_WAVES + wave) * GEMV_ROWS;
    // several matrices in one launch (q / k / v): a wave's rows lie in ONE of them (row counts are multiples of GEMV_ROWS)
    if constexpr (NW_MATS == 1) {
        if (n0 >= p.N && p.Nb > 0) {
            n0 -= p.N;
            if (n0 < p.Nb) { p.W = p.Wb; p.y = p.yb; p.N = p.Nb; p.bias = p.bias_b; }
            else { n0 -= p.Nb; p.W = p.Wc; p.y = p.yc; p.N = p.Nc; p.bias = p.bias_c; }
        }
    }
    if (n0 >= p.N) return;
    const int nchunk = p.K >> 3;
    float acc[NW_MATS][GEMV_ROWS];
#pragma unroll
    for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) acc[m][r] = 0.f;
    const bf16_t* wrow[NW_MATS][GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
        const int n = n0 + r < p.N ? n0 + r : p.N - 1;
        wrow[0][r] = p.W + (size_t)n * p.K;
        if constexpr (NW_MATS == 2) wrow[1][r] = p.W2 + (size_t)n * p.K;
    }
    // UN chunks per lane per trip: UN x ROWS x NW_MATS weight loads of 16 B (and the x / norm-weight chunks they meet) are in
    // flight per lane before the first use. Every trip is a full batch: chunks past the row's end are clamped to a valid
    // address and meet x = 0 (a remainder loop of single loads costs one memory round trip per iteration: 3 us of the
    // K = 11008 launch).
    u32x4 wv[NW_MATS][GEMV_ROWS][UN], xv[UN], nv[NORM ? UN : 1];
    auto issue = [&](int c) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int cu = c + 64 * u < nchunk ? c + 64 * u : c;
#pragma unroll
            for (int m = 0; m < NW_MATS; ++m)
#pragma unroll
                for (int r = 0; r < GEMV_ROWS; ++r) wv[m][r][u] = __builtin_nontemporal_load((const u32x4*)(wrow[m][r] + cu * 8));
            xv[u] = *(const u32x4*)(p.x + cu * 8);
            if constexpr (NORM) nv[u] = *(const u32x4*)(p.norm_w + cu * 8);
        }
    };
    // fused RMSNorm: the wave reduces mean(x^2) over the whole input once (8 KB, L2-resident) BEHIND the first trip's weight
    // loads, then normalises each chunk it multiplies
    int c = lane;
    if (c < nchunk) issue(c);
    float rstd = 0.f;
    if constexpr (NORM) {
        float ss = 0.f;
        for (int cc = lane; cc < nchunk; cc += 64) {
            float f[8];
            unpack8f(*(const u32x4*)(p.x + cc * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
        }
        rstd = rsqrtf(wave_sum64(ss) / (float)p.K + p.norm_eps);
    }
    for (; c < nchunk; c += 64 * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float xf[8];
            unpack8f(xv[u], xf);
            if constexpr (NORM) {
                float wn[8];
                unpack8f(nv[u], wn);
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = round_bf(wn[j] * round_bf(xf[j] * rstd));
            }
            if (c + 64 * u >= nchunk) {
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
        if (c + 64 * UN < nchunk) issue(c + 64 * UN);
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
            float v = round_bf(NW_MATS == 1 && p.bias ? acc[0][r] + bf2f(p.bias[n]) : acc[0][r]);  // the nn.Linear output as a bf16 tensor
            if constexpr (NW_MATS == 2) {
                const float g = v;
                const float s = round_bf(g / (1.f + __expf(-g)));  // F.silu on a bf16 tensor
                v = s * round_bf(acc[1][r]);
            } else if (p.res) {
                v = v + bf2f(p.res[n]);  // x + linear(...), rounded once more below
            }
            if (p.y32) p.y32[n] = v;  // logits: .float() of the bf16 linear output
            else p.y[n] = f2bf(v);
        }
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

// ---- decode attention: grid (H, nsplit); 16 lanes per cache position (16 B of a 128-wide K / V row each; hd == 128) ----
// Each block takes positions [s * chunk, (s + 1) * chunk) of [0, pos]; every 16-lane group keeps a running (m, l, o[8]);
// groups and waves are merged through LDS; the block writes (m, l, o[hd]) to the workspace.
constexpr int DA_THREADS = 256;
constexpr int DA_UN = 4;  // cache positions per 16-lane group per trip: 2 x DA_UN 16-byte loads in flight per lane

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
// cache positions j0 + g, j0 + g + 16, ... < jc, DA_UN per trip: all K / V rows of a trip are requested before the first is
// used, and the trip's scores share one running-max update (a trip per position left the kernel waiting for one HBM round
// trip per position: 15 us per layer at 1050 positions for 17 MB of cache)
MERV_DEVICE void da_range(DaState& st, const float (&qf)[8], const bf16_t* Kc, const bf16_t* Vc, int j0, int jc, int g, int sub, float sc) {
    for (int j = j0 + g; j < jc; j += 16 * DA_UN) {
        u32x4 kr[DA_UN], vr[DA_UN];
#pragma unroll
        for (int u = 0; u < DA_UN; ++u) {
            const int ju = j + 16 * u < jc ? j + 16 * u : j;  // past the range: a valid row, its score is masked below
            kr[u] = *(const u32x4*)(Kc + (size_t)ju * 128 + sub * 8);
            vr[u] = *(const u32x4*)(Vc + (size_t)ju * 128 + sub * 8);
        }
        float d[DA_UN], mn = st.m;
#pragma unroll
        for (int u = 0; u < DA_UN; ++u) {
            float kf[8];
            unpack8f(kr[u], kf);
            d[u] = j + 16 * u < jc ? da_dot16(qf, kf) * sc : -INFINITY;
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
__global__ __launch_bounds__(DA_THREADS) void decode_attn_fused_kernel(DecodeAttnFusedArgs p) {
    __shared__ float sm_m[16], sm_l[16];
    __shared__ __attribute__((aligned(16))) float sm_o[16][128];
    __shared__ unsigned sm_ticket;
    const int h = blockIdx.x, s = blockIdx.y;
    const int grp_heads = p.H / p.Hkv;
    const int hkv = h / grp_heads;
    const long pos = *p.pos;
    const long npos = pos + 1;
    const int chunk = (int)((npos + p.nsplit - 1) / p.nsplit);
    const int j0 = s * chunk, j1 = (long)(j0 + chunk) < npos ? j0 + chunk : (int)npos;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int g = wave * 4 + grp;
    // rotary at *pos: lane `sub` owns dims 8 sub .. 8 sub + 7, their rotate_half partners sit in lane sub ^ 8 of the group
    float cf[8], sf[8];
    unpack8f(*(const u32x4*)(p.cos + pos * 128 + sub * 8), cf);
    unpack8f(*(const u32x4*)(p.sin + pos * 128 + sub * 8), sf);
    auto rotary = [&](const bf16_t* vec, float (&r)[8]) {
        const u32x4 own = *(const u32x4*)(vec + sub * 8);
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
    rotary(p.q + h * 128, qf);
    bf16_t* Kc = p.k_cache + (size_t)hkv * p.max_len * 128;
    bf16_t* Vc = p.v_cache + (size_t)hkv * p.max_len * 128;
    DaState st;
    st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) st.o[i] = 0.f;
    const float sc = p.scale * 1.4426950408889634f;
    da_range(st, qf, Kc, Vc, j0, j1 < pos ? j1 : (int)pos, g, sub, sc);  // cached positions of this range end before the current token
    if (pos >= j0 && pos < j1 && g == (int)((pos - j0) & 15)) {  // uniform per 16-lane group: the shuffles inside stay in the group
        float kf[8], vf[8];
        rotary(p.k + hkv * 128, kf);
        const u32x4 vraw = *(const u32x4*)(p.v + hkv * 128 + sub * 8);
        unpack8f(vraw, vf);
        da_one(st, qf, kf, vf, sc);
        if (h % grp_heads == 0) {
            u32x4 kr;
#pragma unroll
            for (int q = 0; q < 4; ++q) kr[q] = pack2bf(kf[2 * q], kf[2 * q + 1]);
            *(u32x4*)(Kc + (size_t)pos * 128 + sub * 8) = kr;
            *(u32x4*)(Vc + (size_t)pos * 128 + sub * 8) = vraw;
        }
    }
    if (sub == 0) { sm_m[g] = st.m; sm_l[g] = st.l; }
#pragma unroll
    for (int i = 0; i < 8; ++i) sm_o[g][sub * 8 + i] = st.o[i];
    __syncthreads();
    float* ws_h = p.ws + (size_t)h * p.nsplit * (128 + 2);
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
        // the partials travel between blocks (possibly between XCDs, each with its own L2) inside one launch: device-scope
        // relaxed atomics write through / read past the non-coherent cache levels, which costs nothing beside an ordinary
        // store here, whereas a device-scope FENCE writes back and invalidates the whole L2 (measured: +15 us per layer)
        float* ws = ws_h + (size_t)s * (128 + 2);
        __hip_atomic_store(ws + d, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == 0) {
            __hip_atomic_store(ws + 128, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws + 129, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // publish (every store of this block acknowledged), then take a ticket; the last arrival of this head merges.
    // This is the write-through form of the in-launch hand-off (cdna_hip_programming.md section 5, "Projection GEMM at M = 256"
    // item 2, "Equally valid": sc1 slab stores -> every wave vmcnt(0) -> __syncthreads -> relaxed agent-scope fetch_add; the
    // reducer reads the slabs with sc1 loads, EVERY load of them): relaxed agent-scope atomic stores / loads ARE the sc1 forms,
    // so no release / acquire fence pair is needed and none is paid for (+15 us per layer when tried). Two invariants carry it:
    // (1) every partial is written and read ONLY through these atomics, (2) one workspace serves one stream -- launches that
    // share it must not overlap (HipDecoder owns one per decoder and zeroes it in prefill()).
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    unsigned* counter = (unsigned*)(p.ws + (size_t)p.H * p.nsplit * (128 + 2)) + h * 32;  // one 128-byte line per head
    if (threadIdx.x == 0) sm_ticket = atomicAdd(counter, 1u);
    __syncthreads();
    if (sm_ticket != (unsigned)(p.nsplit - 1)) return;
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

}  // namespace

hipError_t launch_decode_rmsnorm(const DecodeRmsArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    if (a.D % 8 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int ROWS, int UN>
static hipError_t launch_gemv_cfg(const DecodeGemvArgs& a, hipStream_t s) {
    const int rows_per_block = ROWS * GEMV_WAVES;
    dim3 grid((a.N + a.Nb + a.Nc + rows_per_block - 1) / rows_per_block);
    const dim3 blk(GEMV_WAVES * 64);
    if (a.W2) {
        if (a.norm_w) hipLaunchKernelGGL((gemv_kernel<2, (UN > 4 ? 4 : UN), true, ROWS>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((gemv_kernel<2, (UN > 4 ? 4 : UN), false, ROWS>), grid, blk, 0, s, a);
    } else if (a.norm_w) {
        hipLaunchKernelGGL((gemv_kernel<1, UN, true, ROWS>), grid, blk, 0, s, a);
    } else {
        hipLaunchKernelGGL((gemv_kernel<1, UN, false, ROWS>), grid, blk, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_decode_gemv(const DecodeGemvArgs& a, hipStream_t s) {
    if (a.N <= 0 || a.K <= 0 || a.K % 8 != 0) return hipErrorInvalidValue;
    if (a.Nb > 0 || a.Nc > 0) {
        if (a.W2 || a.res || a.y32 || a.N % 2 || a.Nb % 2 || a.Nc % 2 || (a.Nb > 0 && (!a.Wb || !a.yb)) ||
            (a.Nc > 0 && (!a.Wc || !a.yc || a.Nb <= 0)))
            return hipErrorInvalidValue;
    }
    // UN (chunks per lane per trip) at two rows per wave, step time with Llama-2-7B geometry on MI355X: 4: 3.24 ms, 2: 3.26 ms,
    // 8: 4.60 ms (284 registers with the norm arrays live); the norm is a template flag: 3.21 ms.
    // Rows per wave / unroll for the plain projections (o_proj, down_proj: one matrix, no fused norm, 4096 rows): one row per wave
    // with the whole 16-byte-chunk batch of a trip doubled (1 x 8) keeps the same 8 loads per lane in flight on twice the waves:
    // o_proj 7.9 -> 7.8 us, down_proj 20.0 -> 18.6 (tools/probes/decode_kernels.py, MERV_GEMV_CFG sweep). With the RMSNorm fused
    // in (q / k / v) every wave reduces mean(x^2) itself, so halving the rows per wave doubles that work: 22.1 -> 29.8 us; those
    // launches and the gate / up pair stay at two rows per wave.
    static const char* cfg = getenv("MERV_GEMV_CFG");  // tuning hook: "<rows><un>", e.g. "14", "18", "24"
    const long rows_total = (long)a.N + a.Nb + a.Nc;
    int rows = (!a.W2 && !a.norm_w && rows_total <= 16384) ? 1 : 2, un = rows == 1 ? 8 : 4;
    if (cfg && cfg[0] && cfg[1]) { rows = cfg[0] - '0'; un = cfg[1] - '0'; }
    if (rows == 1 && un == 8) return launch_gemv_cfg<1, 8>(a, s);
    if (rows == 1) return launch_gemv_cfg<1, 4>(a, s);
    return launch_gemv_cfg<2, 4>(a, s);
}

hipError_t launch_decode_rope_cache(const DecodeRopeArgs& a, hipStream_t s) {
    if (a.hd % 2 != 0 || a.H <= 0 || a.Hkv <= 0) return hipErrorInvalidValue;
    const int total = (a.H + 2 * a.Hkv) * a.hd;
    hipLaunchKernelGGL(rope_cache_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
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

}  // namespace merv
