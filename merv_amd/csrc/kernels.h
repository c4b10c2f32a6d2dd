// Internal kernel launch interface (not part of the C-ABI; see include/merv_hip.h for that).
#pragma once
#include "common.h"

namespace merv {

struct GemmArgs {
    const bf16_t* A;      // [M, K] bf16, leading dim lda
    const bf16_t* W;      // [N, K] bf16 (nn.Linear layout), leading dim ldw
    bf16_t* C;            // [*, N] bf16, leading dim ldc
    const float* bias;    // [N] fp32 or nullptr
    const float* lscale;  // [N] fp32 LayerScale gamma or nullptr (applied after activation, before residual)
    const bf16_t* res;    // residual / position embedding (bf16) or nullptr, leading dim ldres
    int M, N, K;
    int lda, ldw, ldc, ldres;
    int res_row_mod;  // >0: residual row = m % res_row_mod (position-embedding add in patch embedding)
    int out_group;    // >0: output row = (m / out_group) * out_stride + out_off + (m % out_group)
    int out_stride;
    int out_off;
    int act;          // ACT_*
    int group_m;      // 0 = default; m-tiles per column sweep of the tile order (L2 locality knob)
    // MXFP8 launches only (launch_gemm_mx): A / W then point to OCP e4m3 bytes (lda / ldw in elements = bytes)
    const void* mx_scale_a;  // E8M0 block scales of A, layout of mx_quantize (below)
    const void* mx_scale_w;
    int mx_groups_a, mx_groups_w;  // 64-row groups in each scale array (ceil(rows / 64))
    // LayerNorm folded into the GEMM (A = the raw residual stream, W = bf16(W * gamma)): the epilogue's value becomes
    // acc * row_stats[m].x + row_stats[m].y * ln_colsum[n] + bias[n], with row_stats = {rstd, -mean * rstd} and
    // ln_colsum[n] = sum_k W'[n][k]; bias then holds W.beta + the layer's bias (all prepared once by launch_ln_fold)
    const float* row_stats;   // [M][2] or nullptr
    const float* ln_colsum;   // [N]
    // any launch whose output C is the residual stream a folded LayerNorm will read next: when stats_out is set the epilogue
    // also writes, per output row and per 64-column wave tile, {sum, M2 = sum of squared deviations from the tile's own mean}
    // of the bf16-ROUNDED values it stores: stats_out[((n / 64) * stats_ld + m) * 2 .. +1] -- column tile major, so a wave's rows
    // are consecutive and go out as one coalesced store per epilogue part (round 4; [m][n / 64] made every row an 8-byte
    // partial-line write and doubled the epilogue's store instructions). launch_stats_finalize combines the N / 64 partials of
    // a row (Chan's parallel variance: exact, no E[x^2] - mean^2 cancellation) into row_stats for the consuming GEMM, so no
    // kernel re-reads the residual stream for LayerNorm.
    float* stats_out;
    int stats_ld;             // rows of the partials array (0: M of this launch)
    // a second, row-indexed additive term AFTER the residual add: C[m] = bf16(bf16(lin + res) + row_add[(m / row_add_div) % row_add_mod])
    // (fp32 [row_add_mod, N]) -- LanguageBind's next block starts with x += temporal_embedding[frame]; its fc2 does that add, so
    // the temporal LayerNorm can be folded like the others (statistics from stats_out, which sees the final values). Launches
    // without an activation only; row_add_div >= 256 (a tile part then meets at most one boundary).
    const float* row_add;
    int row_add_div, row_add_mod;
    int row_add_row0;         // row of the caller's problem that this launch's row 0 is (launch_gemm's second launch starts at rows1)
    // any launch: when mx_out_q is set the epilogue writes its (bf16-rounded) result as MXFP8 -- e4m3 [M, N] + block
    // scales in the layout of mx_quantize -- instead of bf16 C (the next GEMM's quantised input, e.g. fc1 -> fc2)
    uint8_t* mx_out_q;
    uint8_t* mx_out_scales;
    int mx_out_groups;
    int mx_out_keep_c;        // with mx_out_q: 1 = bf16 C is written AS WELL (the residual stream and its MXFP8 copy for the next, LayerNorm-folded MX GEMM)
    int subround_min_tiles;   // launch_gemm: a launch of less than one round of 256 x 256 tiles takes the eight-phase kernel from this many tiles on
                              // (0 = the default, SUBROUND_MIN_TILES); set per encoder by the orchestration (merv_encoder_set_latency_critical)
    int no_static_form;       // test hook (merv_debug_gemm_mxfp8_forms): 1 = launch_gemm_mx keeps the run-time epilogue form whatever the shape
    int mx_group0_a;          // MXFP8 launches: 64-row group of mx_scale_a this launch's row 0 belongs to (launch_gemm_mx's second launch starts at rows1)
};
void set_rest_fork(hipStream_t main, hipStream_t aux);  // hooks build only (gemm.hip)
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_mx(const GemmArgs& a, hipStream_t s);

// bf16 [rows, K] -> MXFP8: q [rows, K] OCP e4m3 bytes + one E8M0 scale per 32 consecutive k (shared exponent
// floor(log2(amax)) - 8, +1 when the scaled maximum would exceed 448; elements rounded to nearest even). Scale layout, chosen so that the
// GEMM's lane (row % 16, kblock % 4) reads the scales of four row fragments as one dword:
//   scales[kblock / 4][row / 64][(kblock % 4) * 16 + row % 16][(row % 64) / 16]      (bytes; rows padded to 64)
struct MxQuantArgs {
    const bf16_t* x;  // [rows, K], leading dim ld
    uint8_t* q;       // [rows, K] contiguous
    uint8_t* scales;  // [K / 128][ceil(rows / 64)][64][4]
    int rows, K, ld;
};
hipError_t launch_mx_quantize(const MxQuantArgs& a, hipStream_t s);
size_t mx_scale_bytes(int rows, int K);
// colsum[n] = sum_k of the DE-QUANTISED row n of an MXFP8 matrix (q [N, K] + scales in mx_quantize's layout): the LayerNorm fold's
// column-sum term for a quantised folded weight (the correction -mean * rstd * colsum must cancel what the scaled MFMA summed)
hipError_t launch_mx_colsum(const uint8_t* q, const uint8_t* scales, float* colsum, int N, int K, hipStream_t s);
void set_gemm_variant(int v);  // low byte: 0 auto, 1: 128x128 2-deep ring, 3: 256x128, 4: 256x128 staggered, 6: 128x128 4-deep ring,
                               // 7: 256x256 eight-phase (2 and 5 were the retired two-stage 256x256 forms);
                               // second byte: tile-order group size override (tuning / tests)

// Row LayerNorm (fp32 statistics), optional fused "x += add[(row / add_div) % add_mod]" written back in place
// (LanguageBind temporal embedding, modeling_video.py:138-141).
struct LayerNormArgs {
    bf16_t* x;            // [M, D] bf16 (updated in place only when add != nullptr)
    bf16_t* y;            // [M, D] bf16 normalised output (may alias x when add == nullptr)
    const float* gamma;   // [D]
    const float* beta;    // [D]
    const float* add;     // [add_mod, D] fp32 or nullptr
    int M, D;
    int add_div, add_mod;
    float eps;
    // MXFP8 mode: when mx_q is set the normalised row is written as e4m3 + block scales (the layout mx_quantize
    // produces) instead of bf16 y -- the consumer is an MXFP8 GEMM
    uint8_t* mx_q;        // [M, D]
    uint8_t* mx_scales;
    int mx_groups;        // ceil(M / 64)
};
hipError_t launch_layernorm(const LayerNormArgs& a, hipStream_t s);

// Per-row LayerNorm statistics only (for the folded form): stats[m] = {rstd, -mean * rstd}.
struct RowStatsArgs {
    const bf16_t* x;  // [M, D]
    float* stats;     // [M][2]
    int M, D;
    float eps;
};
hipError_t launch_row_stats(const RowStatsArgs& a, hipStream_t s);

// Combine the per-64-column partials a producer GEMM's epilogue wrote (GemmArgs::stats_out) into stats[m] = {rstd, -mean * rstd}.
struct StatsFinalizeArgs {
    const float* parts;  // [nparts][M][2] = {sum, M2} over 64 columns each
    float* stats;        // [M][2]
    int M, nparts;       // D = 64 * nparts (nparts <= 32)
    float eps;
};
hipError_t launch_stats_finalize(const StatsFinalizeArgs& a, hipStream_t s);

// One-time fold of a LayerNorm into the Linear that consumes it: wf[n][k] = bf16(w[n][k] * gamma[k]),
// colsum[n] = sum_k wf[n][k] (of the ROUNDED values, so the algebra stays exact), dbias[n] = sum_k w[n][k] * beta[k] + bias[n].
struct LnFoldArgs {
    const bf16_t* w;      // [N, K]
    const float* gamma;   // [K]
    const float* beta;    // [K]
    const float* bias;    // [N] or nullptr
    bf16_t* wf;           // [N, K]
    float* colsum;        // [N]
    float* dbias;         // [N]
    int N, K;
    // rows n < scale_rows are additionally multiplied by row_scale BEFORE the one rounding to bf16 (the q rows of a qkv weight
    // take the attention's scale * log2(e) this way: AttnArgs::q_prescaled); dbias likewise, colsum of the rounded values
    int scale_rows;
    float row_scale;
};
hipError_t launch_ln_fold(const LnFoldArgs& a, hipStream_t s);

// Multi-head self attention over packed QKV rows: qkv[row][{q,k,v} * D + head * 64 + d], head_dim = 64.
// Sequence s covers rows [s*L, (s+1)*L). out[row][head*64 + d].
struct AttnArgs {
    const bf16_t* qkv;  // [nseq*L, 3*D]
    bf16_t* out;        // [nseq*L, D]
    int nseq, L, heads, D;
    float scale;        // 1/sqrt(head_dim)
    // online softmax: the running reference m of a query moves only when a key tile's maximum exceeds it by more than this many
    // binary orders of magnitude (attention.hip, "deferred max"); set by the launcher (default 8, MERV_ATTN_RESCALE_THR overrides)
    float rescale_thr;
    // MXFP8 mode: when mx_q is set the output goes out as e4m3 [rows, D] + block scales (layout of mx_quantize)
    // instead of bf16 `out` -- it is the quantised input of the out-projection GEMM
    uint8_t* mx_q;
    uint8_t* mx_scales;
    int mx_groups;
    // the q third of qkv already carries scale * log2(e) (folded into the q rows of the qkv weight and bias by launch_ln_fold):
    // the kernel then uses q as it is instead of scaling and re-rounding it
    int q_prescaled;
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
// Causal attention of the LLM prompt prefill: head_dim 128, one sequence of S positions, GQA (kv head = head / (H / Hkv)).
struct PrefillAttnArgs {
    const bf16_t* q;     // [S, ldq], head h at columns [128 h, 128 h + 128) (rotary already applied)
    const bf16_t* k;     // kv head g, position s at k + g * kv_head_stride + s * ldk (the KV cache: ldk = 128, stride = max_len * 128)
    const bf16_t* v;
    bf16_t* out;         // [S, ldo], same column layout as q
    int S, H, Hkv;
    int ldq, ldk, ldo;   // elements
    long kv_head_stride; // elements
    float scale;         // 1 / sqrt(128)
    float rescale_thr;   // set by the launcher (see AttnArgs)
};
hipError_t launch_prefill_attention(const PrefillAttnArgs& a, hipStream_t s);
void set_attn_rescale_thr(float thr);  // test / probe hook (default 8, or MERV_ATTN_RESCALE_THR read once per process)

// LanguageBind temporal attention: for every (clip, token, head) an 8x8 attention over the clip's t frames.
// Rows are frame-major: row = frame * ntok + token, frame = clip * t + i (modeling_video.py:133-155).
struct TemporalAttnArgs {
    const bf16_t* qkv;  // [nclips*t*ntok, 3*D]
    bf16_t* out;        // [nclips*t*ntok, D]
    int nclips, t, ntok, heads, D;
    float scale;
    uint8_t* mx_q;      // MXFP8 mode, as in AttnArgs
    uint8_t* mx_scales;
    int mx_groups;
};
hipError_t launch_temporal_attention(const TemporalAttnArgs& a, hipStream_t s);

// im2col for Conv2d(k=s=p) / Conv3d(k=s=(tt,p,p)) patch embedding. Output A[m][k] bf16 with
// m = ((b*Fo + fo)*Hp + py)*Wp + px and k = ((c*tt + dt)*p + dy)*p + dx, zero padded to kpad.
struct Im2colArgs {
    const void* pix;   // fp32 or bf16 pixels
    int pix_is_bf16;
    bf16_t* out;       // [B*Fo*Hp*Wp, kpad]
    int B, frames, img, patch, tt, kpad;
    long long sB, sF, sC;  // element strides of batch / frame / channel in `pix` (row stride = img, col stride = 1)
};
hipError_t launch_im2col(const Im2colArgs& a, hipStream_t s);

// Broadcast prefix token rows (cls / register tokens, position already folded in) into every sequence.
struct PrefixArgs {
    const bf16_t* prefix;  // [npre, D]
    bf16_t* x;             // [nseq*ntok, D]
    int nseq, ntok, npre, D;
};
hipError_t launch_prefix(const PrefixArgs& a, hipStream_t s);

// Strip prefix tokens: out[b][f*S + s][:] = x[row(b,f) + s][:], row(b,f) = b*bstride + f*fstride + prefix.
// (VideoBackbone.forward output contract, [B, num_patches, embed_dim]).
struct GatherTokensArgs {
    const bf16_t* x;
    bf16_t* out;
    int B, T, S, D;
    int bstride, fstride, prefix;
};
hipError_t launch_gather_tokens(const GatherTokensArgs& a, hipStream_t s);

// Attention pooling with ONE fixed query per head (timm AttentionPoolLatent / SigLIP "MAP" head): per sequence n and head h,
// out[n][h*64 + d] = sum_s softmax_s(scale * q[h*64:] . k[n][s][h*64:]) v[n][s][h*64 + d]; kv rows hold [k (D) | v (D)] bf16.
struct MapPoolArgs {
    const bf16_t* kv;  // [nseq * ntok, 2 * D]
    const float* q;    // [D] the projected latent query
    bf16_t* out;       // [nseq, D]
    int nseq, ntok, heads;
    float scale;       // 1 / sqrt(64)
};
hipError_t launch_map_pool(const MapPoolArgs& a, hipStream_t s);

// out[g][:] = bf16(mean_r x[g][r][:]), fp32 accumulation (torch's mean on a bf16 tensor). x [groups, rows, D] with a row stride
// of D and a group stride of group_stride rows.
struct MeanRowsArgs {
    const bf16_t* x;
    bf16_t* out;
    int groups, rows, D;
    long long group_stride;  // in rows
};
hipError_t launch_mean_rows(const MeanRowsArgs& a, hipStream_t s);

// AdaptiveAvgPool3d((T, Ho, Ho)) over tokens laid out [B, T, S*S, C] (nn_utils.py:320-328); T is kept
// (output_frames == temporal_resolution for 3davg), spatial S -> Ho with torch's window rule
// [floor(i*S/Ho), ceil((i+1)*S/Ho)). Output [B*T*Ho*Ho, C] bf16.
struct PoolArgs {
    const bf16_t* x;   // [B, T*S*S, C] contiguous tokens (prefix already stripped)
    bf16_t* out;
    int B, T, S, Ho, C;
};
hipError_t launch_pool(const PoolArgs& a, hipStream_t s);

// Cross-encoder fusion (nn_utils.py:487-521, averagetoken=True): score[b][e] = mean_t(V_e[b][t]) . u,
// w = softmax_e(score), out = sum_e w_e V_e. u = Wk^T (Wq Q + bq) / sqrt(embed_dim) is folded on the host.
struct FusionArgs {
    const bf16_t* v[8];   // E tensors [B, T, C]
    int E, B, T, C;
    const float* u;       // [C]
    float* partial;       // workspace [B*E*nchunk]
    float* weights;       // [B, E] fp32 out
    bf16_t* out;          // [B, T, C]
};
hipError_t launch_fusion(const FusionArgs& a, hipStream_t s);
int fusion_partial_floats(int B, int E, int T);

// ---- backward of the trainable tail (backward.hip) ----
struct FusionBwdArgs {
    const bf16_t* v[8];      // E tensors [B, T, C] (the forward inputs)
    const bf16_t* grad_out;  // [B, T, C]
    int E, B, T, C;
    float* partial_vs;       // workspace, set by the launcher
    float* partial_dw;       // workspace, set by the launcher
    float* dw;               // [B, E] out: sum_{t,c} grad_out * V_e
    float* vbar;             // [B, E, C] out: mean_t V_e
};
size_t fusion_bwd_workspace_floats(int B, int E, int T, int C);
hipError_t launch_fusion_bwd_reduce(FusionBwdArgs a, float* ws, hipStream_t s);

struct FusionBwdMixArgs {
    const bf16_t* grad_out;  // [B, T, C]
    const float* w;          // [B, E] forward softmax weights
    const float* ds;         // [B, E] gradient w.r.t. the scores
    const float* u;          // [C]
    bf16_t* dv[8];           // E outputs [B, T, C]
    int E, B, T, C;
};
hipError_t launch_fusion_bwd_mix(const FusionBwdMixArgs& a, hipStream_t s);

struct TransposeArgs {
    const bf16_t* in;  // [R, C], leading dim ldi
    bf16_t* out;       // [C, Rpad], leading dim ldo; columns R..Rpad-1 are zero-filled
    int R, C, ldi, ldo, Rpad;
};
hipError_t launch_transpose(const TransposeArgs& a, hipStream_t s);

struct ColsumArgs {
    const bf16_t* x;  // [M, N], leading dim ld
    float* partial;   // workspace [colsum_workspace_floats(N)]
    float* out;       // [N]
    int M, N, ld;
};
size_t colsum_workspace_floats(int N);
hipError_t launch_colsum(const ColsumArgs& a, hipStream_t s);

// Splice fused visual tokens after the BOS embedding (merv.py:633-640):
// out[b] = cat(emb[b, :bos], vis[b], emb[b, bos:]).
struct SpliceArgs {
    const bf16_t* emb;  // [B, S, C]
    const bf16_t* vis;  // [B, T, C]
    bf16_t* out;        // [B, S+T, C]
    int B, S, T, C, bos;
};
hipError_t launch_splice(const SpliceArgs& a, hipStream_t s);

// Frame preprocessing (preproc.hip). kind: 0 = bilinear, 1 = bicubic (Pillow filters).
size_t pil_workspace_bytes(int T, int H, int W, int out);
hipError_t launch_pil_resize_normalize(const uint8_t* frames, int T, int H, int W, int out, int kind, const float* mean,
                                       const float* sd, void* out_pix, int out_bf16, uint8_t* resized_u8_opt, char* ws, hipStream_t s);
hipError_t launch_languagebind_transform(const uint8_t* frames, int T, int H, int W, int S, int flip, const float* mean, const float* sd,
                                         void* out_pix, int out_bf16, hipStream_t s);


// ---- batch-1 token decode of the LLM hand-off (decode.hip; row f-3) ----
struct DecodeRmsArgs {   // LlamaRMSNorm: y = w * bf16(x * rsqrt(mean(x^2) + eps))
    const bf16_t* x;     // [rows, D]
    const bf16_t* w;     // [D]
    bf16_t* y;           // [rows, D]
    int rows, D;
    float eps;
};
hipError_t launch_decode_rmsnorm(const DecodeRmsArgs& a, hipStream_t s);
// x <- bf16(x + delta) in place (a.x is written), y = RMSNorm of the updated rows; D <= 8192
hipError_t launch_add_rmsnorm(const DecodeRmsArgs& a, const bf16_t* delta, hipStream_t s);

struct DecodeGemvArgs {  // y = bf16(W x) (+ res), or with W2: y = silu(bf16(W x)) * bf16(W2 x); x [K], y [N]
    // up to three matrices that share x in ONE launch (q / k / v projections): rows [0, N) belong to W / y, rows
    // [N, N + Nb) to Wb / yb, rows [N + Nb, N + Nb + Nc) to Wc / yc (Nb = Nc = 0: a single matrix)
    const bf16_t* Wb;
    const bf16_t* Wc;
    bf16_t* yb;
    bf16_t* yc;
    int Nb, Nc;
    const bf16_t* W;     // [N, K] (nn.Linear layout)
    const bf16_t* W2;    // [N, K] or nullptr
    const bf16_t* x;     // [K]
    const bf16_t* res;   // [N] or nullptr (may alias y)
    bf16_t* y;           // [N] bf16 output ...
    float* y32;          // ... or, when set, fp32 output (logits)
    int N, K;
    // RMSNorm of the input fused in (input_layernorm -> q/k/v, post_attention_layernorm -> gate/up, norm -> lm_head): x is the
    // raw residual stream and every wave normalises the chunks it multiplies, w * bf16(x * rsqrt(mean(x^2) + eps)) -- the
    // same two roundings as the stand-alone kernel, so the result is bit-identical to rmsnorm + gemv
    const bf16_t* norm_w;  // [K] or nullptr
    float norm_eps;
    // nn.Linear biases of the (up to three) matrices, bf16 [N] / [Nb] / [Nc] or nullptr (Qwen2's q / k / v projections):
    // added to the fp32 accumulator before the one rounding of the output
    const bf16_t* bias;
    const bf16_t* bias_b;
    const bf16_t* bias_c;
};
hipError_t launch_decode_gemv(const DecodeGemvArgs& a, hipStream_t s);

struct DecodeRopeArgs {  // rotary embedding of q / k at *pos, cache[pos] <- k, v
    const bf16_t* q;     // [H * hd]
    const bf16_t* k;     // [Hkv * hd]
    const bf16_t* v;     // [Hkv * hd]
    bf16_t* q_out;       // [H * hd] rotated query
    bf16_t* k_cache;     // [Hkv, max_len, hd]
    bf16_t* v_cache;
    const bf16_t* cos;   // [max_len, hd]
    const bf16_t* sin;
    const long* pos;     // device int64: position of the token being decoded
    int H, Hkv, hd, max_len;
};
hipError_t launch_decode_rope_cache(const DecodeRopeArgs& a, hipStream_t s);

// Prompt prefill (batch 1, S positions at once) around the library GEMMs: the elementwise parts of a decoder layer.
struct PrefillRopeArgs {  // apply_rotary_pos_emb on S positions + cache fill: q rotated in place, cache[:, pos0 + s] <- rot(k), v
    bf16_t* q;           // [S, ldq], H * hd columns used
    const bf16_t* k;     // [S, ldk], Hkv * hd columns used
    const bf16_t* v;     // [S, ldk]
    bf16_t* k_cache;     // [Hkv, max_len, hd]
    bf16_t* v_cache;
    const bf16_t* cos;   // [max_len, hd]
    const bf16_t* sin;
    int S, pos0, H, Hkv, hd, max_len;
    int ldq, ldk;        // row strides in elements (q, k and v may be column ranges of one [S, (H + 2 Hkv) * hd] projection output)
};
hipError_t launch_prefill_rope_cache(const PrefillRopeArgs& a, hipStream_t s);
struct SiluMulArgs {     // LlamaMLP's act_fn(gate) * up on materialised bf16 tensors: out = bf16(bf16(silu(gate)) * up)
    const bf16_t* gate;
    const bf16_t* up;
    bf16_t* out;         // may alias gate or up
    long n;              // elements, a multiple of 8
};
hipError_t launch_silu_mul(const SiluMulArgs& a, hipStream_t s);

struct DecodeAttnArgs {  // softmax(q K^T * scale) V over cache positions 0 .. *pos, one query per head; hd == 128
    const bf16_t* q;     // [H * hd]
    const bf16_t* k_cache;
    const bf16_t* v_cache;
    bf16_t* out;         // [H * hd]
    float* ws;           // [H][nsplit][hd + 2] partial (o, m, l)
    const long* pos;
    int H, Hkv, hd, max_len, nsplit;
    float scale;
};
hipError_t launch_decode_attention(const DecodeAttnArgs& a, hipStream_t s);

struct DecodeAttnFusedArgs {  // rotary(q, k at *pos) + cache[*pos] <- k, v + split attention + merge, ONE launch; hd == 128
    const bf16_t* q;     // [H * hd] raw q_proj output
    const bf16_t* k;     // [Hkv * hd] raw k_proj output
    const bf16_t* v;     // [Hkv * hd]
    const bf16_t* cos;   // [max_len, hd]
    const bf16_t* sin;
    bf16_t* k_cache;     // [Hkv, max_len, hd]
    bf16_t* v_cache;
    bf16_t* out;         // [H * hd]
    float* ws;           // [H][nsplit][hd + 2] partial (o, m, l), then H arrival counters (uint32, zero between launches)
    const long* pos;
    int H, Hkv, hd, max_len, nsplit;
    float scale;
    // split launch only (round 6): the NEXT launch's weight matrix (the o-projection's W_o), touched by extra blocks while this launch --
    // a chain of dependent round trips over 17 MB of cache that leaves HBM idle -- runs: prefetch block b reads bytes [b, b + 1) * prefetch_block_bytes,
    // the rows o-projection block b will read, and lands on the same XCD as that block (workgroups go to XCDs round-robin), i.e. in ITS L2
    const void* prefetch;
    int prefetch_block_bytes, prefetch_blocks;
};
hipError_t launch_decode_attention_fused(const DecodeAttnFusedArgs& a, hipStream_t s);
// the same launch ending at the partials (a.out unused; no counters): ws [H][nsplit][132] = (o[128], m, l, 2 pad) is complete when
// the launch is; decode_attention_split_workspace_floats(H, nsplit) floats, 16-byte aligned
hipError_t launch_decode_attention_split(const DecodeAttnFusedArgs& a, hipStream_t s);
size_t decode_attention_split_workspace_floats(int H, int nsplit);
struct DecodeOprojMergeArgs {  // y = res + W . merge(ws): the o-projection behind launch_decode_attention_split
    const bf16_t* W;     // [N, H * 128]
    const bf16_t* res;   // [N] or nullptr
    bf16_t* y;           // [N]
    const float* ws;     // [H][nsplit][132], the split launch's layout
    bf16_t* attn_out;    // [H * 128] merged attention output (optional)
    int N, H, nsplit;
};
hipError_t launch_decode_oproj_merge(const DecodeOprojMergeArgs& a, hipStream_t s);
// greedy decoding inside a captured step: tok[0] <- argmax(logits[V]) (first maximum), out_tokens[*pos - pos0] <- it, *pos += 1
hipError_t launch_decode_greedy_advance(const float* logits, int V, long* tok, long* pos, long* out_tokens, long pos0, hipStream_t s);
hipError_t launch_decode_sample_advance(const float* logits, int V, const void* params, long* tok, long* pos, long* out_tokens, long pos0, hipStream_t s);

}  // namespace merv
