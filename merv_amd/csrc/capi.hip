// C-ABI of libmerv_hip.so (see include/merv_hip.h): argument checking, encoder orchestration (a stream-ordered
// sequence of kernel launches per encoder, no allocation, no synchronisation) and thin kernel wrappers.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/merv_hip.h"
#include "common.h"
#include "kernels.h"

using namespace merv;

static thread_local std::string g_err;
extern "C" void merv_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* merv_last_error(void) { return g_err.c_str(); }
extern "C" int merv_abi_version(void) { return MERV_ABI_VERSION; }

#define MERV_CHECK(cond, msg)        \
    do {                             \
        if (!(cond)) {               \
            merv_set_error(msg);     \
            return 1;                \
        }                            \
    } while (0)

#define MERV_HIP(expr)                                                                  \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            char buf__[256];                                                            \
            snprintf(buf__, sizeof buf__, "%s failed: %s", #expr, hipGetErrorString(e__)); \
            merv_set_error(buf__);                                                      \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

// Every entry point launches on the caller's stream. HIP launches go to the CURRENT device, which need not be the
// stream's: make the stream's device current for the duration of the call (and restore it), so that a caller that
// holds tensors on cuda:1 while cuda:0 is current gets correct launches instead of a foreign-device stream error.
struct StreamDeviceGuard {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit StreamDeviceGuard(void* stream_) {
        err = hipGetDevice(&prev);
        if (err != hipSuccess) return;
        dev = prev;
        hipStream_t s = (hipStream_t)stream_;
        if (s != nullptr) {  // the null stream always belongs to the current device
            hipDevice_t d;
            if (hipStreamGetDevice(s, &d) == hipSuccess) dev = (int)d;
            else (void)hipGetLastError();  // e.g. a capturing stream on an older runtime: keep the current device
        }
        if (dev != prev) err = hipSetDevice(dev);
    }
    ~StreamDeviceGuard() {
        if (dev != prev && prev >= 0) (void)hipSetDevice(prev);
    }
};
#define MERV_STREAM_DEVICE(stream_)                                                   \
    StreamDeviceGuard sdg__(stream_);                                                 \
    if (sdg__.err != hipSuccess) {                                                    \
        merv_set_error("could not make the stream's device current");                \
        return 2;                                                                     \
    }

struct MxLayer {  // MXFP8 copies of one block's four GEMM weights (elements + block scales)
    const uint8_t *qkv_q, *qkv_s, *proj_q, *proj_s, *fc1_q, *fc1_s, *fc2_q, *fc2_s;
    const uint8_t *tqkv_q, *tqkv_s, *tproj_q, *tproj_s;  // LanguageBind temporal sub-block (null otherwise)
    // with the LayerNorm fold (mx_folded): qkv_q / fc1_q / tqkv_q above are quantised from the FOLDED weights bf16(W * gamma) and these are the
    // column sums of their de-quantised rows (the fold's -mean * rstd * colsum term has to cancel what the scaled MFMA summed); block 0 of
    // LanguageBind keeps a LayerNorm kernel in front of its temporal qkv (its x comes from the embedding), so the unfolded weight is kept too
    const float *qkv_cs, *fc1_cs, *tqkv_cs;
    const uint8_t *tqkv_raw_q, *tqkv_raw_s;
};

struct FoldLayer {  // LN1 folded into qkv, LN2 into fc1 (LanguageBind: the temporal LayerNorm into the temporal qkv): bf16(W * gamma), sum_k of it, W.beta + bias
    const bf16_t *qkv_w, *fc1_w, *tqkv_w;
    const float *qkv_cs, *qkv_db, *fc1_cs, *fc1_db, *tqkv_cs, *tqkv_db;
};

struct merv_encoder {
    merv_encoder_desc d;
    merv_encoder_weights w;
    std::vector<merv_layer_weights> layers;
    bool mx = false;            // MXFP8 mode enabled (merv_encoder_enable_mxfp8)
    bool mx_folded = false;     // ... on the LayerNorm-folded weights (the fold was enabled first): no LayerNorm / quantisation pass in front of qkv / fc1
    int mx_mask = 15;           // which block GEMMs use it: bit 0 qkv (+ temporal qkv), 1 out-projection (+ temporal), 2 fc1, 3 fc2
    std::vector<MxLayer> mxl;
    bool latency_critical = true;  // this encoder's chain ends its step (merv_encoder_set_latency_critical): sub-round GEMM launches take the fast wide form
    bool fold = false;          // LayerNorm folded into the qkv / fc1 GEMMs (merv_encoder_enable_ln_fold)
    std::vector<FoldLayer> fl;
    // derived geometry
    int hp;        // patches per side
    int P;         // patch tokens per sequence
    int ntok;      // tokens per sequence (prefix + P)
    int seq_per_video;
    int T_out;     // temporal resolution of the output (frames, or frames / tubelet for ViViT)
    int S_out;     // spatial tokens per output frame
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static constexpr float ATTN_SCALE = 0.125f;  // 1 / sqrt(head_dim = 64)

extern "C" int merv_encoder_create(const merv_encoder_desc* desc, const merv_encoder_weights* w, merv_encoder** out) {
    MERV_CHECK(desc && w && out, "merv_encoder_create: null argument");
    const merv_encoder_desc& d = *desc;
    MERV_CHECK(d.dim > 0 && d.dim % 128 == 0 && d.heads * 64 == d.dim, "encoder: dim must be heads*64 and a multiple of 128");
    MERV_CHECK(d.mlp_dim > 0 && d.mlp_dim % 128 == 0, "encoder: mlp_dim must be a multiple of 128");
    MERV_CHECK(d.layers >= 0, "encoder: layers < 0");
    MERV_CHECK(d.patch > 0 && d.img % d.patch == 0, "encoder: img must be a multiple of patch");
    MERV_CHECK(d.tubelet >= 1 && d.frames >= 1 && d.frames % d.tubelet == 0, "encoder: frames must be a multiple of tubelet");
    MERV_CHECK(d.k_pad % 64 == 0 && d.k_pad >= 3 * d.tubelet * d.patch * d.patch, "encoder: k_pad must be a multiple of 64 covering the patch");
    MERV_CHECK(d.prefix_tokens >= 0, "encoder: prefix_tokens < 0");
    MERV_CHECK(!d.joint_space_time || d.temporal_frames == 0, "encoder: temporal attention needs per-frame sequences");
    MERV_CHECK(d.temporal_frames == 0 || (d.temporal_frames == 8 && d.frames % 8 == 0),
               "encoder: temporal attention supports t == 8 with frames % 8 == 0");
    MERV_CHECK(d.act >= MERV_ACT_NONE && d.act <= MERV_ACT_QUICK_GELU, "encoder: bad activation");
    MERV_CHECK(w->patch_w && w->pos, "encoder: patch_w / pos missing");
    MERV_CHECK(d.prefix_tokens == 0 || w->prefix, "encoder: prefix rows missing");
    MERV_CHECK(!d.pre_ln || (w->pre_ln_w && w->pre_ln_b), "encoder: pre_ln weights missing");
    MERV_CHECK(!d.final_ln || (w->final_ln_w && w->final_ln_b), "encoder: final_ln weights missing");
    MERV_CHECK(d.layers == 0 || w->layers, "encoder: layer table missing");
    for (int i = 0; i < d.layers; ++i) {
        const merv_layer_weights& L = w->layers[i];
        MERV_CHECK(L.ln1_w && L.ln1_b && L.qkv_w && L.qkv_b && L.proj_w && L.proj_b && L.ln2_w && L.ln2_b && L.fc1_w &&
                       L.fc1_b && L.fc2_w && L.fc2_b,
                   "encoder: a block weight is missing");
        MERV_CHECK(!d.layerscale || (L.ls1 && L.ls2), "encoder: LayerScale weights missing");
        MERV_CHECK(d.temporal_frames == 0 ||
                       (L.t_emb && L.t_ln_w && L.t_ln_b && L.t_qkv_w && L.t_qkv_b && L.t_proj_w && L.t_proj_b),
                   "encoder: temporal block weights missing");
    }
    merv_encoder* e = new merv_encoder();
    e->d = d;
    e->w = *w;
    e->layers.assign(w->layers, w->layers + d.layers);
    e->w.layers = e->layers.data();
    e->hp = d.img / d.patch;
    const int fo = d.frames / d.tubelet;
    if (d.joint_space_time) {
        e->P = fo * e->hp * e->hp;
        e->seq_per_video = 1;
    } else {
        e->P = e->hp * e->hp;
        e->seq_per_video = fo;
    }
    e->ntok = d.prefix_tokens + e->P;
    e->T_out = fo;
    e->S_out = e->hp * e->hp;
    *out = e;
    return 0;
}

extern "C" void merv_encoder_destroy(merv_encoder* enc) { delete enc; }

extern "C" int32_t merv_encoder_num_patches(const merv_encoder* enc) { return enc ? enc->T_out * enc->S_out : 0; }

namespace {
struct Workspace {
    bf16_t *x, *y, *qkv, *h;
    uint8_t *aq, *asc;  // MXFP8 mode: quantised [M, dim] GEMM input (LayerNorm / attention output) and its block scales
    uint8_t *hq, *hsc;  // MXFP8 mode: quantised [M, mlp_dim] MLP hidden activations, written by fc1's epilogue
    uint8_t *xq, *xsc;  // MXFP8 mode with the LayerNorm fold: the residual stream's MXFP8 copy, written by the epilogue of the GEMM that wrote x
    float* stats;       // folded LayerNorm: {rstd, -mean * rstd} per row
    float* parts;       // folded LayerNorm: per-row {sum, M2} partials per 64 columns, written by the producing GEMM's epilogue
    size_t total;
};
Workspace carve(const merv_encoder* e, int nseq, char* base) {
    const size_t M = (size_t)nseq * e->ntok;
    const size_t D = e->d.dim;
    size_t hcols = e->d.mlp_dim;
    if ((size_t)e->d.k_pad > hcols) hcols = e->d.k_pad;  // im2col matrix aliases the MLP hidden buffer
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return (bf16_t*)p;
    };
    w.x = take(M * D * 2);
    w.y = take(M * D * 2);
    w.qkv = take(M * 3 * D * 2);
    w.h = take(M * hcols * 2);
    w.aq = w.asc = w.hq = w.hsc = w.xq = w.xsc = nullptr;
    if (e->mx) {
        w.aq = (uint8_t*)take(M * D);
        w.asc = (uint8_t*)take(mx_scale_bytes((int)M, (int)D));
        w.hq = (uint8_t*)take(M * e->d.mlp_dim);
        w.hsc = (uint8_t*)take(mx_scale_bytes((int)M, e->d.mlp_dim));
        if (e->mx_folded) {
            w.xq = (uint8_t*)take(M * D);
            w.xsc = (uint8_t*)take(mx_scale_bytes((int)M, (int)D));
        }
    }
    w.stats = e->fold ? (float*)take(M * 2 * sizeof(float)) : nullptr;
    w.parts = e->fold ? (float*)take(M * (D / 64) * 2 * sizeof(float)) : nullptr;
    w.total = off;
    return w;
}
}  // namespace

extern "C" int merv_encoder_set_latency_critical(merv_encoder* e, int32_t critical) {
    MERV_CHECK(e, "merv_encoder_set_latency_critical: null encoder");
    e->latency_critical = critical != 0;
    return 0;
}

extern "C" int merv_encoder_set_mxfp8_mask(merv_encoder* e, int32_t mask) {
    MERV_CHECK(e, "merv_encoder_set_mxfp8_mask: null encoder");
    MERV_CHECK(mask >= 0 && mask <= 15, "merv_encoder_set_mxfp8_mask: mask is 4 bits (qkv, out-projection, fc1, fc2)");
    e->mx_mask = mask;
    return 0;
}

// ---- LayerNorm folded into the consuming GEMM (bf16 path) ----
static size_t fold_weight_bytes(int N, int K) { return align_up((size_t)N * K * 2, 256) + 2 * align_up((size_t)N * 4, 256); }

extern "C" size_t merv_encoder_ln_fold_bytes(const merv_encoder* e) {
    if (!e) return 0;
    return (size_t)e->d.layers * ((e->d.temporal_frames > 0 ? 2 : 1) * fold_weight_bytes(3 * e->d.dim, e->d.dim) +
                                  fold_weight_bytes(e->d.mlp_dim, e->d.dim));
}

extern "C" int merv_encoder_enable_ln_fold(merv_encoder* e, void* buf, size_t bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(e && buf, "merv_encoder_enable_ln_fold: null argument");
    MERV_CHECK(bytes >= merv_encoder_ln_fold_bytes(e), "merv_encoder_enable_ln_fold: buffer too small");
    MERV_CHECK(((uintptr_t)buf & 255) == 0, "merv_encoder_enable_ln_fold: buffer must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    const int D = e->d.dim, H = e->d.mlp_dim;
    char* p = (char*)buf;
    e->fl.assign(e->d.layers, FoldLayer{});
    // scale_rows / row_scale: the q rows of the qkv weight also take the attention's scale * log2(e) (AttnArgs::q_prescaled)
    auto fold = [&](const void* w, const float* gamma, const float* beta, const float* bias, int N, int K, const bf16_t*& wf,
                    const float*& cs, const float*& db, int scale_rows, float row_scale) -> hipError_t {
        bf16_t* wd = (bf16_t*)p;
        float* csd = (float*)(p + align_up((size_t)N * K * 2, 256));
        float* dbd = (float*)((char*)csd + align_up((size_t)N * 4, 256));
        p += fold_weight_bytes(N, K);
        wf = wd; cs = csd; db = dbd;
        LnFoldArgs a{(const bf16_t*)w, gamma, beta, bias, wd, csd, dbd, N, K, scale_rows, row_scale};
        return launch_ln_fold(a, s);
    };
    for (int i = 0; i < e->d.layers; ++i) {
        const merv_layer_weights& L = e->layers[i];
        FoldLayer& f = e->fl[i];
        MERV_HIP(fold(L.qkv_w, L.ln1_w, L.ln1_b, L.qkv_b, 3 * D, D, f.qkv_w, f.qkv_cs, f.qkv_db, D, ATTN_SCALE * 1.4426950408889634f));
        MERV_HIP(fold(L.fc1_w, L.ln2_w, L.ln2_b, L.fc1_b, H, D, f.fc1_w, f.fc1_cs, f.fc1_db, 0, 1.0f));
        if (e->d.temporal_frames > 0)  // (the temporal attention kernel applies its own scale)
            MERV_HIP(fold(L.t_qkv_w, L.t_ln_w, L.t_ln_b, L.t_qkv_b, 3 * D, D, f.tqkv_w, f.tqkv_cs, f.tqkv_db, 0, 1.0f));
    }
    e->fold = true;
    return 0;
}

// ---- MXFP8 mode of the encoder blocks (BASELINE.json configs[4]) ----
static size_t mx_weight_bytes(int N, int K) { return align_up((size_t)N * K, 256) + align_up(mx_scale_bytes(N, K), 256); }

extern "C" size_t merv_encoder_mxfp8_bytes(const merv_encoder* e) {
    if (!e) return 0;
    const int D = e->d.dim, H = e->d.mlp_dim;
    size_t per_layer = mx_weight_bytes(3 * D, D) + mx_weight_bytes(D, D) + mx_weight_bytes(H, D) + mx_weight_bytes(D, H);
    if (e->d.temporal_frames > 0) per_layer += mx_weight_bytes(3 * D, D) + mx_weight_bytes(D, D);
    if (e->fold) {  // column sums of the quantised folded weights (+ LanguageBind: the unfolded temporal qkv for block 0's LayerNorm kernel)
        per_layer += align_up((size_t)3 * D * 4, 256) + align_up((size_t)H * 4, 256);
        if (e->d.temporal_frames > 0) per_layer += align_up((size_t)3 * D * 4, 256) + mx_weight_bytes(3 * D, D);
    }
    return (size_t)e->d.layers * per_layer;
}

extern "C" int merv_encoder_enable_mxfp8(merv_encoder* e, void* buf, size_t bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(e && buf, "merv_encoder_enable_mxfp8: null argument");
    const int D = e->d.dim, H = e->d.mlp_dim;
    MERV_CHECK(D % 256 == 0 && D >= 512 && H % 256 == 0 && H >= 512, "merv_encoder_enable_mxfp8: dim and mlp_dim must be multiples of 256, >= 512");
    MERV_CHECK(bytes >= merv_encoder_mxfp8_bytes(e), "merv_encoder_enable_mxfp8: buffer too small");
    MERV_CHECK(((uintptr_t)buf & 255) == 0, "merv_encoder_enable_mxfp8: buffer must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    char* p = (char*)buf;
    e->mxl.assign(e->d.layers, MxLayer{});
    auto quant = [&](const void* w, int N, int K, const uint8_t*& q, const uint8_t*& sc) -> hipError_t {
        uint8_t* qd = (uint8_t*)p;
        uint8_t* sd = (uint8_t*)(p + align_up((size_t)N * K, 256));
        p += mx_weight_bytes(N, K);
        q = qd; sc = sd;
        hipError_t err = hipMemsetAsync(sd, 0, mx_scale_bytes(N, K), s);  // padding rows of the last 64-row group
        if (err != hipSuccess) return err;
        MxQuantArgs a{(const bf16_t*)w, qd, sd, N, K, K};
        return launch_mx_quantize(a, s);
    };
    // With the LayerNorm fold enabled (the default: merv_encoder_enable_ln_fold ran first) the MXFP8 weights of qkv / fc1 / temporal qkv are
    // quantised from the FOLDED bf16(W * gamma): their GEMMs then read the MXFP8 copy of the raw residual stream that the producing GEMM's
    // epilogue wrote (GemmArgs::mx_out_keep_c) and normalise in their own epilogue, like the bf16 fold -- no LayerNorm + quantisation kernel
    // re-reads the stream (round 6; round 5's MX mode ran 163 such passes per step).
    const bool folded = e->fold;
    auto colsum = [&](const uint8_t* q, const uint8_t* sc, int N, int K, const float*& cs) -> hipError_t {
        float* d = (float*)p;
        p += align_up((size_t)N * 4, 256);
        cs = d;
        return launch_mx_colsum(q, sc, d, N, K, s);
    };
    for (int i = 0; i < e->d.layers; ++i) {
        const merv_layer_weights& L = e->layers[i];
        MxLayer& m = e->mxl[i];
        MERV_HIP(quant(folded ? (const void*)e->fl[i].qkv_w : L.qkv_w, 3 * D, D, m.qkv_q, m.qkv_s));
        MERV_HIP(quant(L.proj_w, D, D, m.proj_q, m.proj_s));
        MERV_HIP(quant(folded ? (const void*)e->fl[i].fc1_w : L.fc1_w, H, D, m.fc1_q, m.fc1_s));
        MERV_HIP(quant(L.fc2_w, D, H, m.fc2_q, m.fc2_s));
        if (folded) {
            MERV_HIP(colsum(m.qkv_q, m.qkv_s, 3 * D, D, m.qkv_cs));
            MERV_HIP(colsum(m.fc1_q, m.fc1_s, H, D, m.fc1_cs));
        }
        if (e->d.temporal_frames > 0) {
            MERV_HIP(quant(folded ? (const void*)e->fl[i].tqkv_w : L.t_qkv_w, 3 * D, D, m.tqkv_q, m.tqkv_s));
            MERV_HIP(quant(L.t_proj_w, D, D, m.tproj_q, m.tproj_s));
            if (folded) {
                MERV_HIP(colsum(m.tqkv_q, m.tqkv_s, 3 * D, D, m.tqkv_cs));
                MERV_HIP(quant(L.t_qkv_w, 3 * D, D, m.tqkv_raw_q, m.tqkv_raw_s));
            }
        }
    }
    e->mx = true;
    e->mx_folded = folded;
    return 0;
}

extern "C" size_t merv_encoder_workspace_bytes(const merv_encoder* enc, int32_t batch) {
    if (!enc || batch <= 0) return 0;
    return carve(enc, batch * enc->seq_per_video, nullptr).total;
}

static GemmArgs gemm_args(const bf16_t* A, int lda, const void* W, int K, bf16_t* C, int ldc, int M, int N,
                          const float* bias, int act) {
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = A; g.lda = lda; g.W = (const bf16_t*)W; g.ldw = K; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act;
    return g;
}

extern "C" int merv_encoder_forward(const merv_encoder* e, const void* pixels, int32_t pix_dtype, int32_t batch,
                                    void* out_tokens, void* workspace, size_t workspace_bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(e, "merv_encoder_forward: null argument");
    return merv_encoder_forward_frames(e, pixels, pix_dtype, batch, e->d.frames, out_tokens, workspace, workspace_bytes, stream_);
}

extern "C" int merv_encoder_forward_frames(const merv_encoder* e, const void* pixels, int32_t pix_dtype, int32_t batch,
                                           int32_t frames, void* out_tokens, void* workspace, size_t workspace_bytes,
                                           void* stream_) {
    return merv_encoder_forward_select(e, pixels, pix_dtype, batch, frames, MERV_OUT_PATCHES, out_tokens, workspace, workspace_bytes,
                                       stream_);
}

extern "C" int merv_encoder_forward_select(const merv_encoder* e, const void* pixels, int32_t pix_dtype, int32_t batch,
                                           int32_t frames, int32_t select, void* out_tokens, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(select == MERV_OUT_PATCHES || select == MERV_OUT_ALL, "merv_encoder_forward_select: bad token selection");
    MERV_CHECK(e && pixels && out_tokens && workspace, "merv_encoder_forward: null argument");
    MERV_CHECK(batch > 0, "merv_encoder_forward: batch must be positive");
    MERV_CHECK(pix_dtype == MERV_DT_F32 || pix_dtype == MERV_DT_BF16, "merv_encoder_forward: bad pixel dtype");
    MERV_CHECK(((uintptr_t)workspace & 255) == 0, "merv_encoder_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    const merv_encoder_desc& d = e->d;
    // A frame-range unit: per-frame encoders treat every frame as its own sequence (LanguageBind: every clip of
    // `temporal_frames` frames), so any such count runs with the same weights; a joint space-time encoder (ViViT) does not split.
    MERV_CHECK(frames == d.frames || (!d.joint_space_time && frames >= 1 && frames % d.tubelet == 0 &&
                                      (d.temporal_frames == 0 || frames % d.temporal_frames == 0)),
               "merv_encoder_forward_frames: frames must equal the descriptor's count (joint space-time) or be whole clips / frames");
    const int seq_per_video = d.joint_space_time ? 1 : frames / d.tubelet;
    const int T_out = frames / d.tubelet;
    Workspace ws = carve(e, batch * seq_per_video, (char*)workspace);
    MERV_CHECK(workspace_bytes >= ws.total, "merv_encoder_forward: workspace too small");

    const int D = d.dim, nseq = batch * seq_per_video, ntok = e->ntok, M = nseq * ntok;
    const float scale = ATTN_SCALE;

    // every bf16 GEMM of this forward carries the encoder's sub-round policy (GemmArgs::subround_min_tiles): the chain that ends a concurrent step
    // takes the fast wide form for its narrow launches, the others the form that leaves it the CUs (gemm.hip, SUBROUND_MIN_TILES_BESIDE)
    auto gemm = [&](GemmArgs g) -> hipError_t {
        g.subround_min_tiles = e->latency_critical ? 0 : 32;
        return launch_gemm(g, s);
    };
    // ---- patch / tubelet embedding: im2col + GEMM (+bias +pos), rows scattered past the prefix tokens ----
    {
        Im2colArgs ic;
        ic.pix = pixels; ic.pix_is_bf16 = (pix_dtype == MERV_DT_BF16); ic.out = ws.h;
        ic.B = batch; ic.frames = frames; ic.img = d.img; ic.patch = d.patch; ic.tt = d.tubelet; ic.kpad = d.k_pad;
        const long long hw = (long long)d.img * d.img;
        if (d.pix_layout == MERV_PIX_BCFHW) { ic.sB = 3LL * frames * hw; ic.sC = frames * hw; ic.sF = hw; }
        else { ic.sB = 3LL * frames * hw; ic.sF = 3 * hw; ic.sC = hw; }
        MERV_HIP(launch_im2col(ic, s));
        GemmArgs g = gemm_args(ws.h, d.k_pad, e->w.patch_w, d.k_pad, ws.x, D, nseq * e->P, D, e->w.patch_b, ACT_NONE);
        g.res = (const bf16_t*)e->w.pos; g.ldres = D; g.res_row_mod = e->P;
        g.out_group = e->P; g.out_stride = ntok; g.out_off = d.prefix_tokens;
        MERV_HIP(gemm(g));
        if (d.prefix_tokens > 0) {
            PrefixArgs pa{(const bf16_t*)e->w.prefix, ws.x, nseq, ntok, d.prefix_tokens, D};
            MERV_HIP(launch_prefix(pa, s));
        }
        if (d.pre_ln) {
            LayerNormArgs ln{ws.x, ws.x, e->w.pre_ln_w, e->w.pre_ln_b, nullptr, M, D, 1, 1, d.ln_eps};
            MERV_HIP(launch_layernorm(ln, s));
        }
    }

    // MXFP8 mode. Every GEMM input arrives quantised from its producer: the LayerNorms, the attention kernels and fc1's
    // epilogue write e4m3 + block scales directly (no separate quantisation pass, no bf16 copy of those tensors).
    const int mx_groups = (M + 63) / 64;
    const bool mx_qkv = e->mx && (e->mx_mask & 1), mx_proj = e->mx && (e->mx_mask & 2), mx_fc1 = e->mx && (e->mx_mask & 4),
               mx_fc2 = e->mx && (e->mx_mask & 8);
    auto mx_gemm = [&](GemmArgs g, const uint8_t* aq, const uint8_t* asc, const uint8_t* wq, const uint8_t* wsc) -> hipError_t {
        g.A = (const bf16_t*)aq; g.lda = g.K;
        g.W = (const bf16_t*)wq; g.ldw = g.K;
        g.mx_scale_a = asc; g.mx_scale_w = wsc; g.mx_groups_a = mx_groups; g.mx_groups_w = g.N / 64;
        return launch_gemm_mx(g, s);
    };

    // Folded LayerNorm statistics of the residual stream. The GEMM that last wrote ws.x (t_proj / proj / fc2) leaves per-row
    // partials in ws.parts from its epilogue (GemmArgs::stats_out); a one-thread-per-row kernel combines them. Only when x
    // was written by something else (the embedding, an in-place temporal-embedding add) is the stream re-read.
    bool parts_valid = false;
    auto ln_stats = [&]() -> hipError_t {
        if (parts_valid) {
            // (timing-only probe, hooks build: the finalize launch issued TWICE -- same result; the step's slow-down is what the launches cost on
            // their chains, i.e. the most that folding them into a GEMM could gain. Dropping them instead leaves another layer's statistics: NaNs)
            static const bool abl_twice = merv_tuning_env("MERV_ABL_DOUBLE_FINALIZE") != nullptr;
            if (abl_twice) {
                StatsFinalizeArgs fb{ws.parts, ws.stats, M, D / 64, d.ln_eps};
                if (hipError_t err = launch_stats_finalize(fb, s); err != hipSuccess) return err;
            }
            StatsFinalizeArgs fa{ws.parts, ws.stats, M, D / 64, d.ln_eps};
            return launch_stats_finalize(fa, s);
        }
        RowStatsArgs rs{ws.x, ws.stats, M, D, d.ln_eps};
        return launch_row_stats(rs, s);
    };
    // does a folded LayerNorm read x right after the GEMM that writes it here? (then that GEMM produces the partials)
    static const bool fused_stats = !(merv_tuning_env("MERV_LN_FUSED_STATS") && merv_tuning_env("MERV_LN_FUSED_STATS")[0] == '0');  // A/B hook
    // MXFP8 mode on the folded weights (merv_encoder_enable_mxfp8 after merv_encoder_enable_ln_fold): the MX qkv / fc1 are folded like the bf16
    // ones -- A = the MXFP8 copy of the raw stream (ws.xq, written by the producing GEMM's epilogue beside bf16 x), W = the quantised
    // bf16(W * gamma), normalisation in the epilogue with the column sums of the DE-QUANTISED weight. Without the fold's weights (mx_folded
    // false) an MX qkv / fc1 keeps its LayerNorm + quantisation kernel.
    const bool mxf = e->mx && e->mx_folded;
    const bool folded_qkv = e->fold && (!mx_qkv || mxf), folded_fc1 = e->fold && (!mx_fc1 || mxf);
    const bool fold_qkv = folded_qkv && fused_stats, fold_fc1 = folded_fc1 && fused_stats;
    // LanguageBind: the temporal LayerNorm of blocks 1 .. folds into the temporal qkv GEMM as well -- the previous block's fc2 epilogue
    // adds this block's temporal embedding into x (GemmArgs::row_add) and leaves the statistics partials, so no kernel re-reads x.
    // Block 0 keeps the LayerNorm kernel (its x comes from the embedding / pre-LayerNorm).
    static const bool fold_t_env = !(merv_tuning_env("MERV_LN_FOLD_TEMPORAL") && merv_tuning_env("MERV_LN_FOLD_TEMPORAL")[0] == '0');  // A/B hook
    const bool fold_tqkv = fold_qkv && d.temporal_frames > 0 && ntok >= 256 && fold_t_env;
    // the stream's MXFP8 copy: valid when the GEMM that last wrote x left it (produce_x); else one quantisation pass makes it (block 0)
    bool xq_valid = false;
    auto need_xq = [&]() -> hipError_t {
        if (xq_valid) return hipSuccess;
        xq_valid = true;
        MxQuantArgs qa{ws.x, ws.xq, ws.xsc, M, D, D};
        return launch_mx_quantize(qa, s);
    };
    // `o` writes the residual stream; the next reader is a folded LayerNorm (`stats`) whose GEMM runs on MXFP8 operands (`mx_copy`)
    auto produce_x = [&](GemmArgs& o, bool stats, bool mx_copy) {
        if (stats) o.stats_out = ws.parts;
        if (stats && mx_copy) { o.mx_out_q = ws.xq; o.mx_out_scales = ws.xsc; o.mx_out_groups = mx_groups; o.mx_out_keep_c = 1; }
        parts_valid = stats;
        xq_valid = stats && mx_copy;
    };
    // a LayerNorm-folded consumer of x: statistics, then either the bf16 fold's operands or (MX) the stream's MXFP8 copy
    auto fold_consumer = [&](GemmArgs& g, bool is_mx, const bf16_t* wf, const float* db, const float* cs, const float* cs_mx) -> hipError_t {
        if (hipError_t err = ln_stats(); err != hipSuccess) return err;
        g.bias = db; g.row_stats = ws.stats;
        if (is_mx) { g.ln_colsum = cs_mx; return need_xq(); }
        g.A = ws.x; g.W = wf; g.ln_colsum = cs;
        return hipSuccess;
    };

    // ---- transformer blocks ----
    for (int li = 0; li < d.layers; ++li) {
        const merv_layer_weights& L = e->layers[li];
        if (d.temporal_frames > 0) {
            // x += temporal_embedding[t]; x += out_proj(temporal_attn(LN_t(x)))   (modeling_video.py:133-155)
            GemmArgs q = gemm_args(ws.y, D, L.t_qkv_w, D, ws.qkv, 3 * D, M, 3 * D, L.t_qkv_b, ACT_NONE);
            const bool tfold = fold_tqkv && li > 0;  // x already carries temporal_embedding (previous fc2) and its partials are in ws.parts
            if (tfold) {
                MERV_HIP(fold_consumer(q, mx_qkv, e->fl[li].tqkv_w, e->fl[li].tqkv_db, e->fl[li].tqkv_cs, mx_qkv ? e->mxl[li].tqkv_cs : nullptr));
            } else {
                LayerNormArgs ln{ws.x, ws.y, L.t_ln_w, L.t_ln_b, L.t_emb, M, D, ntok, d.temporal_frames, d.ln_eps};
                if (mx_qkv) { ln.mx_q = ws.aq; ln.mx_scales = ws.asc; ln.mx_groups = mx_groups; }
                MERV_HIP(launch_layernorm(ln, s));
                parts_valid = xq_valid = false;  // (the kernel added the temporal embedding into x)
            }
            if (mx_qkv) {
                if (tfold) MERV_HIP(mx_gemm(q, ws.xq, ws.xsc, e->mxl[li].tqkv_q, e->mxl[li].tqkv_s));
                else if (mxf) MERV_HIP(mx_gemm(q, ws.aq, ws.asc, e->mxl[li].tqkv_raw_q, e->mxl[li].tqkv_raw_s));  // (tqkv_q holds the folded weight)
                else MERV_HIP(mx_gemm(q, ws.aq, ws.asc, e->mxl[li].tqkv_q, e->mxl[li].tqkv_s));
            } else {
                MERV_HIP(gemm(q));
            }
            TemporalAttnArgs ta{ws.qkv, ws.y, nseq / d.temporal_frames, d.temporal_frames, ntok, d.heads, D, scale};
            if (mx_proj) { ta.mx_q = ws.aq; ta.mx_scales = ws.asc; ta.mx_groups = mx_groups; }
            MERV_HIP(launch_temporal_attention(ta, s));
            GemmArgs o = gemm_args(ws.y, D, L.t_proj_w, D, ws.x, D, M, D, L.t_proj_b, ACT_NONE);
            o.res = ws.x; o.ldres = D;
            produce_x(o, fold_qkv, mx_qkv);  // LN1 reads this x next
            if (mx_proj) MERV_HIP(mx_gemm(o, ws.aq, ws.asc, e->mxl[li].tproj_q, e->mxl[li].tproj_s));
            else MERV_HIP(gemm(o));
        }
        {
            const bool folded = folded_qkv;  // no LayerNorm pass; the normalisation is algebra in the GEMM epilogue
            GemmArgs q = gemm_args(ws.y, D, L.qkv_w, D, ws.qkv, 3 * D, M, 3 * D, L.qkv_b, ACT_NONE);
            if (folded) {
                MERV_HIP(fold_consumer(q, mx_qkv, e->fl[li].qkv_w, e->fl[li].qkv_db, e->fl[li].qkv_cs, mx_qkv ? e->mxl[li].qkv_cs : nullptr));
            } else {
                LayerNormArgs ln{ws.x, ws.y, L.ln1_w, L.ln1_b, nullptr, M, D, 1, 1, d.ln_eps};
                if (mx_qkv) { ln.mx_q = ws.aq; ln.mx_scales = ws.asc; ln.mx_groups = mx_groups; }
                MERV_HIP(launch_layernorm(ln, s));
            }
            if (mx_qkv) MERV_HIP(mx_gemm(q, folded ? ws.xq : ws.aq, folded ? ws.xsc : ws.asc, e->mxl[li].qkv_q, e->mxl[li].qkv_s));
            else MERV_HIP(gemm(q));
            AttnArgs at{ws.qkv, ws.y, nseq, ntok, d.heads, D, scale};
            at.q_prescaled = folded;  // the folded qkv weight carries scale * log2(e) in its q rows
            if (mx_proj) { at.mx_q = ws.aq; at.mx_scales = ws.asc; at.mx_groups = mx_groups; }
            MERV_HIP(launch_attention(at, s));
            GemmArgs o = gemm_args(ws.y, D, L.proj_w, D, ws.x, D, M, D, L.proj_b, ACT_NONE);
            o.res = ws.x; o.ldres = D; o.lscale = d.layerscale ? L.ls1 : nullptr;
            produce_x(o, fold_fc1, mx_fc1);  // LN2 reads this x next
            if (mx_proj) MERV_HIP(mx_gemm(o, ws.aq, ws.asc, e->mxl[li].proj_q, e->mxl[li].proj_s));
            else MERV_HIP(gemm(o));
        }
        {
            const bool folded = folded_fc1;
            GemmArgs f1 = gemm_args(ws.y, D, L.fc1_w, D, ws.h, d.mlp_dim, M, d.mlp_dim, L.fc1_b, d.act);
            if (folded) {
                MERV_HIP(fold_consumer(f1, mx_fc1, e->fl[li].fc1_w, e->fl[li].fc1_db, e->fl[li].fc1_cs, mx_fc1 ? e->mxl[li].fc1_cs : nullptr));
            } else {
                LayerNormArgs ln{ws.x, ws.y, L.ln2_w, L.ln2_b, nullptr, M, D, 1, 1, d.ln_eps};
                if (mx_fc1) { ln.mx_q = ws.aq; ln.mx_scales = ws.asc; ln.mx_groups = mx_groups; }
                MERV_HIP(launch_layernorm(ln, s));
            }
            if (mx_fc2) { f1.mx_out_q = ws.hq; f1.mx_out_scales = ws.hsc; f1.mx_out_groups = mx_groups; }  // fc2's input, whichever kernel runs fc1
            if (mx_fc1) MERV_HIP(mx_gemm(f1, folded ? ws.xq : ws.aq, folded ? ws.xsc : ws.asc, e->mxl[li].fc1_q, e->mxl[li].fc1_s));
            else MERV_HIP(gemm(f1));
            GemmArgs f2 = gemm_args(ws.h, d.mlp_dim, L.fc2_w, d.mlp_dim, ws.x, D, M, D, L.fc2_b, ACT_NONE);
            f2.res = ws.x; f2.ldres = D; f2.lscale = d.layerscale ? L.ls2 : nullptr;
            // the next reader of x is the following block's LN1 -- unless that block starts with the temporal sub-block
            // (its LayerNorm kernel first adds the temporal embedding into x) or this was the last block
            const bool next_ln1 = fold_qkv && li + 1 < d.layers && (d.temporal_frames == 0 || fold_tqkv);
            produce_x(f2, next_ln1, mx_qkv);
            if (next_ln1 && d.temporal_frames > 0) {  // x += the NEXT block's temporal_embedding[frame of the row], here
                f2.row_add = e->layers[li + 1].t_emb; f2.row_add_div = ntok; f2.row_add_mod = d.temporal_frames;
            }
            if (mx_fc2) MERV_HIP(mx_gemm(f2, ws.hq, ws.hsc, e->mxl[li].fc2_q, e->mxl[li].fc2_s));
            else MERV_HIP(gemm(f2));
        }
    }

    // ---- output selection: optional final LayerNorm, strip prefix tokens ----
    const bf16_t* src = ws.x;
    if (d.final_ln) {
        LayerNormArgs ln{ws.x, ws.y, e->w.final_ln_w, e->w.final_ln_b, nullptr, M, D, 1, 1, d.ln_eps};
        MERV_HIP(launch_layernorm(ln, s));
        src = ws.y;
    }
    GatherTokensArgs ga;
    ga.x = src; ga.out = (bf16_t*)out_tokens; ga.B = batch; ga.T = T_out; ga.S = e->S_out; ga.D = D;
    ga.prefix = d.prefix_tokens;
    if (d.joint_space_time) { ga.bstride = ntok; ga.fstride = e->S_out; }
    else { ga.bstride = seq_per_video * ntok; ga.fstride = ntok; }
    if (select == MERV_OUT_ALL) {  // every token of every sequence, prefix tokens first: a plain copy of the stream
        ga.prefix = 0; ga.T = seq_per_video; ga.S = ntok; ga.bstride = seq_per_video * ntok; ga.fstride = ntok;
    }
    MERV_HIP(launch_gather_tokens(ga, s));
    return 0;
}

extern "C" int merv_projector_forward(const void* tokens, int32_t batch, int32_t T, int32_t S, int32_t C,
                                      int32_t out_size, const void* proj_w, const float* proj_b, int32_t llm_dim,
                                      void* pooled_ws, void* out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(tokens && proj_w && pooled_ws && out, "merv_projector_forward: null argument");
    MERV_CHECK(batch > 0 && T > 0 && S > 0 && out_size > 0, "merv_projector_forward: bad geometry");  // S < out_size: AdaptiveAvgPool replicates (window rule unchanged)
    MERV_CHECK(C % 64 == 0 && llm_dim % 128 == 0, "merv_projector_forward: C % 64 and llm_dim % 128 required");
    hipStream_t s = (hipStream_t)stream_;
    PoolArgs pa{(const bf16_t*)tokens, (bf16_t*)pooled_ws, batch, T, S, out_size, C};
    MERV_HIP(launch_pool(pa, s));
    const int M = batch * T * out_size * out_size;
    GemmArgs g = gemm_args((const bf16_t*)pooled_ws, C, proj_w, C, (bf16_t*)out, llm_dim, M, llm_dim, proj_b, ACT_NONE);
    MERV_HIP(launch_gemm(g, s));
    return 0;
}

extern "C" size_t merv_fusion_workspace_floats(int32_t batch, int32_t E, int32_t T) {
    if (batch <= 0 || E <= 0 || T <= 0) return 0;
    return (size_t)fusion_partial_floats(batch, E, T);
}

extern "C" int merv_fusion_forward(const void* const* v, int32_t E, int32_t batch, int32_t T, int32_t C, const float* u,
                                   float* partial_ws, float* weights_out, void* out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(v && u && partial_ws && weights_out && out, "merv_fusion_forward: null argument");
    MERV_CHECK(E >= 1 && E <= 8, "merv_fusion_forward: 1..8 encoders supported");
    MERV_CHECK(batch > 0 && T > 0 && C > 0 && C % 8 == 0, "merv_fusion_forward: bad geometry");
    FusionArgs fa;
    memset(&fa, 0, sizeof fa);
    for (int e = 0; e < E; ++e) {
        MERV_CHECK(v[e], "merv_fusion_forward: null encoder tensor");
        fa.v[e] = (const bf16_t*)v[e];
    }
    fa.E = E; fa.B = batch; fa.T = T; fa.C = C; fa.u = u; fa.partial = partial_ws; fa.weights = weights_out;
    fa.out = (bf16_t*)out;
    MERV_HIP(launch_fusion(fa, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_splice_forward(const void* emb, const void* vis, int32_t batch, int32_t S, int32_t T, int32_t C,
                                   int32_t bos, void* out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(emb && vis && out, "merv_splice_forward: null argument");
    MERV_CHECK(batch > 0 && S >= 0 && T >= 0 && C % 8 == 0 && bos >= 0 && bos <= S, "merv_splice_forward: bad geometry");
    SpliceArgs sa{(const bf16_t*)emb, (const bf16_t*)vis, (bf16_t*)out, batch, S, T, C, bos};
    MERV_HIP(launch_splice(sa, (hipStream_t)stream_));
    return 0;
}

// ---- backward of the trainable tail (row f-4) ----
extern "C" size_t merv_fusion_backward_workspace_floats(int32_t batch, int32_t E, int32_t T, int32_t C) {
    if (batch <= 0 || E <= 0 || T <= 0 || C <= 0) return 0;
    return fusion_bwd_workspace_floats(batch, E, T, C);
}

extern "C" int merv_fusion_backward_reduce(const void* const* v, int32_t E, int32_t batch, int32_t T, int32_t C,
                                           const void* grad_out, float* ws, float* dw_out, float* vbar_out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(v && grad_out && ws && dw_out && vbar_out, "merv_fusion_backward_reduce: null argument");
    MERV_CHECK(E >= 1 && E <= 8, "merv_fusion_backward_reduce: 1..8 encoders supported");
    MERV_CHECK(batch > 0 && T > 0 && C > 0 && C % 8 == 0, "merv_fusion_backward_reduce: bad geometry");
    FusionBwdArgs a;
    memset(&a, 0, sizeof a);
    for (int e = 0; e < E; ++e) {
        MERV_CHECK(v[e], "merv_fusion_backward_reduce: null encoder tensor");
        a.v[e] = (const bf16_t*)v[e];
    }
    a.grad_out = (const bf16_t*)grad_out;
    a.E = E; a.B = batch; a.T = T; a.C = C; a.dw = dw_out; a.vbar = vbar_out;
    MERV_HIP(launch_fusion_bwd_reduce(a, ws, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_fusion_backward_mix(const void* grad_out, const float* weights, const float* ds, const float* u,
                                        int32_t E, int32_t batch, int32_t T, int32_t C, void* const* dv_out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(grad_out && weights && ds && u && dv_out, "merv_fusion_backward_mix: null argument");
    MERV_CHECK(E >= 1 && E <= 8, "merv_fusion_backward_mix: 1..8 encoders supported");
    MERV_CHECK(batch > 0 && T > 0 && C > 0 && C % 8 == 0, "merv_fusion_backward_mix: bad geometry");
    FusionBwdMixArgs a;
    memset(&a, 0, sizeof a);
    for (int e = 0; e < E; ++e) {
        MERV_CHECK(dv_out[e], "merv_fusion_backward_mix: null output tensor");
        a.dv[e] = (bf16_t*)dv_out[e];
    }
    a.grad_out = (const bf16_t*)grad_out; a.w = weights; a.ds = ds; a.u = u;
    a.E = E; a.B = batch; a.T = T; a.C = C;
    MERV_HIP(launch_fusion_bwd_mix(a, (hipStream_t)stream_));
    return 0;
}

static inline int pad64(int v) { return (v + 63) / 64 * 64; }

extern "C" size_t merv_projector_backward_workspace_bytes(int32_t M, int32_t C, int32_t llm) {
    if (M <= 0 || C <= 0 || llm <= 0) return 0;
    const size_t mp = (size_t)pad64(M);
    return align_up((size_t)llm * mp * 2, 256) + align_up((size_t)C * mp * 2, 256) + align_up(colsum_workspace_floats(llm) * 4, 256);
}

extern "C" int merv_projector_backward(const void* grad_out, const void* pooled, int32_t M, int32_t C, int32_t llm, void* ws,
                                       size_t ws_bytes, void* grad_w, float* grad_b, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(grad_out && pooled && ws && grad_w && grad_b, "merv_projector_backward: null argument");
    MERV_CHECK(M > 0 && C > 0 && C % 128 == 0 && llm > 0 && llm % 8 == 0, "merv_projector_backward: need C % 128 == 0, llm % 8 == 0");
    MERV_CHECK(ws_bytes >= merv_projector_backward_workspace_bytes(M, C, llm), "merv_projector_backward: workspace too small");
    MERV_CHECK(((uintptr_t)ws & 255) == 0, "merv_projector_backward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    const int mp = pad64(M);
    char* base = (char*)ws;
    bf16_t* gT = (bf16_t*)base;                                   // [llm, mp]
    bf16_t* pT = (bf16_t*)(base + align_up((size_t)llm * mp * 2, 256));  // [C, mp]
    float* cs = (float*)((char*)pT + align_up((size_t)C * mp * 2, 256));
    TransposeArgs tg{(const bf16_t*)grad_out, gT, M, llm, llm, mp, mp};
    MERV_HIP(launch_transpose(tg, s));
    TransposeArgs tp{(const bf16_t*)pooled, pT, M, C, C, mp, mp};
    MERV_HIP(launch_transpose(tp, s));
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = gT; g.W = pT; g.C = (bf16_t*)grad_w;
    g.M = llm; g.N = C; g.K = mp; g.lda = mp; g.ldw = mp; g.ldc = C; g.act = ACT_NONE;
    MERV_HIP(launch_gemm(g, s));
    ColsumArgs c{(const bf16_t*)grad_out, cs, grad_b, M, llm, llm};
    MERV_HIP(launch_colsum(c, s));
    return 0;
}

extern "C" int merv_transpose_bf16(const void* in, int32_t R, int32_t C, int32_t ldi, void* out, int32_t ldo, int32_t Rpad,
                                   void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(in && out, "merv_transpose_bf16: null argument");
    MERV_CHECK(R > 0 && C > 0 && Rpad >= R && Rpad % 2 == 0 && ldi % 2 == 0 && ldo % 2 == 0 && ldi >= C && ldo >= Rpad,
               "merv_transpose_bf16: bad geometry (ldi, ldo, Rpad even; ldi >= C; ldo >= Rpad >= R)");
    TransposeArgs t{(const bf16_t*)in, (bf16_t*)out, R, C, ldi, ldo, Rpad};
    MERV_HIP(launch_transpose(t, (hipStream_t)stream_));
    return 0;
}

// ---- MXFP8 mode ----
extern "C" size_t merv_mxfp8_scale_bytes(int32_t rows, int32_t K) {
    if (rows <= 0 || K <= 0 || K % 128 != 0) return 0;
    return mx_scale_bytes(rows, K);
}

extern "C" int merv_quantize_mxfp8(const void* x, int32_t rows, int32_t K, int32_t ld, void* q, void* scales, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(x && q && scales, "merv_quantize_mxfp8: null argument");
    MERV_CHECK(rows > 0 && K > 0 && K % 128 == 0 && ld % 8 == 0 && ld >= K, "merv_quantize_mxfp8: need K % 128 == 0, ld % 8 == 0, ld >= K");
    MxQuantArgs a{(const bf16_t*)x, (uint8_t*)q, (uint8_t*)scales, rows, K, ld};
    MERV_HIP(launch_mx_quantize(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_gemm_mxfp8(const void* A8, const void* scale_a, const void* W8, const void* scale_w, void* C, const float* bias,
                               const float* lscale, const void* res, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                               int32_t ldc, int32_t ldres, int32_t act, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A8 && scale_a && W8 && scale_w && C, "merv_gemm_mxfp8: null argument");
    MERV_CHECK(M > 0 && N > 0 && N % 256 == 0 && K >= 512 && K % 256 == 0, "merv_gemm_mxfp8: need N % 256 == 0, K % 256 == 0, K >= 512");
    MERV_CHECK(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldc % 8 == 0 && ldc >= N, "merv_gemm_mxfp8: bad leading dimension");
    MERV_CHECK(act >= ACT_NONE && act <= ACT_QUICK_GELU, "merv_gemm_mxfp8: unknown activation");
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A8; g.W = (const bf16_t*)W8; g.C = (bf16_t*)C; g.bias = bias; g.lscale = lscale; g.res = (const bf16_t*)res;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldres = ldres; g.act = act;
    g.mx_scale_a = scale_a; g.mx_scale_w = scale_w; g.mx_groups_a = (M + 63) / 64; g.mx_groups_w = N / 64;
    MERV_HIP(launch_gemm_mx(g, (hipStream_t)stream_));
    return 0;
}

// test hook: bf16 GEMM (automatic tile choice, incl. the split plan) whose epilogue writes MXFP8 instead of bf16
extern "C" int merv_debug_gemm_mx_out(const void* A, const void* W, void* C_unused, int32_t M, int32_t N, int32_t K, void* q_out,
                                      void* scales_out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A && W && q_out && scales_out, "merv_debug_gemm_mx_out: null argument");
    MERV_CHECK(N % 128 == 0 && K % 64 == 0, "merv_debug_gemm_mx_out: bad geometry");
    GemmArgs g = gemm_args((const bf16_t*)A, K, W, K, (bf16_t*)C_unused, N, M, N, nullptr, ACT_NONE);
    g.mx_out_q = (uint8_t*)q_out; g.mx_out_scales = (uint8_t*)scales_out; g.mx_out_groups = (M + 63) / 64;
    MERV_HIP(launch_gemm(g, (hipStream_t)stream_));
    return 0;
}

// test hook: merv_gemm_bf16's epilogue set plus the LayerNorm-partials output the encoder requests from the GEMMs that write
// its residual stream (GemmArgs::stats_out): lets the parity tests check the epilogue statistics through the C ABI.
extern "C" int merv_debug_gemm_mxfp8_forms(const void* A8, const void* scale_a, const void* W8, const void* scale_w, void* C, const float* bias,
                                          const float* lscale, const void* res, int32_t M, int32_t N, int32_t K, int32_t act,
                                          const float* row_stats, const float* ln_colsum, float* stats_out, const float* row_add,
                                          int32_t row_add_div, int32_t row_add_mod, void* mx_out_q, void* mx_out_scales, int32_t keep_c,
                                          int32_t no_static_form, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A8 && scale_a && W8 && scale_w && C, "merv_debug_gemm_mxfp8_forms: null argument");
    MERV_CHECK(M > 0 && N > 0 && N % 256 == 0 && K >= 512 && K % 256 == 0, "merv_debug_gemm_mxfp8_forms: need N % 256 == 0, K % 256 == 0, K >= 512");
    MERV_CHECK(act >= ACT_NONE && act <= ACT_QUICK_GELU, "merv_debug_gemm_mxfp8_forms: unknown activation");
    MERV_CHECK(!mx_out_q || mx_out_scales, "merv_debug_gemm_mxfp8_forms: an MXFP8 output needs its scale array");
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A8; g.W = (const bf16_t*)W8; g.C = (bf16_t*)C; g.bias = bias; g.lscale = lscale; g.res = (const bf16_t*)res;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.ldres = N; g.act = act;
    g.mx_scale_a = scale_a; g.mx_scale_w = scale_w; g.mx_groups_a = (M + 63) / 64; g.mx_groups_w = N / 64;
    g.row_stats = row_stats; g.ln_colsum = ln_colsum; g.stats_out = stats_out;
    g.row_add = row_add; g.row_add_div = row_add_div; g.row_add_mod = row_add_mod;
    g.mx_out_q = (uint8_t*)mx_out_q; g.mx_out_scales = (uint8_t*)mx_out_scales; g.mx_out_groups = (M + 63) / 64; g.mx_out_keep_c = keep_c;
    g.no_static_form = no_static_form;
    MERV_HIP(launch_gemm_mx(g, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_debug_gemm_stats(const void* A, const void* W, void* C, const float* bias, const float* lscale, const void* res,
                                     int32_t M, int32_t N, int32_t K, int32_t act, float* stats_out, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A && W && C && stats_out, "merv_debug_gemm_stats: null argument");
    MERV_CHECK(K > 0 && K % 64 == 0 && N > 0 && N % 128 == 0, "merv_debug_gemm_stats: K % 64 == 0 and N % 128 == 0 required");
    MERV_CHECK(act >= ACT_NONE && act <= ACT_QUICK_GELU, "merv_debug_gemm_stats: unknown activation");
    GemmArgs g = gemm_args((const bf16_t*)A, K, W, K, (bf16_t*)C, N, M, N, bias, act);
    g.lscale = lscale; g.res = (const bf16_t*)res; g.ldres = N; g.stats_out = stats_out;
    MERV_HIP(launch_gemm(g, (hipStream_t)stream_));
    return 0;
}

// test hook: the same with the row-indexed add of GemmArgs::row_add (table fp32 [row_add_mod, N]) behind the residual add
extern "C" int merv_debug_gemm_row_add(const void* A, const void* W, void* C, const float* bias, const void* res, int32_t M, int32_t N,
                                       int32_t K, const float* row_add, int32_t row_add_div, int32_t row_add_mod, float* stats_out,
                                       void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A && W && C && row_add && stats_out, "merv_debug_gemm_row_add: null argument");
    MERV_CHECK(K > 0 && K % 64 == 0 && N > 0 && N % 128 == 0, "merv_debug_gemm_row_add: K % 64 == 0 and N % 128 == 0 required");
    MERV_CHECK(row_add_div >= 256 && row_add_mod > 0, "merv_debug_gemm_row_add: row_add_div >= 256 and row_add_mod > 0 required");
    GemmArgs g = gemm_args((const bf16_t*)A, K, W, K, (bf16_t*)C, N, M, N, bias, ACT_NONE);
    g.res = (const bf16_t*)res; g.ldres = N; g.stats_out = stats_out;
    g.row_add = row_add; g.row_add_div = row_add_div; g.row_add_mod = row_add_mod;
    MERV_HIP(launch_gemm(g, (hipStream_t)stream_));
    return 0;
}

// ---- frame preprocessing (row a3) ----
extern "C" size_t merv_preprocess_workspace_bytes(int32_t T, int32_t H, int32_t W, int32_t out_size) {
    if (T <= 0 || H <= 0 || W <= 0 || out_size <= 0) return 0;
    return pil_workspace_bytes(T, H, W, out_size);
}

extern "C" int merv_preprocess_pil(const void* frames_u8, int32_t T, int32_t H, int32_t W, int32_t out_size, int32_t filter,
                                   const float* mean3, const float* std3, void* out_pixels, int32_t out_dtype, void* resized_u8,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(frames_u8 && mean3 && std3 && workspace && (out_pixels || resized_u8), "merv_preprocess_pil: null argument");
    MERV_CHECK(T > 0 && H > 0 && W > 0 && out_size > 0, "merv_preprocess_pil: bad geometry");
    MERV_CHECK(filter == 0 || filter == 1, "merv_preprocess_pil: filter must be 0 (bilinear) or 1 (bicubic)");
    MERV_CHECK(out_dtype == MERV_DT_F32 || out_dtype == MERV_DT_BF16, "merv_preprocess_pil: bad output dtype");
    MERV_CHECK(((uintptr_t)workspace & 255) == 0, "merv_preprocess_pil: workspace must be 256-byte aligned");
    MERV_CHECK(workspace_bytes >= pil_workspace_bytes(T, H, W, out_size), "merv_preprocess_pil: workspace too small");
    MERV_HIP(launch_pil_resize_normalize((const uint8_t*)frames_u8, T, H, W, out_size, filter, mean3, std3, out_pixels,
                                         out_dtype == MERV_DT_BF16, (uint8_t*)resized_u8, (char*)workspace, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_preprocess_languagebind(const void* frames_u8, int32_t T, int32_t H, int32_t W, int32_t out_size, int32_t flip,
                                            const float* mean3, const float* std3, void* out_pixels, int32_t out_dtype, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(frames_u8 && mean3 && std3 && out_pixels, "merv_preprocess_languagebind: null argument");
    MERV_CHECK(T > 0 && H > 0 && W > 0 && out_size > 0, "merv_preprocess_languagebind: bad geometry");
    MERV_CHECK(out_dtype == MERV_DT_F32 || out_dtype == MERV_DT_BF16, "merv_preprocess_languagebind: bad output dtype");
    MERV_HIP(launch_languagebind_transform((const uint8_t*)frames_u8, T, H, W, out_size, flip != 0, mean3, std3, out_pixels,
                                           out_dtype == MERV_DT_BF16, (hipStream_t)stream_));
    return 0;
}

// Tuning / test hook: force a GEMM tile configuration (0 = automatic choice).
extern "C" int merv_tuning_hooks(void) { return MERV_HOOKS ? 1 : 0; }
extern "C" void merv_debug_set_gemm_variant(int32_t v) { set_gemm_variant(v); }
extern "C" void merv_debug_set_attn_rescale_thr(float thr) { set_attn_rescale_thr(thr); }
extern "C" void merv_debug_set_rest_fork(void* main_stream, void* aux_stream) { set_rest_fork((hipStream_t)main_stream, (hipStream_t)aux_stream); }

// ---- single-kernel wrappers ----
extern "C" int merv_gemm_bf16(const void* A, const void* W, void* C, const float* bias, const float* lscale,
                              const void* res, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc,
                              int32_t ldres, int32_t res_row_mod, int32_t out_group, int32_t out_stride, int32_t out_off,
                              int32_t act, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(A && W && C, "merv_gemm_bf16: null argument");
    MERV_CHECK(K > 0 && K % 64 == 0 && N > 0 && N % 128 == 0, "merv_gemm_bf16: K % 64 == 0 and N % 128 == 0 required");
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = (bf16_t*)C; g.bias = bias; g.lscale = lscale;
    g.res = (const bf16_t*)res; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldres = ldres;
    g.res_row_mod = res_row_mod; g.out_group = out_group; g.out_stride = out_stride; g.out_off = out_off; g.act = act;
    MERV_HIP(launch_gemm(g, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_layernorm(void* x, void* y, const float* gamma, const float* beta, const float* add, int32_t M,
                              int32_t D, int32_t add_div, int32_t add_mod, float eps, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(x && y && gamma && beta, "merv_layernorm: null argument");
    LayerNormArgs ln{(bf16_t*)x, (bf16_t*)y, gamma, beta, add, M, D, add_div, add_mod, eps};
    MERV_HIP(launch_layernorm(ln, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_attention(const void* qkv, void* out, int32_t nseq, int32_t L, int32_t heads, int32_t D, float scale,
                              void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(qkv && out, "merv_attention: null argument");
    AttnArgs a{(const bf16_t*)qkv, (bf16_t*)out, nseq, L, heads, D, scale};
    MERV_HIP(launch_attention(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_temporal_attention(const void* qkv, void* out, int32_t nclips, int32_t t, int32_t ntok, int32_t heads,
                                       int32_t D, float scale, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(qkv && out, "merv_temporal_attention: null argument");
    TemporalAttnArgs a{(const bf16_t*)qkv, (bf16_t*)out, nclips, t, ntok, heads, D, scale};
    MERV_HIP(launch_temporal_attention(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_im2col(const void* pix, int32_t pix_dtype, void* out, int32_t B, int32_t frames, int32_t img,
                           int32_t patch, int32_t tubelet, int32_t k_pad, int64_t sB, int64_t sF, int64_t sC,
                           void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(pix && out, "merv_im2col: null argument");
    Im2colArgs ic;
    ic.pix = pix; ic.pix_is_bf16 = (pix_dtype == MERV_DT_BF16); ic.out = (bf16_t*)out; ic.B = B; ic.frames = frames;
    ic.img = img; ic.patch = patch; ic.tt = tubelet; ic.kpad = k_pad; ic.sB = sB; ic.sF = sF; ic.sC = sC;
    MERV_HIP(launch_im2col(ic, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_pool3d(const void* tokens, void* out, int32_t B, int32_t T, int32_t S, int32_t out_size, int32_t C,
                           void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(tokens && out, "merv_pool3d: null argument");
    PoolArgs pa{(const bf16_t*)tokens, (bf16_t*)out, B, T, S, out_size, C};
    MERV_HIP(launch_pool(pa, (hipStream_t)stream_));
    return 0;
}

// ---- batch-1 token decode (row f-3): the per-token forward of LlamaForCausalLM / MistralForCausalLM as HBM streams ----
extern "C" int merv_decode_rmsnorm(const void* x, const void* w, void* y, int32_t rows, int32_t D, float eps, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(x && w && y, "merv_decode_rmsnorm: null argument");
    MERV_CHECK(rows > 0 && D > 0 && D % 8 == 0, "merv_decode_rmsnorm: rows > 0 and D % 8 == 0 required");
    DecodeRmsArgs a{(const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, D, eps};
    MERV_HIP(launch_decode_rmsnorm(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_gemv(const void* W, const void* W2, const void* x, const void* res, void* y, float* y32, int32_t N,
                                int32_t K, const void* norm_w, float norm_eps, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(W && x && (y || y32), "merv_decode_gemv: null argument");
    MERV_CHECK(N > 0 && K > 0 && K % 8 == 0, "merv_decode_gemv: N > 0 and K % 8 == 0 required");
    MERV_CHECK(!(W2 && res), "merv_decode_gemv: the gated form takes no residual");
    MERV_CHECK(!norm_w || K <= 16384, "merv_decode_gemv: the fused RMSNorm takes K <= 16384 (call merv_decode_rmsnorm first for a wider input)");
    MERV_CHECK(((uintptr_t)W & 15) == 0 && ((uintptr_t)x & 15) == 0 && (!W2 || ((uintptr_t)W2 & 15) == 0), "merv_decode_gemv: 16-byte alignment required");
    DecodeGemvArgs a{};
    a.W = (const bf16_t*)W; a.W2 = (const bf16_t*)W2; a.x = (const bf16_t*)x; a.res = (const bf16_t*)res; a.y = (bf16_t*)y; a.y32 = y32;
    a.N = N; a.K = K; a.norm_w = (const bf16_t*)norm_w; a.norm_eps = norm_eps;
    MERV_HIP(launch_decode_gemv(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_gemv3(const void* Wa, const void* Wb, const void* Wc, const void* x, void* ya, void* yb, void* yc,
                                 int32_t Na, int32_t Nb, int32_t Nc, int32_t K, const void* norm_w, float norm_eps, void* stream_) {
    return merv_decode_gemv3_bias(Wa, Wb, Wc, x, ya, yb, yc, Na, Nb, Nc, K, norm_w, norm_eps, nullptr, nullptr, nullptr, stream_);
}

extern "C" int merv_decode_gemv3_bias(const void* Wa, const void* Wb, const void* Wc, const void* x, void* ya, void* yb, void* yc,
                                      int32_t Na, int32_t Nb, int32_t Nc, int32_t K, const void* norm_w, float norm_eps,
                                      const void* bias_a, const void* bias_b, const void* bias_c, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(Wa && Wb && Wc && x && ya && yb && yc, "merv_decode_gemv3: null argument");
    MERV_CHECK(!norm_w || K <= 16384, "merv_decode_gemv3: the fused RMSNorm takes K <= 16384");
    MERV_CHECK(Na > 0 && Nb > 0 && Nc > 0 && Na % 2 == 0 && Nb % 2 == 0 && Nc % 2 == 0 && K > 0 && K % 8 == 0,
               "merv_decode_gemv3: even row counts and K % 8 == 0 required");
    DecodeGemvArgs a{};
    a.W = (const bf16_t*)Wa; a.Wb = (const bf16_t*)Wb; a.Wc = (const bf16_t*)Wc; a.x = (const bf16_t*)x;
    a.y = (bf16_t*)ya; a.yb = (bf16_t*)yb; a.yc = (bf16_t*)yc; a.N = Na; a.Nb = Nb; a.Nc = Nc; a.K = K;
    a.norm_w = (const bf16_t*)norm_w; a.norm_eps = norm_eps;
    a.bias = (const bf16_t*)bias_a; a.bias_b = (const bf16_t*)bias_b; a.bias_c = (const bf16_t*)bias_c;
    MERV_HIP(launch_decode_gemv(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_rope_cache(const void* q, const void* k, const void* v, void* q_out, void* k_cache, void* v_cache,
                                      const void* cos_t, const void* sin_t, const int64_t* pos, int32_t H, int32_t Hkv, int32_t hd,
                                      int32_t max_len, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && q_out && k_cache && v_cache && cos_t && sin_t && pos, "merv_decode_rope_cache: null argument");
    MERV_CHECK(H > 0 && Hkv > 0 && hd > 0 && hd % 2 == 0 && max_len > 0, "merv_decode_rope_cache: bad geometry");
    DecodeRopeArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache,
                     (const bf16_t*)cos_t, (const bf16_t*)sin_t, (const long*)pos, H, Hkv, hd, max_len};
    MERV_HIP(launch_decode_rope_cache(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_prefill_rope_cache(void* q, const void* k, const void* v, void* k_cache, void* v_cache, const void* cos_t,
                                       const void* sin_t, int32_t S, int32_t pos0, int32_t H, int32_t Hkv, int32_t hd, int32_t max_len,
                                       int32_t ldq, int32_t ldk, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && k_cache && v_cache && cos_t && sin_t, "merv_prefill_rope_cache: null argument");
    MERV_CHECK(S > 0 && pos0 >= 0 && H > 0 && Hkv > 0 && hd > 0 && hd % 16 == 0 && max_len > 0 && pos0 + S <= max_len,
               "merv_prefill_rope_cache: bad geometry (hd % 16 == 0, pos0 + S <= max_len required)");
    MERV_CHECK(ldq >= H * hd && ldk >= Hkv * hd && ldq % 8 == 0 && ldk % 8 == 0, "merv_prefill_rope_cache: row strides must cover the rows, multiples of 8");
    MERV_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "merv_prefill_rope_cache: 16-byte alignment required");
    PrefillRopeArgs a{(bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)k_cache, (bf16_t*)v_cache, (const bf16_t*)cos_t,
                      (const bf16_t*)sin_t, S, pos0, H, Hkv, hd, max_len, ldq, ldk};
    MERV_HIP(launch_prefill_rope_cache(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_add_rmsnorm(void* x, const void* delta, const void* w, void* y, int32_t rows, int32_t D, float eps, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(x && delta && w && y, "merv_add_rmsnorm: null argument");
    MERV_CHECK(rows > 0 && D > 0 && D % 8 == 0 && D <= 8192, "merv_add_rmsnorm: rows > 0, D % 8 == 0 and D <= 8192 required");
    DecodeRmsArgs a{(const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, D, eps};
    MERV_HIP(launch_add_rmsnorm(a, (const bf16_t*)delta, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_prefill_attention(const void* q, const void* k, const void* v, void* out, int32_t S, int32_t H, int32_t Hkv,
                                      int32_t hd, int32_t ldq, int32_t ldk, int64_t kv_head_stride, int32_t ldo, float scale,
                                      void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && out, "merv_prefill_attention: null argument");
    MERV_CHECK(hd == 128, "merv_prefill_attention: head_dim must be 128");
    MERV_CHECK(S > 0 && H > 0 && Hkv > 0 && H % Hkv == 0, "merv_prefill_attention: bad geometry");
    MERV_CHECK(ldq >= H * hd && ldo >= H * hd && ldk >= hd && ldq % 8 == 0 && ldk % 8 == 0 && ldo % 8 == 0 && kv_head_stride % 8 == 0,
               "merv_prefill_attention: strides must cover the rows and be multiples of 8 elements");
    MERV_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "merv_prefill_attention: 16-byte alignment required");
    PrefillAttnArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, S, H, Hkv, ldq, ldk, ldo, (long)kv_head_stride,
                      scale, 0.f};
    MERV_HIP(launch_prefill_attention(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_silu_mul(const void* gate, const void* up, void* out, int64_t n, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(gate && up && out, "merv_silu_mul: null argument");
    MERV_CHECK(n > 0 && n % 8 == 0, "merv_silu_mul: n > 0 and n % 8 == 0 required");
    SiluMulArgs a{(const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, (long)n};
    MERV_HIP(launch_silu_mul(a, (hipStream_t)stream_));
    return 0;
}

extern "C" size_t merv_decode_attention_workspace_floats(int32_t H, int32_t nsplit) {
    if (H <= 0 || nsplit <= 0) return 0;
    return (size_t)H * nsplit * (128 + 2);
}

extern "C" int merv_decode_attention(const void* q, const void* k_cache, const void* v_cache, void* out, float* ws, const int64_t* pos,
                                     int32_t H, int32_t Hkv, int32_t hd, int32_t max_len, int32_t nsplit, float scale, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k_cache && v_cache && out && ws && pos, "merv_decode_attention: null argument");
    MERV_CHECK(hd == 128, "merv_decode_attention: head_dim must be 128 (Llama-2 / Mistral 7B and 13B)");
    MERV_CHECK(H > 0 && Hkv > 0 && H % Hkv == 0 && nsplit > 0 && nsplit <= 64 && max_len > 0, "merv_decode_attention: bad geometry");
    DecodeAttnArgs a{(const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, ws, (const long*)pos, H, Hkv, hd,
                     max_len, nsplit, scale};
    MERV_HIP(launch_decode_attention(a, (hipStream_t)stream_));
    return 0;
}

extern "C" size_t merv_decode_attention_fused_workspace_floats(int32_t H, int32_t nsplit) {
    if (H <= 0 || nsplit <= 0) return 0;
    return (size_t)H * nsplit * (128 + 2) + (size_t)H * 32;  // partials + one arrival counter per head, each on its own 128-byte line
}

extern "C" int merv_decode_attention_fused(const void* q, const void* k, const void* v, const void* cos_t, const void* sin_t,
                                           const int64_t* pos, void* k_cache, void* v_cache, void* out, float* ws, int32_t H, int32_t Hkv,
                                           int32_t hd, int32_t max_len, int32_t nsplit, float scale, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && cos_t && sin_t && pos && k_cache && v_cache && out && ws, "merv_decode_attention_fused: null argument");
    MERV_CHECK(hd == 128, "merv_decode_attention_fused: head_dim must be 128 (Llama-2 / Mistral 7B and 13B)");
    MERV_CHECK(H > 0 && Hkv > 0 && H % Hkv == 0 && nsplit > 0 && nsplit <= 64 && max_len > 0, "merv_decode_attention_fused: bad geometry");
    DecodeAttnFusedArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)k_cache,
                          (bf16_t*)v_cache, (bf16_t*)out, ws, (const long*)pos, H, Hkv, hd, max_len, nsplit, scale};
    MERV_HIP(launch_decode_attention_fused(a, (hipStream_t)stream_));
    return 0;
}

extern "C" size_t merv_decode_attention_split_workspace_floats(int32_t H, int32_t nsplit) {
    if (H <= 0 || nsplit <= 0) return 0;
    return decode_attention_split_workspace_floats(H, nsplit);
}

extern "C" int merv_decode_attention_split(const void* q, const void* k, const void* v, const void* cos_t, const void* sin_t,
                                           const int64_t* pos, void* k_cache, void* v_cache, float* ws, int32_t H, int32_t Hkv,
                                           int32_t hd, int32_t max_len, int32_t nsplit, float scale, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && cos_t && sin_t && pos && k_cache && v_cache && ws, "merv_decode_attention_split: null argument");
    MERV_CHECK(hd == 128, "merv_decode_attention_split: head_dim must be 128");
    MERV_CHECK(H > 0 && Hkv > 0 && H % Hkv == 0 && nsplit > 0 && nsplit <= 64 && max_len > 0, "merv_decode_attention_split: bad geometry");
    DecodeAttnFusedArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)k_cache,
                          (bf16_t*)v_cache, nullptr, ws, (const long*)pos, H, Hkv, hd, max_len, nsplit, scale};
    MERV_HIP(launch_decode_attention_split(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_attention_split_prefetch(const void* q, const void* k, const void* v, const void* cos_t, const void* sin_t,
                                                    const int64_t* pos, void* k_cache, void* v_cache, float* ws, int32_t H, int32_t Hkv,
                                                    int32_t hd, int32_t max_len, int32_t nsplit, float scale, const void* next_w,
                                                    int64_t next_w_bytes, int32_t next_block_bytes, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(q && k && v && cos_t && sin_t && pos && k_cache && v_cache && ws, "merv_decode_attention_split_prefetch: null argument");
    MERV_CHECK(hd == 128, "merv_decode_attention_split_prefetch: head_dim must be 128");
    MERV_CHECK(H > 0 && Hkv > 0 && H % Hkv == 0 && nsplit > 0 && nsplit <= 64 && max_len > 0, "merv_decode_attention_split_prefetch: bad geometry");
    MERV_CHECK(!next_w || (next_block_bytes > 0 && next_block_bytes % 16 == 0 && next_w_bytes > 0 && next_w_bytes / next_block_bytes <= (1 << 20) &&
                           ((uintptr_t)next_w & 15) == 0 && (H * nsplit) % 8 == 0),
               "merv_decode_attention_split_prefetch: next_w needs 16-byte alignment, a block size that is a multiple of 16 and H * nsplit % 8 == 0");
    DecodeAttnFusedArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)k_cache,
                          (bf16_t*)v_cache, nullptr, ws, (const long*)pos, H, Hkv, hd, max_len, nsplit, scale,
                          next_w, next_w ? next_block_bytes : 0, next_w ? (int)(next_w_bytes / next_block_bytes) : 0};
    MERV_HIP(launch_decode_attention_split(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_oproj_merge(const void* Wo, const void* res, void* y, const float* ws, void* attn_out, int32_t N, int32_t H,
                                       int32_t hd, int32_t nsplit, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(Wo && y && ws, "merv_decode_oproj_merge: null argument");
    MERV_CHECK(hd == 128 && H > 0 && H <= 256 && N > 0 && nsplit > 0 && nsplit <= 64, "merv_decode_oproj_merge: bad geometry (hd == 128, H <= 256)");
    MERV_CHECK(((uintptr_t)Wo & 15) == 0 && ((uintptr_t)ws & 15) == 0, "merv_decode_oproj_merge: 16-byte alignment required");
    DecodeOprojMergeArgs a{(const bf16_t*)Wo, (const bf16_t*)res, (bf16_t*)y, ws, (bf16_t*)attn_out, N, H, nsplit};
    MERV_HIP(launch_decode_oproj_merge(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_greedy_advance(const float* logits, int32_t V, int64_t* tok, int64_t* pos, int64_t* out_tokens, int64_t pos0,
                                          void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(logits && tok && pos, "merv_decode_greedy_advance: null argument");
    MERV_CHECK(V > 0, "merv_decode_greedy_advance: V > 0 required");
    static_assert(sizeof(long) == sizeof(int64_t), "int64 tokens");
    MERV_HIP(launch_decode_greedy_advance(logits, V, (long*)tok, (long*)pos, (long*)out_tokens, (long)pos0, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_decode_sample_advance(const float* logits, int32_t V, const void* params, int64_t* tok, int64_t* pos, int64_t* out_tokens,
                                          int64_t pos0, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(logits && params && tok && pos, "merv_decode_sample_advance: null argument");
    MERV_CHECK(V > 0, "merv_decode_sample_advance: V > 0 required");
    MERV_CHECK(((uintptr_t)params & 7) == 0, "merv_decode_sample_advance: params must be 8-byte aligned");
    MERV_HIP(launch_decode_sample_advance(logits, V, params, (long*)tok, (long*)pos, (long*)out_tokens, (long)pos0, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_mean_rows(const void* x, void* out, int32_t groups, int32_t rows, int32_t D, int64_t group_stride_rows,
                              void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(x && out, "merv_mean_rows: null argument");
    MERV_CHECK(groups >= 0 && rows > 0 && D > 0 && D % 8 == 0 && group_stride_rows >= 0, "merv_mean_rows: bad geometry");
    MeanRowsArgs a{(const bf16_t*)x, (bf16_t*)out, groups, rows, D, group_stride_rows};
    MERV_HIP(launch_mean_rows(a, (hipStream_t)stream_));
    return 0;
}

extern "C" int merv_map_pool_attention(const void* kv, const float* q, void* out, int32_t nseq, int32_t ntok, int32_t heads,
                                       float scale, void* stream_) {
    MERV_STREAM_DEVICE(stream_);
    MERV_CHECK(kv && q && out, "merv_map_pool_attention: null argument");
    MERV_CHECK(nseq >= 0 && ntok > 0 && ntok <= 1024 && heads > 0, "merv_map_pool_attention: bad geometry (head dim 64, at most 1024 tokens)");
    MapPoolArgs a{(const bf16_t*)kv, q, (bf16_t*)out, nseq, ntok, heads, scale};
    MERV_HIP(launch_map_pool(a, (hipStream_t)stream_));
    return 0;
}
