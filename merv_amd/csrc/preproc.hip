// GPU-side frame preprocessing: the per-encoder CPU transforms of the reference
// (64 Python PIL calls per video) as HIP kernels, uint8 frames [T,3,H,W] in HBM -> normalised pixel tensors.
//
//  * DINOv2 / SigLIP / ViViT (dinov2_video.py:76-124, siglip.py:86-134, vivit.py:50-92): torchvision
//    Resize((224,224)) on a PIL image == Pillow Image.resize: a separable convolution whose support grows with the
//    down-scale factor (antialias), 8-bit fixed-point coefficients (PRECISION_BITS = 22), horizontal pass then
//    vertical pass with an 8-bit intermediate image -- reproduced here stage by stage so the resized uint8 image is
//    BIT-EXACT with Pillow (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResample{Horizontal,
//    Vertical}_8bpc); then ToTensor (/255) and Normalize ((x - mean) / std), each a separate fp32 rounding.
//    Coefficients are computed on the device in float64 with explicitly unfused operations (no host tables, no copies:
//    the whole transform is stream-ordered and graph-capturable).
//  * LanguageBind (languagebind/video/processing_video.py:63-79): x/255 -> (x-mean)/std -> F.interpolate(bilinear,
//    align_corners=False, no antialias) to short side 224 -> centre crop 224 -> optional horizontal flip (the
//    reference flips at random at inference time, SURVEY Appendix B.2: here it is an explicit, default-off switch).
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace merv {
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// float64 helpers with one rounding per operation (hipcc would otherwise contract a*b+c into an fma)
MERV_DEVICE double dmul(double a, double b) { return __dmul_rn(a, b); }
MERV_DEVICE double dadd(double a, double b) { return __dadd_rn(a, b); }
MERV_DEVICE double dsub(double a, double b) { return __dsub_rn(a, b); }

MERV_DEVICE double pil_filter(int kind, double x) {
    if (x < 0.0) x = -x;
    if (kind == 0) {  // bilinear: triangle
        return x < 1.0 ? dsub(1.0, x) : 0.0;
    }
    // bicubic, a = -0.5 (Pillow): ((a+2)x - (a+3)) x x + 1   |   (((x-5)x + 8)x - 4) a
    const double a = -0.5;
    if (x < 1.0) return dadd(dmul(dmul(dsub(dmul(a + 2.0, x), a + 3.0), x), x), 1.0);
    if (x < 2.0) return dmul(dsub(dmul(dadd(dmul(dsub(x, 5.0), x), 8.0), x), 4.0), a);
    return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc for one axis: one thread per output index
__global__ void pil_coeffs_kernel(int inSize, int outSize, int kind, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= outSize) return;
    const float in0 = 0.f, in1 = (float)inSize;
    double scale = (double)(in1 - in0) / (double)outSize;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = dmul(kind == 0 ? 1.0 : 2.0, filterscale);
    const double center = dadd((double)in0, dmul(dadd((double)xx, 0.5), scale));
    const double ss = 1.0 / filterscale;
    int xmin = (int)dadd(dsub(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)dadd(dadd(center, support), 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double ww = 0.0;
    int* k = kk + (size_t)xx * ksize;
    // first pass: weights (kept in registers via recomputation: ksize <= ~30), sum in tap order
    for (int x = 0; x < xmax; ++x) ww = dadd(ww, pil_filter(kind, dmul(dadd(dsub((double)(x + xmin), center), 0.5), ss)));
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = pil_filter(kind, dmul(dadd(dsub((double)(x + xmin), center), 0.5), ss));
            if (ww != 0.0) w = w / ww;
        }
        const double sc = dmul(w, (double)(1 << PRECISION_BITS));
        k[x] = w < 0 ? (int)dadd(-0.5, sc) : (int)dadd(0.5, sc);
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

MERV_DEVICE uint8_t clip8(int v) {
    v >>= PRECISION_BITS;  // arithmetic shift, like Pillow's clip8 lookup index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [planes, H, W] -> dst [planes, H, outW]
__global__ __launch_bounds__(256) void pil_resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long planes_h,
                                                            int W, int outW, int ksize, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk) {
    const long long total = planes_h * outW;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int xx = (int)(g % outW);
        const long long row = g / outW;
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (size_t)xx * ksize;
        const uint8_t* s = src + row * W + xmin;
        int ss0 = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < xmax; ++x) ss0 += (int)s[x] * k[x];
        dst[g] = clip8(ss0);
    }
}

// vertical pass: src [planes, H, W] -> dst [planes, outH, W]
__global__ __launch_bounds__(256) void pil_resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int planes, int H,
                                                            int W, int outH, int ksize, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk) {
    const long long total = (long long)planes * outH * W;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int x = (int)(g % W);
        const long long t = g / W;
        const int yy = (int)(t % outH);
        const long long plane = t / outH;
        const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
        const int* k = kk + (size_t)yy * ksize;
        const uint8_t* s = src + (plane * H + ymin) * W + x;
        int ss0 = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < ymax; ++y) ss0 += (int)s[(size_t)y * W] * k[y];
        dst[g] = clip8(ss0);
    }
}

// ToTensor + Normalize: out = (u / 255 - mean[c]) / std[c], three fp32 roundings; planes are (frame, channel) pairs
template <bool BF16_OUT>
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src, void* __restrict__ dst, long long total, int plane_px,
                                                          float m0, float m1, float m2, float s0, float s1, float s2) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int c = (int)((g / plane_px) % 3);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const float v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)src[g], 255.0f), mean), sd);
        if constexpr (BF16_OUT) ((bf16_t*)dst)[g] = f2bf(v);
        else ((float*)dst)[g] = v;
    }
}

// LanguageBind: normalise, bilinear (align_corners = False) to (newH, newW), crop at (top, left), optional flip.
// src [T,3,H,W] u8 -> dst [3,T,S,S]
template <bool BF16_OUT>
__global__ __launch_bounds__(256) void languagebind_kernel(const uint8_t* __restrict__ src, void* __restrict__ dst, int T, int H, int W, int newH,
                                                          int newW, int top, int left, int S, int flip, float m0, float m1, float m2,
                                                          float s0, float s1, float s2) {
    const long long total = 3LL * T * S * S;
    const float sh = (float)H / (float)newH, sw = (float)W / (float)newW;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        int x = (int)(g % S);
        long long r = g / S;
        const int y = (int)(r % S); r /= S;
        const int t = (int)(r % T);
        const int c = (int)(r / T);
        if (flip) x = S - 1 - x;
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        // area_pixel_compute_source_index (ATen UpSample.h), align_corners = false
        float fy = __fsub_rn(__fmul_rn(sh, (float)(y + top) + 0.5f), 0.5f);
        float fx = __fsub_rn(__fmul_rn(sw, (float)(x + left) + 0.5f), 0.5f);
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
        const float ly1 = __fsub_rn(fy, (float)y0), lx1 = __fsub_rn(fx, (float)x0);
        const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
        const uint8_t* p = src + ((size_t)(t * 3 + c) * H + y0) * W + x0;
        auto px = [&](int dy, int dx) {
            return __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[(size_t)dy * W + dx], 255.0f), mean), sd);
        };
        const float top_row = __fadd_rn(__fmul_rn(lx0, px(0, 0)), __fmul_rn(lx1, px(0, xp)));
        const float bot_row = __fadd_rn(__fmul_rn(lx0, px(yp, 0)), __fmul_rn(lx1, px(yp, xp)));
        const float v = __fadd_rn(__fmul_rn(ly0, top_row), __fmul_rn(ly1, bot_row));
        const long long o = ((long long)(c * T + t) * S + y) * S + (flip ? S - 1 - x : x);
        if constexpr (BF16_OUT) ((bf16_t*)dst)[o] = f2bf(v);
        else ((float*)dst)[o] = v;
    }
}

inline int grid_for(long long items) {
    long long g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

int pil_ksize(int inSize, int outSize, int kind) {
    double filterscale = (double)((float)inSize) / (double)outSize;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (kind == 0 ? 1.0 : 2.0) * filterscale;
    return (int)ceil(support) * 2 + 1;
}

size_t pil_workspace_bytes(int T, int H, int W, int out) {
    const size_t coef = (size_t)2 * out * (2 + (size_t)(pil_ksize(H > W ? H : W, out, 1))) * sizeof(int);  // both axes, bounds + kk
    const size_t tmp = (size_t)T * 3 * H * out;        // horizontally resampled image
    const size_t res = (size_t)T * 3 * out * out;      // resized uint8 image
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return up(coef) + up(tmp) + up(res);
}

// Resize((out,out)) with Pillow semantics. resized_u8 (optional) receives the uint8 image [T,3,out,out]; out_pix the
// normalised tensor [T,3,out,out] (fp32 or bf16).
hipError_t launch_pil_resize_normalize(const uint8_t* frames, int T, int H, int W, int out, int kind, const float* mean,
                                       const float* sd, void* out_pix, int out_bf16, uint8_t* resized_u8_opt, char* ws, hipStream_t s) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const int ks_h = pil_ksize(W, out, kind), ks_v = pil_ksize(H, out, kind);
    int* bounds_h = (int*)ws;
    int* kk_h = bounds_h + 2 * out;
    int* bounds_v = kk_h + (size_t)out * ks_h;
    int* kk_v = bounds_v + 2 * out;
    const size_t coef_bytes = (size_t)2 * out * (2 + (size_t)pil_ksize(H > W ? H : W, out, 1)) * sizeof(int);
    uint8_t* tmp = (uint8_t*)(ws + up(coef_bytes));
    uint8_t* res = tmp + up((size_t)T * 3 * H * out);
    const bool need_h = W != out, need_v = H != out;
    const uint8_t* cur = frames;
    int curW = W;
    if (need_h) {
        hipLaunchKernelGGL(pil_coeffs_kernel, dim3((out + 63) / 64), dim3(64), 0, s, W, out, kind, ks_h, bounds_h, kk_h);
        uint8_t* dst = need_v ? tmp : res;
        const long long planes_h = (long long)T * 3 * H;
        hipLaunchKernelGGL(pil_resample_h_kernel, dim3(grid_for(planes_h * out)), dim3(256), 0, s, cur, dst, planes_h, W, out, ks_h,
                           bounds_h, kk_h);
        cur = dst;
        curW = out;
    }
    if (need_v) {
        hipLaunchKernelGGL(pil_coeffs_kernel, dim3((out + 63) / 64), dim3(64), 0, s, H, out, kind, ks_v, bounds_v, kk_v);
        hipLaunchKernelGGL(pil_resample_v_kernel, dim3(grid_for((long long)T * 3 * out * curW)), dim3(256), 0, s, cur, res, T * 3, H, curW,
                           out, ks_v, bounds_v, kk_v);
        cur = res;
    }
    const long long total = (long long)T * 3 * out * out;
    if (resized_u8_opt && cur != resized_u8_opt) {
        hipError_t e = hipMemcpyAsync(resized_u8_opt, cur, (size_t)total, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    if (out_pix) {
        if (out_bf16)
            hipLaunchKernelGGL(normalize_u8_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, cur, out_pix, total, out * out, mean[0],
                               mean[1], mean[2], sd[0], sd[1], sd[2]);
        else
            hipLaunchKernelGGL(normalize_u8_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, cur, out_pix, total, out * out, mean[0],
                               mean[1], mean[2], sd[0], sd[1], sd[2]);
    }
    return hipGetLastError();
}

hipError_t launch_languagebind_transform(const uint8_t* frames, int T, int H, int W, int S, int flip, const float* mean, const float* sd,
                                         void* out_pix, int out_bf16, hipStream_t s) {
    // ShortSideScale(224) (processing_video.py:52-58) and CenterCropVideo(224) (torchvision _functional_video.center_crop)
    int newH, newW;
    if (W < H) { newH = (int)floor(((double)((float)H) / W) * S); newW = S; }
    else { newH = S; newW = (int)floor(((double)((float)W) / H) * S); }
    const int top = (int)nearbyint((newH - S) / 2.0), left = (int)nearbyint((newW - S) / 2.0);  // Python round(): half to even
    const long long total = 3LL * T * S * S;
    if (out_bf16)
        hipLaunchKernelGGL(languagebind_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, frames, out_pix, T, H, W, newH, newW, top,
                           left, S, flip, mean[0], mean[1], mean[2], sd[0], sd[1], sd[2]);
    else
        hipLaunchKernelGGL(languagebind_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, frames, out_pix, T, H, W, newH, newW, top,
                           left, S, flip, mean[0], mean[1], mean[2], sd[0], sd[1], sd[2]);
    return hipGetLastError();
}

}  // namespace merv
