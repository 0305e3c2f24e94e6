// VGGT input preprocessing on the device: utils/model_utils.py:16-85 `preprocess_images_from_numpy`
//   frames uint8 [T, H, W, 3]  ->  float32 [T, 3, out_h, out_w] in [0, 1]
//   = PIL bicubic resize (:51) + ToTensor /255 (:52) + centre crop (:54-56, crop mode) or white pad to 518 x 518 (:58-71, pad mode).
// The resize is Pillow's 8-bit ImagingResample (src/libImaging/Resample.c, Pillow 12.2.0; restated in oracle/preprocess.py):
// two separable passes, bicubic a = -0.5 with support 2 max(scale, 1), double-precision weights normalised per output sample
// and fixed to 22 fractional bits, int32 accumulation from 1 << 21, clip of (acc >> 22) to [0, 255] after EACH pass.  Integer
// work end to end, so the result is bit-exact; the weights are made on the device in IEEE double with contraction off (Pillow is
// compiled without FMA) by pp_coeffs_kernel.
// Byte streaming, HBM-bound and tiny next to the scorer's other stages (10 frames of 720 x 1280: 27.6 MB in, 32 MB out).
#include "common.h"

#define PP_TARGET 518
#define PP_BITS 22

struct PpPlan {
    int out_w, out_h;       // resized image
    int fin_w, fin_h;       // returned image (after crop / pad)
    int crop_y;             // first resized row that is kept (crop mode)
    int pad_top, pad_left;  // pad mode
    int ks_x, ks_y;         // taps per output sample
};

static inline int pp_ksize(int in_size, int out_size) {
    double fs = (double)in_size / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(2.0 * fs) * 2 + 1;
}

// sizes of utils/model_utils.py:36-48; Python's round() is round-half-to-even = nearbyint in the default rounding mode
static int pp_plan(int H, int W, int mode, PpPlan* p) {
    if (H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return VGPA_ERR_INVALID;
    int nw, nh;
    if (mode == 1 && W < H) {
        nh = PP_TARGET;
        nw = (int)nearbyint((double)W * ((double)nh / (double)H) / 14.0) * 14;
    } else {
        nw = PP_TARGET;
        nh = (int)nearbyint((double)H * ((double)nw / (double)W) / 14.0) * 14;
    }
    if (nw <= 0 || nh <= 0) return VGPA_ERR_INVALID;   // PIL raises on an empty size
    p->out_w = nw; p->out_h = nh;
    p->crop_y = 0; p->pad_top = 0; p->pad_left = 0;
    p->fin_w = nw; p->fin_h = nh;
    if (mode == 0 && nh > PP_TARGET) { p->crop_y = (nh - PP_TARGET) / 2; p->fin_h = PP_TARGET; }
    if (mode == 1) {
        // F.pad with a negative amount would crop; it cannot happen: the longer side is 518 and the other is rounded from <= 518
        const int hp = PP_TARGET - nh, wp = PP_TARGET - nw;
        if (hp < 0 || wp < 0) return VGPA_ERR_INVALID;
        p->pad_top = hp / 2; p->pad_left = wp / 2;
        p->fin_h = PP_TARGET; p->fin_w = PP_TARGET;
    }
    p->ks_x = pp_ksize(W, nw);
    p->ks_y = pp_ksize(H, nh);
    return VGPA_OK;
}

__device__ __forceinline__ double pp_bicubic(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// one thread per output sample of one axis: bounds[xx] = {first input sample, tap count}, kk[xx][0 .. ksize)
__global__ __launch_bounds__(64) void pp_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * 64 + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pp_bicubic((x + xmin - center + 0.5) * ss);
    int* k = kk + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
        int q = 0;
        if (x < xmax) {
            double w = pp_bicubic((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            q = w < 0 ? (int)(-0.5 + w * (double)(1 << PP_BITS)) : (int)(0.5 + w * (double)(1 << PP_BITS));
        }
        k[x] = q;
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int pp_clip8(int acc) {
    const int v = acc >> PP_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[t][y][xx][c] (uint8, 3 channels interleaved); one thread per (t, y, xx)
__global__ __launch_bounds__(256) void pp_horizontal_kernel(const uint8_t* __restrict__ frames, int H, int W, int out_w, int ksize,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk, uint8_t* __restrict__ tmp) {
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= out_w) return;
    const size_t row = (size_t)blockIdx.z * H + blockIdx.y;
    const uint8_t* src = frames + row * (size_t)W * 3;
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < xmax; ++x) {
        const int w = k[x];
        const uint8_t* px = src + (size_t)(x + xmin) * 3;
        a0 += (int)px[0] * w; a1 += (int)px[1] * w; a2 += (int)px[2] * w;
    }
    uint8_t* o = tmp + (row * out_w + xx) * 3;
    o[0] = (uint8_t)pp_clip8(a0); o[1] = (uint8_t)pp_clip8(a1); o[2] = (uint8_t)pp_clip8(a2);
}

// vertical pass + crop / pad + /255, planar output: one thread per (t, final row, final column), all three channels
__global__ __launch_bounds__(256) void pp_vertical_kernel(const uint8_t* __restrict__ tmp /* [T][H][out_w][3] */, int H,
                                                            PpPlan p, const int* __restrict__ bounds, const int* __restrict__ kk, float* __restrict__ out) {
    const int fx = blockIdx.x * 256 + threadIdx.x;
    if (fx >= p.fin_w) return;
    const int fy = blockIdx.y, t = blockIdx.z;
    const int ox = fx - p.pad_left, oy = fy - p.pad_top + p.crop_y;
    int v0 = 255, v1 = 255, v2 = 255;   // white pad (value 1.0)
    if (ox >= 0 && ox < p.out_w && oy >= 0 && oy < p.out_h) {
        const int ymin = bounds[2 * oy], ymax = bounds[2 * oy + 1];
        const int* k = kk + (size_t)oy * p.ks_y;
        int a0 = 1 << (PP_BITS - 1), a1 = a0, a2 = a0;
        const uint8_t* col = tmp + (((size_t)t * H + ymin) * p.out_w + ox) * 3;
        for (int y = 0; y < ymax; ++y) {
            const int w = k[y];
            const uint8_t* px = col + (size_t)y * p.out_w * 3;
            a0 += (int)px[0] * w; a1 += (int)px[1] * w; a2 += (int)px[2] * w;
        }
        v0 = pp_clip8(a0); v1 = pp_clip8(a1); v2 = pp_clip8(a2);
    }
    const size_t plane = (size_t)p.fin_h * p.fin_w;
    float* o = out + (size_t)t * 3 * plane + (size_t)fy * p.fin_w + fx;
    o[0] = (float)v0 / 255.0f; o[plane] = (float)v1 / 255.0f; o[2 * plane] = (float)v2 / 255.0f;
}

static inline size_t pp_al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int32_t vgpa_preprocess_shape(int32_t H, int32_t W, int32_t mode, int32_t* out_h, int32_t* out_w) {
    PpPlan p;
    const int rc = pp_plan(H, W, mode, &p);
    if (rc != VGPA_OK) return rc;
    if (out_h) *out_h = p.fin_h;
    if (out_w) *out_w = p.fin_w;
    return VGPA_OK;
}

extern "C" size_t vgpa_preprocess_workspace_bytes(int32_t T, int32_t H, int32_t W, int32_t mode) {
    PpPlan p;
    if (T <= 0 || pp_plan(H, W, mode, &p) != VGPA_OK) return 0;
    return pp_al((size_t)p.out_w * (2 + p.ks_x) * 4) + pp_al((size_t)p.out_h * (2 + p.ks_y) * 4) + pp_al((size_t)T * H * p.out_w * 3);
}

extern "C" int32_t vgpa_preprocess_frames(const void* frames, int32_t T, int32_t H, int32_t W, int32_t mode, float* out, void* workspace,
                                          size_t ws_bytes, hipStream_t stream) {
    PpPlan p;
    if (!frames || !out || T <= 0) return VGPA_ERR_INVALID;
    const int rc = pp_plan(H, W, mode, &p);
    if (rc != VGPA_OK) return rc;
    if (T > 65535 || H > 65535 || p.fin_h > 65535) return VGPA_ERR_INVALID;
    if (!workspace || ws_bytes < vgpa_preprocess_workspace_bytes(T, H, W, mode)) return VGPA_ERR_WORKSPACE;
    uint8_t* ws = (uint8_t*)workspace;
    int* bx = (int*)ws;
    int* kx = bx + 2 * (size_t)p.out_w;
    ws += pp_al((size_t)p.out_w * (2 + p.ks_x) * 4);
    int* by = (int*)ws;
    int* ky = by + 2 * (size_t)p.out_h;
    ws += pp_al((size_t)p.out_h * (2 + p.ks_y) * 4);
    uint8_t* tmp = ws;
    VGPA_LAUNCH(pp_coeffs_kernel, dim3((p.out_w + 63) / 64), dim3(64), 0, stream, W, p.out_w, p.ks_x, bx, kx);
    VGPA_LAUNCH(pp_coeffs_kernel, dim3((p.out_h + 63) / 64), dim3(64), 0, stream, H, p.out_h, p.ks_y, by, ky);
    // Pillow skips a pass whose size does not change; an unchanged axis has the single weight 1.0, so running it is the identity too
    VGPA_LAUNCH(pp_horizontal_kernel, dim3((p.out_w + 255) / 256, H, T), dim3(256), 0, stream, (const uint8_t*)frames, H, W, p.out_w, p.ks_x, bx, kx, tmp);
    VGPA_LAUNCH(pp_vertical_kernel, dim3((p.fin_w + 255) / 256, p.fin_h, T), dim3(256), 0, stream, tmp, H, p, by, ky, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
