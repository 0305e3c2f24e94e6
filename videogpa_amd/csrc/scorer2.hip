// Geometry-consistency scorer, second file: the pieces around the z-buffer splat (scorer.hip) that the reference's
// VideoProcessor runs per video (pipelines/process_video.py:66-196):
//   * confidence cut of get_colored_pointcloud (utils/pointcloud_utils.py:44-73): the k-th largest valid confidence by an
//     on-device 4 x 8-bit radix select -- no sort, no torch.topk, no host round trip; the splat kernel reads the threshold
//     from device memory,
//   * MSE with the bilinear-resize branch and PSNR (metrics/mse.py:24-25,56-80),
//   * MVCS, the multi-view depth-consistency score (metrics/mvcs.py:12-114),
//   * DA3 unprojection (depth_anything_3/utils/geometry.py:54-59,434-497 as pipelines/process_video.py:151-156 uses it),
//   * VGGT pose-encoding decoder (vggt/utils/pose_enc.py:62-124, vggt/utils/rotation.py:14-44).
// All HBM-bound single passes or latency-sized.  Oracle: oracle/scorer.py (pinned to the reference through
// tests/golden/scorer2.pt).
#include "common.h"

#pragma clang fp contract(off)

#define S2_THREADS 256

static inline unsigned s2_grid(int64_t n, int cap) {
    int64_t nb = (n + S2_THREADS - 1) / S2_THREADS;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

// ---------------------------------------------------------------------------------------------- k-th largest confidence
// Valid confidences are finite and > 1e-5, i.e. positive floats: their raw bit patterns order like the values.
struct SelState {
    uint32_t hist[4][256];
    uint32_t prefix;      // bits fixed so far (high to low)
    uint32_t k_rem;       // rank still to find inside the prefix bucket (1-based, from the top)
    uint32_t n_valid;
    uint32_t done;        // n_valid == 0 -> threshold = -inf
};

__device__ __forceinline__ bool conf_valid(float c) { return isfinite(c) && c > 1e-5f; }

template <int PASS>
__global__ __launch_bounds__(S2_THREADS) void select_hist_kernel(const float* __restrict__ conf, int64_t N, SelState* __restrict__ st) {
    __shared__ uint32_t h[256];
    __shared__ uint32_t nv;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) nv = 0;
    __syncthreads();
    const uint32_t prefix = PASS ? st->prefix : 0u;
    if (PASS && st->done) return;
    constexpr int shift = 24 - 8 * PASS;
    constexpr uint32_t himask = PASS ? (0xFFFFFFFFu << (shift + 8 > 31 ? 31 : shift + 8)) : 0u;   // bits fixed by earlier passes
    uint32_t cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * S2_THREADS) {
        const float c = conf[i];
        if (!conf_valid(c)) continue;
        const uint32_t b = __float_as_uint(c);
        if (PASS == 0) {
            ++cnt;
            atomicAdd(&h[b >> 24], 1u);
        } else if (((b ^ prefix) & himask) == 0u) {
            atomicAdd(&h[(b >> shift) & 255u], 1u);
        }
    }
    if (PASS == 0) {
        cnt = (uint32_t)wave_sum((int)cnt);
        if ((threadIdx.x & 63) == 0) atomicAdd(&nv, cnt);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[PASS][threadIdx.x], h[threadIdx.x]);
    if (PASS == 0 && threadIdx.x == 0 && nv) atomicAdd(&st->n_valid, nv);
}

template <int PASS>
__global__ void select_pick_kernel(SelState* __restrict__ st, double keep_frac, float* __restrict__ thr_out) {
    if (threadIdx.x != 0) return;
    if (PASS == 0) {
        const uint32_t n = st->n_valid;
        if (n == 0) { st->done = 1; thr_out[0] = -INFINITY; return; }
        // k = max(1, int(ceil(N * keep_frac))) in double, as the reference computes it (utils/pointcloud_utils.py:60-61)
        double kk = ceil((double)n * keep_frac);
        uint32_t k = kk < 1.0 ? 1u : (kk > (double)n ? n : (uint32_t)kk);
        st->k_rem = k;
        st->prefix = 0;
    }
    if (st->done) return;
    constexpr int shift = 24 - 8 * PASS;
    uint32_t k = st->k_rem, cum = 0;
    int b = 255;
    for (; b > 0; --b) {
        const uint32_t c = st->hist[PASS][b];
        if (cum + c >= k) break;
        cum += c;
    }
    st->k_rem = k - cum;
    st->prefix |= (uint32_t)b << shift;
    if (PASS == 3) thr_out[0] = __uint_as_float(st->prefix);
}

// ---------------------------------------------------------------------------------------------- MSE / PSNR with optional bilinear resize of rep
__device__ __forceinline__ uint32_t f32_ordered2(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered2(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ __launch_bounds__(S2_THREADS) void minmax1_kernel(const void* __restrict__ a, int dtype, int64_t n, uint32_t* __restrict__ mm /* [2]: ~min, max */) {
    uint32_t lo = 0, hi = 0;
    for (int64_t i = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * S2_THREADS) {
        const float x = dtype == 2 ? (float)reinterpret_cast<const uint8_t*>(a)[i] : reinterpret_cast<const float*>(a)[i];
        const uint32_t o = f32_ordered2(x);
        hi = max(hi, o);
        lo = max(lo, ~o);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
        lo = max(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&mm[0], lo); atomicMax(&mm[1], hi); }
}

__device__ __forceinline__ float to01b(float x, float mn, float mx, int is_tensor) {   // metrics/mse.py:31-54
    if (is_tensor && mn < 0.f) return (x + 1.0f) / 2.0f;
    if (mx > 1.0f) return x / 255.0f;
    return x;
}
__device__ __forceinline__ float img_at2(const void* p, int dtype, int layout, int64_t t, int c, int y, int x, int C, int H, int W) {
    const size_t i = layout ? (((size_t)(t * H + y) * W + x) * C + c) : (((size_t)(t * C + c) * H + y) * W + x);
    return dtype == 2 ? (float)reinterpret_cast<const uint8_t*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}

// F.interpolate(mode='bilinear', align_corners=False) source coordinate: (dst + 0.5) * (in / out) - 0.5 clamped at 0
__device__ __forceinline__ void lin_tap(int dst, int n_in, float scale, int& i0, int& i1, float& w0, float& w1) {
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = fmaxf(src, 0.f);
    i0 = min((int)floorf(src), n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    w1 = src - (float)i0;
    w0 = 1.0f - w1;
}

__global__ __launch_bounds__(S2_THREADS) void mse_resize_kernel(const void* __restrict__ gt, int gt_dtype, int gt_layout, int gt_is_tensor,
                                                                  const void* __restrict__ rep, int rep_dtype, int rep_layout, int rep_is_tensor,
                                                                  int64_t T, int C, int H, int W, int H2, int W2, const uint32_t* __restrict__ mm,
                                                                  double* __restrict__ partial) {
    __shared__ double smem[16];
    const float amin = f32_unordered2(~mm[0]), amax = f32_unordered2(mm[1]), bmin = f32_unordered2(~mm[2]), bmax = f32_unordered2(mm[3]);
    const int64_t n = T * C * (int64_t)H * W;
    const bool same = (H == H2) && (W == W2);
    const float sy = (float)H2 / (float)H, sx = (float)W2 / (float)W;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * S2_THREADS) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t t = i / ((int64_t)W * H * C);
        const float g = to01b(img_at2(gt, gt_dtype, gt_layout, t, c, y, x, C, H, W), amin, amax, gt_is_tensor);
        float r;
        if (same) {
            r = to01b(img_at2(rep, rep_dtype, rep_layout, t, c, y, x, C, H, W), bmin, bmax, rep_is_tensor);
        } else {
            int y0, y1, x0, x1;
            float wy0, wy1, wx0, wx1;
            lin_tap(y, H2, sy, y0, y1, wy0, wy1);
            lin_tap(x, W2, sx, x0, x1, wx0, wx1);
            const float a00 = to01b(img_at2(rep, rep_dtype, rep_layout, t, c, y0, x0, C, H2, W2), bmin, bmax, rep_is_tensor);
            const float a01 = to01b(img_at2(rep, rep_dtype, rep_layout, t, c, y0, x1, C, H2, W2), bmin, bmax, rep_is_tensor);
            const float a10 = to01b(img_at2(rep, rep_dtype, rep_layout, t, c, y1, x0, C, H2, W2), bmin, bmax, rep_is_tensor);
            const float a11 = to01b(img_at2(rep, rep_dtype, rep_layout, t, c, y1, x1, C, H2, W2), bmin, bmax, rep_is_tensor);
            r = (a00 * wx0 + a01 * wx1) * wy0 + (a10 * wx0 + a11 * wx1) * wy1;
        }
        const float d = g - r;
        acc += d * d;
    }
    const double s = block_sum<double>((double)acc, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// LPIPSMetric._to_tensor_neg1_pos1 (metrics/lpips.py:38-63) + the resize of :31-32: any frame container -> fp32 [T,C,Ho,Wo] in [-1,1].
//   torch.Tensor: min >= 0 -> (max > 1 ? x/255 : x) * 2 - 1, else unchanged;  numpy: always (x/255) * 2 - 1.
__device__ __forceinline__ float to_pm1(float x, float mn, float mx, int is_tensor) {
    if (!is_tensor) return (x / 255.0f) * 2.0f - 1.0f;
    if (mn >= 0.f) {
        if (mx > 1.0f) x = x / 255.0f;
        return x * 2.0f - 1.0f;
    }
    return x;
}
__global__ __launch_bounds__(S2_THREADS) void frames_pm1_kernel(const void* __restrict__ src, int dtype, int layout, int is_tensor, int64_t T, int C,
                                                                  int H, int W, int Ho, int Wo, const uint32_t* __restrict__ mm, float* __restrict__ out) {
    const float mn = f32_unordered2(~mm[0]), mx = f32_unordered2(mm[1]);
    const int64_t n = T * C * (int64_t)Ho * Wo;
    const bool same = (H == Ho) && (W == Wo);
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    for (int64_t i = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * S2_THREADS) {
        const int x = (int)(i % Wo);
        const int y = (int)((i / Wo) % Ho);
        const int c = (int)((i / ((int64_t)Wo * Ho)) % C);
        const int64_t t = i / ((int64_t)Wo * Ho * C);
        float r;
        if (same) {
            r = to_pm1(img_at2(src, dtype, layout, t, c, y, x, C, H, W), mn, mx, is_tensor);
        } else {
            int y0, y1, x0, x1;
            float wy0, wy1, wx0, wx1;
            lin_tap(y, H, sy, y0, y1, wy0, wy1);
            lin_tap(x, W, sx, x0, x1, wx0, wx1);
            const float a00 = to_pm1(img_at2(src, dtype, layout, t, c, y0, x0, C, H, W), mn, mx, is_tensor);
            const float a01 = to_pm1(img_at2(src, dtype, layout, t, c, y0, x1, C, H, W), mn, mx, is_tensor);
            const float a10 = to_pm1(img_at2(src, dtype, layout, t, c, y1, x0, C, H, W), mn, mx, is_tensor);
            const float a11 = to_pm1(img_at2(src, dtype, layout, t, c, y1, x1, C, H, W), mn, mx, is_tensor);
            r = (a00 * wx0 + a01 * wx1) * wy0 + (a10 * wx0 + a11 * wx1) * wy1;
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(S2_THREADS) void mse_finish_kernel(const double* __restrict__ partial, int nblk, double inv_n, int psnr, float* __restrict__ out) {
    __shared__ double smem[16];
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += S2_THREADS) s += partial[i];
    s = block_sum<double>(s, smem);
    if (threadIdx.x == 0) {
        const float m = (float)(s * inv_n);
        if (!psnr) out[0] = m;
        else out[0] = (m == 0.f) ? 100.0f : 10.0f * log10f(1.0f / m);       // metrics/mse.py:69-74
    }
}

// ---------------------------------------------------------------------------------------------- small fp64 matrix helpers (per-view set-up)
__device__ void inv3(const double* a, double* o) {   // adjugate / determinant
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// ---------------------------------------------------------------------------------------------- MVCS
// One launch: grid (blocks, T-1).  Pair i -> j = i + 1: back-project every pixel of view i with its depth, move it into
// camera j, project, sample depth_j bilinearly there (grid_sample align_corners=True, zeros padding), compare with the
// reprojected z.  fp32 per-pixel math in the reference's order; the 3x3 inverse / relative pose are formed in fp64 and
// rounded to fp32 once (the reference uses fp32 LU there; 1e-5 relative tolerance, tests/test_gpu_scorer.py).
__global__ __launch_bounds__(S2_THREADS) void mvcs_kernel(const float* __restrict__ depth, const float* __restrict__ Kmat, int k_stride,
                                                            int k_row, const float* __restrict__ Emat, int e_stride, int H, int W,
                                                            double* __restrict__ pair_sum, unsigned long long* __restrict__ pair_cnt) {
    __shared__ double smem[16];
    __shared__ float sK[9], sKi[9], sR[9], sT[3];
    const int i = blockIdx.y, j = i + 1;
    if (threadIdx.x == 0) {
        double Ki[9], Kinv[9], Ei[16], Ej[16];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                Ki[r * 3 + c] = Kmat[(size_t)i * k_stride + r * k_row + c];
                sK[r * 3 + c] = Kmat[(size_t)j * k_stride + r * k_row + c];
            }
        inv3(Ki, Kinv);
        for (int r = 0; r < 9; ++r) sKi[r] = (float)Kinv[r];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) { Ei[r * 4 + c] = Emat[(size_t)i * e_stride + r * 4 + c]; Ej[r * 4 + c] = Emat[(size_t)j * e_stride + r * 4 + c]; }
        // inverse of [A|b; 0 0 0 1] = [A^-1 | -A^-1 b]
        double A[9], Ai[9], bi[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) A[r * 3 + c] = Ei[r * 4 + c];
        inv3(A, Ai);
        for (int r = 0; r < 3; ++r) bi[r] = -(Ai[r * 3] * Ei[3] + Ai[r * 3 + 1] * Ei[7] + Ai[r * 3 + 2] * Ei[11]);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) sR[r * 3 + c] = (float)(Ej[r * 4] * Ai[c] + Ej[r * 4 + 1] * Ai[3 + c] + Ej[r * 4 + 2] * Ai[6 + c]);
            sT[r] = (float)(Ej[r * 4] * bi[0] + Ej[r * 4 + 1] * bi[1] + Ej[r * 4 + 2] * bi[2] + Ej[r * 4 + 3]);
        }
    }
    __syncthreads();
    const float* di = depth + (size_t)i * H * W;
    const float* dj = depth + (size_t)j * H * W;
    const int64_t HW = (int64_t)H * W;
    double acc = 0.0;
    unsigned long long cnt = 0;
    for (int64_t p = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; p < HW; p += (int64_t)gridDim.x * S2_THREADS) {
        const float x = (float)(p % W), y = (float)(p / W);
        const float d = di[p];
        float pi[3], pj[3], hj[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pi[r] = ((sKi[r * 3] * x + sKi[r * 3 + 1] * y) + sKi[r * 3 + 2]) * d;
#pragma unroll
        for (int r = 0; r < 3; ++r) pj[r] = ((sR[r * 3] * pi[0] + sR[r * 3 + 1] * pi[1]) + sR[r * 3 + 2] * pi[2]) + sT[r];
#pragma unroll
        for (int r = 0; r < 3; ++r) hj[r] = (sK[r * 3] * pj[0] + sK[r * 3 + 1] * pj[1]) + sK[r * 3 + 2] * pj[2];
        const float zc = fmaxf(hj[2], 1e-8f);
        const float u = hj[0] / zc, v = hj[1] / zc;
        if (!(u >= 0.f && u < (float)W && v >= 0.f && v < (float)H && pj[2] > 0.f)) continue;
        // grid_sample(align_corners=True): normalise, un-normalise (kept: it rounds), 4 taps with zeros outside
        const float gx = 2.0f * u / (float)(W - 1) - 1.0f, gy = 2.0f * v / (float)(H - 1) - 1.0f;
        const float ix = (gx + 1.0f) / 2.0f * (float)(W - 1), iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
        const float x0 = floorf(ix), y0 = floorf(iy);
        const float wx1 = ix - x0, wy1 = iy - y0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        auto tap = [&](float xx, float yy) -> float {
            if (!(xx >= 0.f && xx <= (float)(W - 1) && yy >= 0.f && yy <= (float)(H - 1))) return 0.f;
            return dj[(size_t)yy * W + (size_t)xx];
        };
        const float s = tap(x0, y0) * (wx0 * wy0) + tap(x0 + 1.f, y0) * (wx1 * wy0) + tap(x0, y0 + 1.f) * (wx0 * wy1) +
                        tap(x0 + 1.f, y0 + 1.f) * (wx1 * wy1);
        const float e = s - pj[2];
        acc += (double)(e * e);
        ++cnt;
    }
    acc = block_sum<double>(acc, smem);
    const double c = block_sum<double>((double)cnt, smem);
    if (threadIdx.x == 0 && c > 0) {
        atomicAdd(&pair_sum[i], acc);
        atomicAdd(&pair_cnt[i], (unsigned long long)c);
    }
}

__global__ void mvcs_finish_kernel(const double* __restrict__ pair_sum, const unsigned long long* __restrict__ pair_cnt, int n_pairs,
                                   float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0;
    int n = 0;
    for (int i = 0; i < n_pairs; ++i)
        if (pair_cnt[i]) { s += (double)(float)(pair_sum[i] / (double)pair_cnt[i]); ++n; }      // err.item() of an fp32 mean, per pair
    out[0] = n ? (float)exp(-1.0 * (s / n)) : 0.f;                                              // metrics/mvcs.py:106-114
}

// ---------------------------------------------------------------------------------------------- DA3 unprojection
__global__ __launch_bounds__(S2_THREADS) void unproject_kernel(const float* __restrict__ depth, const float* __restrict__ Kmat,
                                                                 const float* __restrict__ Emat, int e_stride, int H, int W,
                                                                 float* __restrict__ world) {
    __shared__ float sKi[9], sR[9], sT[3];
    const int t = blockIdx.y;
    if (threadIdx.x == 0) {
        double Kd[9], Kinv[9];
        for (int r = 0; r < 9; ++r) Kd[r] = Kmat[(size_t)t * 9 + r];
        inv3(Kd, Kinv);
        for (int r = 0; r < 9; ++r) sKi[r] = (float)Kinv[r];
        // c2w = affine_inverse(w2c) = [R^T | -R^T t]  (depth_anything_3/utils/geometry.py:54-59)
        const float* E = Emat + (size_t)t * e_stride;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) sR[r * 3 + c] = E[c * 4 + r];
            sT[r] = -((E[0 * 4 + r] * E[3] + E[1 * 4 + r] * E[7]) + E[2 * 4 + r] * E[11]);
        }
    }
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * S2_THREADS + threadIdx.x; p < HW; p += (int64_t)gridDim.x * S2_THREADS) {
        const float x = (float)(p % W), y = (float)(p / W);
        const float d = depth[(size_t)t * HW + p];
        float cam[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) cam[r] = ((sKi[r * 3] * x + sKi[r * 3 + 1] * y) + sKi[r * 3 + 2]) * d;
        float* o = world + ((size_t)t * HW + p) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = ((sR[r * 3] * cam[0] + sR[r * 3 + 1] * cam[1]) + sR[r * 3 + 2] * cam[2]) + sT[r];
    }
}

// ---------------------------------------------------------------------------------------------- VGGT pose encoding -> [R|t], K
__global__ void pose_decode_kernel(const float* __restrict__ pe, int64_t n, float img_h, float img_w, float* __restrict__ ext,
                                   float* __restrict__ intr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pe + i * 9;
    const float qi = p[3], qj = p[4], qk = p[5], qr = p[6];
    const float two_s = 2.0f / (((qi * qi + qj * qj) + qk * qk) + qr * qr);
    float R[9];
    R[0] = 1.f - two_s * (qj * qj + qk * qk); R[1] = two_s * (qi * qj - qk * qr); R[2] = two_s * (qi * qk + qj * qr);
    R[3] = two_s * (qi * qj + qk * qr); R[4] = 1.f - two_s * (qi * qi + qk * qk); R[5] = two_s * (qj * qk - qi * qr);
    R[6] = two_s * (qi * qk - qj * qr); R[7] = two_s * (qj * qk + qi * qr); R[8] = 1.f - two_s * (qi * qi + qj * qj);
    float* e = ext + i * 12;
    for (int r = 0; r < 3; ++r) { e[r * 4] = R[r * 3]; e[r * 4 + 1] = R[r * 3 + 1]; e[r * 4 + 2] = R[r * 3 + 2]; e[r * 4 + 3] = p[r]; }
    if (intr) {
        float* k = intr + i * 9;
        for (int r = 0; r < 9; ++r) k[r] = 0.f;
        k[4] = (img_h / 2.0f) / tanf(p[7] / 2.0f);
        k[0] = (img_w / 2.0f) / tanf(p[8] / 2.0f);
        k[2] = img_w / 2.0f;
        k[5] = img_h / 2.0f;
        k[8] = 1.0f;
    }
}

extern "C" {

size_t vgpa_conf_threshold_workspace_bytes(void) { return sizeof(SelState); }

// thr_out[0] (device fp32) = the k-th largest valid confidence, k = max(1, ceil(n_valid * (1 - conf_thres/100))): points with
// conf >= thr (and valid) are the top (100 - conf_thres) % that get_colored_pointcloud keeps (utils/pointcloud_utils.py:55-73).
// conf_thres <= 0 or no valid point: -inf.  conf fp32 [N].
int32_t vgpa_conf_threshold(const float* conf, int64_t N, float conf_thres, float* thr_out, void* workspace, size_t ws_bytes,
                            hipStream_t stream) {
    if (!conf || !thr_out || !workspace || N <= 0) return VGPA_ERR_INVALID;
    if (ws_bytes < sizeof(SelState)) return VGPA_ERR_WORKSPACE;
    SelState* st = (SelState*)workspace;
    if (hipMemsetAsync(st, 0, sizeof(SelState), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    double keep = 1.0 - (double)conf_thres / 100.0;
    keep = keep < 0.0 ? 0.0 : (keep > 1.0 ? 1.0 : keep);
    const unsigned nb = s2_grid(N, 1024);
    if (conf_thres <= 0.f) {
        if (hipMemsetD32Async((hipDeviceptr_t)thr_out, (int)0xFF800000u /* -inf */, 1, stream) != hipSuccess) return VGPA_ERR_LAUNCH;
        return VGPA_OK;
    }
    VGPA_LAUNCH(select_hist_kernel<0>, dim3(nb), dim3(S2_THREADS), 0, stream, conf, N, st);
    VGPA_LAUNCH(select_pick_kernel<0>, dim3(1), dim3(64), 0, stream, st, keep, thr_out);
    VGPA_LAUNCH(select_hist_kernel<1>, dim3(nb), dim3(S2_THREADS), 0, stream, conf, N, st);
    VGPA_LAUNCH(select_pick_kernel<1>, dim3(1), dim3(64), 0, stream, st, keep, thr_out);
    VGPA_LAUNCH(select_hist_kernel<2>, dim3(nb), dim3(S2_THREADS), 0, stream, conf, N, st);
    VGPA_LAUNCH(select_pick_kernel<2>, dim3(1), dim3(64), 0, stream, st, keep, thr_out);
    VGPA_LAUNCH(select_hist_kernel<3>, dim3(nb), dim3(S2_THREADS), 0, stream, conf, N, st);
    VGPA_LAUNCH(select_pick_kernel<3>, dim3(1), dim3(64), 0, stream, st, keep, thr_out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

size_t vgpa_frame_metric_workspace_bytes(void) { return 1024 * sizeof(double) + 4 * sizeof(uint32_t); }

// MSEMetric / PSNRMetric.compute incl. the size-mismatch branch (metrics/mse.py:14-29,61-74): rep [T,C,H2,W2] is resized
// bilinearly (align_corners=False) to gt's [H,W] after the range heuristics.  psnr = 0: out = mse; 1: out = 10 log10(1/mse),
// 100 when mse == 0.  dtype / layout / is_tensor as vgpa_frame_mse.
int32_t vgpa_frame_metric(const void* gt, int32_t gt_dtype, int32_t gt_layout, int32_t gt_is_tensor, const void* rep, int32_t rep_dtype,
                          int32_t rep_layout, int32_t rep_is_tensor, int64_t T, int64_t C, int64_t H, int64_t W, int64_t H2, int64_t W2,
                          int32_t psnr, float* out, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!gt || !rep || !out || !workspace || T <= 0 || C <= 0 || H <= 0 || W <= 0 || H2 <= 0 || W2 <= 0) return VGPA_ERR_INVALID;
    if ((gt_dtype != 0 && gt_dtype != 2) || (rep_dtype != 0 && rep_dtype != 2)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_frame_metric_workspace_bytes()) return VGPA_ERR_WORKSPACE;
    double* partial = (double*)workspace;
    uint32_t* mm = (uint32_t*)((char*)workspace + 1024 * sizeof(double));
    const int64_t n = T * C * H * W, n2 = T * C * H2 * W2;
    if (hipMemsetAsync(mm, 0, 4 * sizeof(uint32_t), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    VGPA_LAUNCH(minmax1_kernel, dim3(s2_grid(n, 1024)), dim3(S2_THREADS), 0, stream, gt, gt_dtype, n, mm);
    VGPA_LAUNCH(minmax1_kernel, dim3(s2_grid(n2, 1024)), dim3(S2_THREADS), 0, stream, rep, rep_dtype, n2, mm + 2);
    VGPA_CHECK_LAUNCH();
    const unsigned nb = s2_grid(n, 1024);
    VGPA_LAUNCH(mse_resize_kernel, dim3(nb), dim3(S2_THREADS), 0, stream, gt, gt_dtype, gt_layout, gt_is_tensor, rep, rep_dtype, rep_layout,
                rep_is_tensor, T, (int)C, (int)H, (int)W, (int)H2, (int)W2, mm, partial);
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(mse_finish_kernel, dim3(1), dim3(S2_THREADS), 0, stream, partial, (int)nb, 1.0 / (double)n, (int)psnr, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// Input side of LPIPSMetric.compute (metrics/lpips.py:21-63): frames (dtype 0 f32 / 2 u8; layout 0 [T,C,H,W] / 1 [T,H,W,C]; is_tensor
// selects the torch.Tensor vs numpy rule) -> fp32 [T,C,Ho,Wo] in [-1,1], bilinearly resized (align_corners=False) when (Ho,Wo) != (H,W).
// workspace: 2 x uint32 (vgpa_frame_metric_workspace_bytes is enough).
int32_t vgpa_frames_to_pm1(const void* src, int32_t dtype, int32_t layout, int32_t is_tensor, int64_t T, int64_t C, int64_t H, int64_t W,
                           int64_t Ho, int64_t Wo, float* out, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!src || !out || !workspace || T <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || (dtype != 0 && dtype != 2)) return VGPA_ERR_INVALID;
    if (ws_bytes < 2 * sizeof(uint32_t)) return VGPA_ERR_WORKSPACE;
    uint32_t* mm = (uint32_t*)workspace;
    if (hipMemsetAsync(mm, 0, 2 * sizeof(uint32_t), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    const int64_t n_in = T * C * H * W, n_out = T * C * Ho * Wo;
    VGPA_LAUNCH(minmax1_kernel, dim3(s2_grid(n_in, 1024)), dim3(S2_THREADS), 0, stream, src, dtype, n_in, mm);
    VGPA_LAUNCH(frames_pm1_kernel, dim3(s2_grid(n_out, 2048)), dim3(S2_THREADS), 0, stream, src, dtype, layout, is_tensor, T, (int)C, (int)H, (int)W,
                (int)Ho, (int)Wo, mm, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

size_t vgpa_mvcs_workspace_bytes(int64_t T) { return (size_t)(T > 1 ? T - 1 : 1) * 16; }

// MVCSMetric.compute (metrics/mvcs.py:12-114).  depth fp32 [T,H,W]; K fp32 [T,k_dim,k_dim] (k_dim 3 or 4: the top-left 3x3 is
// used); E fp32 [T,e_rows(3|4),4] world-to-camera.  out[0] = exp(-mean over valid pairs of the masked depth MSE), 0 if none.
int32_t vgpa_mvcs(const float* depth, const float* K, int32_t k_dim, const float* E, int32_t e_rows, int64_t T, int64_t H, int64_t W,
                  float* out, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!depth || !K || !E || !out || !workspace || (k_dim != 3 && k_dim != 4) || (e_rows != 3 && e_rows != 4) || T <= 0 || T > 65535 ||
        H <= 1 || W <= 1)
        return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_mvcs_workspace_bytes(T)) return VGPA_ERR_WORKSPACE;
    const int np = (int)(T - 1);
    double* psum = (double*)workspace;
    unsigned long long* pcnt = (unsigned long long*)((char*)workspace + (size_t)(np > 0 ? np : 1) * 8);
    if (hipMemsetAsync(workspace, 0, vgpa_mvcs_workspace_bytes(T), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    if (np > 0) {
        VGPA_LAUNCH(mvcs_kernel, dim3(s2_grid(H * W, 256), (unsigned)np), dim3(S2_THREADS), 0, stream, depth, K, (int)(k_dim * k_dim), (int)k_dim, E,
                    (int)e_rows * 4, (int)H, (int)W, psum, pcnt);
        VGPA_CHECK_LAUNCH();
    }
    VGPA_LAUNCH(mvcs_finish_kernel, dim3(1), dim3(64), 0, stream, psum, pcnt, np, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// World points of DA3's depth maps: world = c2w [K^-1 (x, y, 1) depth] with c2w = affine_inverse(E)
// (pipelines/process_video.py:151-156).  depth fp32 [T,H,W]; K fp32 [T,3,3]; E fp32 [T,e_rows,4]; world fp32 [T,H,W,3].
int32_t vgpa_unproject_depth(const float* depth, const float* K, const float* E, int32_t e_rows, int64_t T, int64_t H, int64_t W, float* world,
                             hipStream_t stream) {
    if (!depth || !K || !E || !world || (e_rows != 3 && e_rows != 4) || T <= 0 || T > 65535 || H <= 0 || W <= 0) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(unproject_kernel, dim3(s2_grid(H * W, 1024), (unsigned)T), dim3(S2_THREADS), 0, stream, depth, K, E, (int)e_rows * 4, (int)H, (int)W,
                world);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// pose_encoding_to_extri_intri ("absT_quaR_FoV", vggt/utils/pose_enc.py:62-124): pe fp32 [n,9] -> ext fp32 [n,3,4],
// intr fp32 [n,3,3] (or NULL: build_intrinsics=False).
int32_t vgpa_pose_decode(const float* pose_enc, int64_t n, float image_h, float image_w, float* ext, float* intr, hipStream_t stream) {
    if (!pose_enc || !ext || n <= 0) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(pose_decode_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, pose_enc, n, image_h, image_w, ext, intr);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
