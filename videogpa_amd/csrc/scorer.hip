// Geometry-consistency scorer kernels: point-cloud reprojection, frame MSE, camera motion score,
// 8-point fundamental matrix + Sampson error.  Replaces, on device:
//   utils/projection_utils.py:12-51 (project_points) / :57-101 (batch_reproject) -- incl. the confidence filter of
//     utils/pointcloud_utils.py:10-80 folded in as a per-point predicate (no compacted cloud is materialised),
//   metrics/mse.py:14-54 (MSEMetric.compute + range heuristics),
//   metrics/consistency_score.py:8-40 (compute_motion_score_vectorized),
//   metrics/epipolar.py:197-213 (kornia find_fundamental + sampson_epipolar_distance, sqrt(d^2 + 1e-8) mean).
// Oracle: oracle/scorer.py.
//
// project_points: the reference sorts all points by depth (descending) and scatters colours so the nearest point
// written last wins a pixel.  Here each point does ONE 64-bit atomicMin of (depth bits << 32 | point index) into a
// z-buffer -- no sort, all T frames in one launch, HBM/atomic-bound -- and a second pass resolves colours.  Depth
// ties go to the lowest point index (the reference's tie order is unspecified).  The projection arithmetic is
// evaluated in a fixed fp32 order with contraction off, bit-identical to the oracle.
#include "common.h"

#pragma clang fp contract(off)

#define SC_THREADS 256
#define ZEMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint32_t f32_ordered(float f) {  // monotonic float -> uint map (for atomicMax on floats)
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(SC_THREADS) void project_zbuf_kernel(const float* __restrict__ pc, const float* __restrict__ colors,
                                                                    const float* __restrict__ conf, float conf_thr_val,
                                                                    const float* __restrict__ conf_thr_dev,
                                                                    const float* __restrict__ Kmat, const float* __restrict__ Emat,
                                                                    int e_stride, int64_t N, int H, int W,
                                                                    unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ cmax) {
    __shared__ uint32_t smax[SC_THREADS / 64];
    const int t = blockIdx.y;
    const float* K = Kmat + t * 9;
    const float* E = Emat + (size_t)t * e_stride;   // row-major [>=3, 4]
    float R[3][3], tr[3], Km[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { R[i][j] = E[i * 4 + j]; Km[i][j] = K[i * 3 + j]; }
        tr[i] = E[i * 4 + 3];
    }
    const float conf_thr = conf_thr_dev ? conf_thr_dev[0] : conf_thr_val;   // device-resident cut (vgpa_conf_threshold): no host sync
    uint32_t lmax = 0;  // ordered-uint of -NaN-free minimum
    for (int64_t i = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * SC_THREADS) {
        if (conf) {
            const float cf = conf[i];
            if (!(isfinite(cf) && cf > 1e-5f && cf >= conf_thr)) continue;
        }
        const float x = pc[3 * i], y = pc[3 * i + 1], z = pc[3 * i + 2];
        float cam[3], pr[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) cam[j] = ((x * R[j][0] + y * R[j][1]) + z * R[j][2]) + tr[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) pr[j] = (cam[0] * Km[j][0] + cam[1] * Km[j][1]) + cam[2] * Km[j][2];
        const float zz = pr[2];
        const float den = zz + 1e-8f;
        const float u = rintf(pr[0] / den), v = rintf(pr[1] / den);
        if (!(u >= 0.f && u < (float)W && v >= 0.f && v < (float)H && zz > 0.f)) continue;
        const float c0 = colors[3 * i], c1 = colors[3 * i + 1], c2 = colors[3 * i + 2];
        const uint32_t m = max(max(f32_ordered(c0), f32_ordered(c1)), f32_ordered(c2));
        lmax = max(lmax, m);
        const unsigned long long key = ((unsigned long long)__float_as_uint(zz) << 32) | (uint32_t)i;
        atomicMin(&zbuf[((size_t)t * H + (int)v) * W + (int)u], key);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, o, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (int i = 0; i < SC_THREADS / 64; ++i) m = max(m, smax[i]);
        if (m) atomicMax(&cmax[t], m);
    }
}

__global__ __launch_bounds__(SC_THREADS) void project_resolve_kernel(const unsigned long long* __restrict__ zbuf,
                                                                       const float* __restrict__ colors, const uint32_t* __restrict__ cmax,
                                                                       int64_t HW, uint8_t* __restrict__ canvas, float* __restrict__ out_f) {
    const int t = blockIdx.y;
    const bool unit = cmax[t] != 0 && f32_unordered(cmax[t]) <= 1.0f;   // c.max() <= 1.0 -> colours are in [0,1]
    for (int64_t p = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x; p < HW; p += (int64_t)gridDim.x * SC_THREADS) {
        const unsigned long long key = zbuf[(size_t)t * HW + p];
        uint8_t c[3] = {0, 0, 0};
        if (key != ZEMPTY) {
            const uint32_t idx = (uint32_t)key;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float f = colors[3 * (size_t)idx + j];
                if (unit) f = f * 255.f;
                f = fminf(fmaxf(f, 0.f), 255.f);
                c[j] = (uint8_t)f;   // truncating cast, as .to(torch.uint8)
            }
        }
        if (canvas) {
            uint8_t* o = canvas + ((size_t)t * HW + p) * 3;
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
        }
        if (out_f) {   // [T,3,H,W] = (u8 / 255) * 2 - 1
#pragma unroll
            for (int j = 0; j < 3; ++j) out_f[((size_t)t * 3 + j) * HW + p] = ((float)c[j] / 255.0f) * 2.0f - 1.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------- MSE with range heuristics
// layout 0: [T,C,H,W]; 1: [T,H,W,C].  dtype 0: f32, 2: u8.
__device__ __forceinline__ float img_at(const void* p, int dtype, int layout, int64_t t, int c, int64_t hw, int C, int64_t HW) {
    const size_t i = layout ? ((size_t)(t * HW + hw) * C + c) : ((size_t)(t * C + c) * HW + hw);
    return dtype == 2 ? (float)reinterpret_cast<const uint8_t*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}

__global__ __launch_bounds__(SC_THREADS) void minmax_kernel(const void* __restrict__ a, int a_dtype, const void* __restrict__ b, int b_dtype,
                                                              int64_t n, uint32_t* __restrict__ mm /* [4]: amin(~), amax, bmin(~), bmax */) {
    uint32_t lo_a = 0, hi_a = 0, lo_b = 0, hi_b = 0;   // min tracked as max of the complemented ordered value
    for (int64_t i = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SC_THREADS) {
        const float x = a_dtype == 2 ? (float)reinterpret_cast<const uint8_t*>(a)[i] : reinterpret_cast<const float*>(a)[i];
        const float y = b_dtype == 2 ? (float)reinterpret_cast<const uint8_t*>(b)[i] : reinterpret_cast<const float*>(b)[i];
        const uint32_t ox = f32_ordered(x), oy = f32_ordered(y);
        hi_a = max(hi_a, ox); lo_a = max(lo_a, ~ox);
        hi_b = max(hi_b, oy); lo_b = max(lo_b, ~oy);
    }
    uint32_t v[4] = {lo_a, hi_a, lo_b, hi_b};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] = max(v[k], (uint32_t)__shfl_xor((int)v[k], o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&mm[k], v[k]);
    }
}

__device__ __forceinline__ float to01(float x, float mn, float mx, int is_tensor) {
    // metrics/mse.py:31-54: tensors: min < 0 -> (x+1)/2, elif max > 1 -> x/255 ; numpy: max > 1 -> x/255
    if (is_tensor && mn < 0.f) return (x + 1.0f) / 2.0f;
    if (mx > 1.0f) return x / 255.0f;
    return x;
}

__global__ __launch_bounds__(SC_THREADS) void mse_kernel(const void* __restrict__ gt, int gt_dtype, int gt_layout, int gt_is_tensor,
                                                           const void* __restrict__ rep, int rep_dtype, int rep_layout, int rep_is_tensor,
                                                           int64_t T, int C, int64_t HW, const uint32_t* __restrict__ mm,
                                                           double* __restrict__ partial) {
    __shared__ double smem[16];
    const float amin = f32_unordered(~mm[0]), amax = f32_unordered(mm[1]), bmin = f32_unordered(~mm[2]), bmax = f32_unordered(mm[3]);
    const int64_t n = T * C * HW;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SC_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SC_THREADS) {
        const int64_t hw = i % HW;
        const int c = (int)((i / HW) % C);
        const int64_t t = i / (HW * C);
        const float x = to01(img_at(gt, gt_dtype, gt_layout, t, c, hw, C, HW), amin, amax, gt_is_tensor);
        const float y = to01(img_at(rep, rep_dtype, rep_layout, t, c, hw, C, HW), bmin, bmax, rep_is_tensor);
        const float d = x - y;
        acc += d * d;
    }
    const double s = block_sum<double>((double)acc, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(SC_THREADS) void mean_finish_kernel(const double* __restrict__ partial, int nblk, double inv_n, float* __restrict__ out) {
    __shared__ double smem[16];
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += SC_THREADS) s += partial[i];
    s = block_sum<double>(s, smem);
    if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}

// ---------------------------------------------------------------------------------------------- motion score
__global__ void motion_score_kernel(const float* __restrict__ E, int e_stride, int T, float* __restrict__ out) {
    // one wave; pairs are few (T ~ 10)
    float st = 0.f, sr = 0.f;
    for (int i = threadIdx.x; i < T - 1; i += 64) {
        const float* A = E + (size_t)i * e_stride;
        const float* B = E + (size_t)(i + 1) * e_stride;
        const float dx = B[3] - A[3], dy = B[7] - A[7], dz = B[11] - A[11];
        st += sqrtf((dx * dx + dy * dy) + dz * dz);
        float trc = 0.f;   // trace(R_{i+1} R_i^T) = sum_{jk} B[j][k] A[j][k]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) d += B[j * 4 + k] * A[j * 4 + k];
            trc += d;
        }
        const float cv = fminf(fmaxf((trc - 1.f) / 2.f, -1.f), 1.f);
        sr += acosf(cv);
    }
    st = wave_sum(st);
    sr = wave_sum(sr);
    if (threadIdx.x == 0) {
        const float n = (float)(T - 1);
        const float s = st / n + 0.1f * (sr / n);
        out[0] = (s != s) ? 0.f : s;   // NaN -> 0 (metrics/consistency_score.py:37-38); T == 1 gives 0/0 -> 0
    }
}

// ---------------------------------------------------------------------------------------------- 8-point + Sampson
__device__ void jacobi_sym(double* a, double* v, int n) {  // a: n*n symmetric (destroyed -> diag = eigenvalues), v: eigenvectors (columns)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[k * n + p], vkq = v[k * n + q];
                    v[k * n + p] = c * vkp - s * vkq;
                    v[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

// one block per frame pair; points of pair i are p1/p2[offsets[i] .. offsets[i+1])
__global__ __launch_bounds__(SC_THREADS) void epipolar_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                                const int64_t* __restrict__ offsets, float* __restrict__ err_out,
                                                                float* __restrict__ F_out) {
    __shared__ double smem[16];
    __shared__ double sh[64];
    __shared__ double Fm[9];
    const int pair = blockIdx.x;
    const int64_t beg = offsets[pair], n = offsets[pair + 1] - beg;
    const float* a = p1 + 2 * beg;
    const float* b = p2 + 2 * beg;
    if (n < 8) {
        if (threadIdx.x == 0) { err_out[pair] = -1.f; if (F_out) for (int i = 0; i < 9; ++i) F_out[pair * 9 + i] = 0.f; }
        return;
    }
    // Hartley normalisation of each set: centroid 0, mean distance sqrt(2)
    double s[4] = {0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < n; i += SC_THREADS) { s[0] += a[2 * i]; s[1] += a[2 * i + 1]; s[2] += b[2 * i]; s[3] += b[2 * i + 1]; }
    double mean[4];
    for (int k = 0; k < 4; ++k) mean[k] = block_sum<double>(s[k], smem) / (double)n;
    double d1 = 0, d2 = 0;
    for (int64_t i = threadIdx.x; i < n; i += SC_THREADS) {
        d1 += sqrt((a[2 * i] - mean[0]) * (a[2 * i] - mean[0]) + (a[2 * i + 1] - mean[1]) * (a[2 * i + 1] - mean[1]));
        d2 += sqrt((b[2 * i] - mean[2]) * (b[2 * i] - mean[2]) + (b[2 * i + 1] - mean[3]) * (b[2 * i + 1] - mean[3]));
    }
    const double sc1 = sqrt(2.0) / (block_sum<double>(d1, smem) / (double)n + 1e-8);
    const double sc2 = sqrt(2.0) / (block_sum<double>(d2, smem) / (double)n + 1e-8);
    // X^T X, 45 unique entries
    double acc[45];
    for (int k = 0; k < 45; ++k) acc[k] = 0;
    for (int64_t i = threadIdx.x; i < n; i += SC_THREADS) {
        const double x1 = (a[2 * i] - mean[0]) * sc1, y1 = (a[2 * i + 1] - mean[1]) * sc1;
        const double x2 = (b[2 * i] - mean[2]) * sc2, y2 = (b[2 * i + 1] - mean[3]) * sc2;
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        int k = 0;
        for (int p = 0; p < 9; ++p)
            for (int q = p; q < 9; ++q) acc[k++] += r[p] * r[q];
    }
    for (int k = 0; k < 45; ++k) {
        const double t = block_sum<double>(acc[k], smem);
        if (threadIdx.x == 0) sh[k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A[81], V[81];
        int k = 0;
        for (int p = 0; p < 9; ++p)
            for (int q = p; q < 9; ++q) { A[p * 9 + q] = sh[k]; A[q * 9 + p] = sh[k]; ++k; }
        jacobi_sym(A, V, 9);
        int imin = 0;
        for (int i = 1; i < 9; ++i) if (A[i * 9 + i] < A[imin * 9 + imin]) imin = i;
        double F[9];
        for (int i = 0; i < 9; ++i) F[i] = V[i * 9 + imin];
        // rank-2 projection: F <- F (I - v3 v3^T), v3 = eigenvector of F^T F with the smallest eigenvalue
        double G[9], W3[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { double t = 0; for (int m = 0; m < 3; ++m) t += F[m * 3 + i] * F[m * 3 + j]; G[i * 3 + j] = t; }
        jacobi_sym(G, W3, 3);
        int jm = 0;
        for (int i = 1; i < 3; ++i) if (G[i * 3 + i] < G[jm * 3 + jm]) jm = i;
        const double v3[3] = {W3[0 * 3 + jm], W3[1 * 3 + jm], W3[2 * 3 + jm]};
        double Fv[3];
        for (int i = 0; i < 3; ++i) Fv[i] = F[i * 3] * v3[0] + F[i * 3 + 1] * v3[1] + F[i * 3 + 2] * v3[2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) F[i * 3 + j] -= Fv[i] * v3[j];
        // denormalise: T2^T F T1 with T = [[s,0,-s mx],[0,s,-s my],[0,0,1]]
        const double T1[9] = {sc1, 0, -sc1 * mean[0], 0, sc1, -sc1 * mean[1], 0, 0, 1};
        const double T2[9] = {sc2, 0, -sc2 * mean[2], 0, sc2, -sc2 * mean[3], 0, 0, 1};
        double FT1[9], R9[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { double t = 0; for (int m = 0; m < 3; ++m) t += F[i * 3 + m] * T1[m * 3 + j]; FT1[i * 3 + j] = t; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { double t = 0; for (int m = 0; m < 3; ++m) t += T2[m * 3 + i] * FT1[m * 3 + j]; R9[i * 3 + j] = t; }
        if (fabs(R9[8]) > 1e-8) { const double d = R9[8] + 1e-8; for (int i = 0; i < 9; ++i) R9[i] /= d; }
        for (int i = 0; i < 9; ++i) Fm[i] = R9[i];
        if (F_out) for (int i = 0; i < 9; ++i) F_out[pair * 9 + i] = (float)R9[i];
    }
    __syncthreads();
    double e = 0;
    for (int64_t i = threadIdx.x; i < n; i += SC_THREADS) {
        const double x1 = a[2 * i], y1 = a[2 * i + 1], x2 = b[2 * i], y2 = b[2 * i + 1];
        const double l0 = Fm[0] * x1 + Fm[1] * y1 + Fm[2], l1 = Fm[3] * x1 + Fm[4] * y1 + Fm[5], l2 = Fm[6] * x1 + Fm[7] * y1 + Fm[8];   // F x1
        const double m0 = Fm[0] * x2 + Fm[3] * y2 + Fm[6], m1 = Fm[1] * x2 + Fm[4] * y2 + Fm[7];                                        // F^T x2
        const double num = (x2 * l0 + y2 * l1 + l2);
        const double d2s = num * num / (l0 * l0 + l1 * l1 + m0 * m0 + m1 * m1);
        e += sqrt(d2s + 1e-8);
    }
    e = block_sum<double>(e, smem);
    if (threadIdx.x == 0) err_out[pair] = (float)(e / (double)n);
}

static inline unsigned sc_grid(int64_t n, int cap) {
    int64_t nb = (n + SC_THREADS - 1) / SC_THREADS;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

extern "C" {

size_t vgpa_project_points_workspace_bytes(int64_t T, int64_t H, int64_t W) { return (size_t)T * H * W * 8 + (size_t)T * 4; }

// Render T views of one coloured cloud.  pc/colors fp32 [N,3]; conf fp32 [N] or NULL (points with non-finite conf,
// conf <= 1e-5 or conf < threshold are skipped; threshold = conf_thr_dev[0] when that device pointer is given, else the
// by-value conf_thr); K fp32 [T,3,3]; E fp32 [T, e_rows(3|4), 4].
// Outputs (either may be NULL): canvas u8 [T,H,W,3]; out_f fp32 [T,3,H,W] in [-1,1].
int32_t vgpa_project_points(const float* pc, const float* colors, const float* conf, float conf_thr, const float* conf_thr_dev,
                            const float* K, const float* E, int32_t e_rows, int64_t N, int64_t T, int64_t H, int64_t W, uint8_t* canvas, float* out_f, void* workspace,
                            size_t ws_bytes, hipStream_t stream) {
    if (!K || !E || !workspace || (N > 0 && (!pc || !colors)) || (e_rows != 3 && e_rows != 4) || N < 0 || N > 0xFFFFFFFFll || T <= 0 ||
        T > 65535 || H <= 0 || W <= 0 || (!canvas && !out_f))
        return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_project_points_workspace_bytes(T, H, W)) return VGPA_ERR_WORKSPACE;
    unsigned long long* zbuf = (unsigned long long*)workspace;
    uint32_t* cmax = (uint32_t*)((char*)workspace + (size_t)T * H * W * 8);
    if (hipMemsetAsync(zbuf, 0xFF, (size_t)T * H * W * 8, stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    if (hipMemsetAsync(cmax, 0, (size_t)T * 4, stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    if (N > 0) {
        VGPA_LAUNCH(project_zbuf_kernel, dim3(sc_grid(N, 2048), (unsigned)T), dim3(SC_THREADS), 0, stream, pc, colors, conf, conf_thr, conf_thr_dev, K, E,
                    (int)e_rows * 4, N, (int)H, (int)W, zbuf, cmax);
        VGPA_CHECK_LAUNCH();
    }
    VGPA_LAUNCH(project_resolve_kernel, dim3(sc_grid(H * W, 1024), (unsigned)T), dim3(SC_THREADS), 0, stream, zbuf, colors, cmax, H * W, canvas,
                out_f);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

size_t vgpa_frame_mse_workspace_bytes(void) { return 1024 * sizeof(double) + 4 * sizeof(uint32_t); }

// mean((gt01 - rep01)^2) with the reference's range heuristics.  dtype: 0 f32, 2 u8; layout: 0 [T,C,H,W], 1 [T,H,W,C];
// is_tensor: 1 = torch.Tensor rules (min<0 -> [-1,1]; max>1 -> [0,255]), 0 = numpy rules (max>1 -> [0,255]).
int32_t vgpa_frame_mse(const void* gt, int32_t gt_dtype, int32_t gt_layout, int32_t gt_is_tensor, const void* rep, int32_t rep_dtype,
                       int32_t rep_layout, int32_t rep_is_tensor, int64_t T, int64_t C, int64_t H, int64_t W, float* out, void* workspace,
                       size_t ws_bytes, hipStream_t stream) {
    if (!gt || !rep || !out || !workspace || T <= 0 || C <= 0 || H <= 0 || W <= 0) return VGPA_ERR_INVALID;
    if ((gt_dtype != 0 && gt_dtype != 2) || (rep_dtype != 0 && rep_dtype != 2)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_frame_mse_workspace_bytes()) return VGPA_ERR_WORKSPACE;
    double* partial = (double*)workspace;
    uint32_t* mm = (uint32_t*)((char*)workspace + 1024 * sizeof(double));
    const int64_t n = T * C * H * W;
    if (hipMemsetAsync(mm, 0, 4 * sizeof(uint32_t), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    const unsigned nb = sc_grid(n, 1024);
    VGPA_LAUNCH(minmax_kernel, dim3(nb), dim3(SC_THREADS), 0, stream, gt, gt_dtype, rep, rep_dtype, n, mm);
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(mse_kernel, dim3(nb), dim3(SC_THREADS), 0, stream, gt, gt_dtype, gt_layout, gt_is_tensor, rep, rep_dtype, rep_layout,
                rep_is_tensor, T, (int)C, H * W, mm, partial);
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(mean_finish_kernel, dim3(1), dim3(SC_THREADS), 0, stream, partial, (int)nb, 1.0 / (double)n, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// out[0] = mean||dt|| + 0.1 * mean acos(clamp((tr(R_{i+1} R_i^T) - 1)/2)), NaN -> 0.  E fp32 [T, e_rows(3|4), 4].
int32_t vgpa_motion_score(const float* E, int32_t e_rows, int64_t T, float* out, hipStream_t stream) {
    if (!E || !out || (e_rows != 3 && e_rows != 4) || T <= 0) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(motion_score_kernel, dim3(1), dim3(64), 0, stream, E, (int)e_rows * 4, (int)T, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// Per frame pair i (matched points p1/p2[offsets[i]:offsets[i+1]], fp32 [.,2]): normalised 8-point fundamental matrix
// (unit weights) and err[i] = mean sqrt(sampson^2 + 1e-8); err = -1 when the pair has < 8 matches.  F_out fp32 [P,9] or NULL.
int32_t vgpa_epipolar_sampson(const float* p1, const float* p2, const int64_t* offsets, int64_t n_pairs, float* err_out, float* F_out,
                              hipStream_t stream) {
    if (!p1 || !p2 || !offsets || !err_out || n_pairs <= 0 || n_pairs > 0x7fffffff) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(epipolar_kernel, dim3((unsigned)n_pairs), dim3(SC_THREADS), 0, stream, p1, p2, offsets, err_out, F_out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
