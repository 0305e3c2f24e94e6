// Dynamic per-row quantisation of bf16 activations to OCP e4m3 (the "fp8 MFMA path" BASELINE.json configs[4] names for
// Wan2.2-TI2V-5B): the operand format of the vendor fp8 GEMM (hipBLASLt through torch._scaled_mm, 2.1-2.9 PFLOP/s at the
// feed-forward shapes against 1.2-1.3 in bf16, profiles/r03_fp8_probe.txt) that the frozen feed-forward projections of
// videogpa_amd/wan_model.py run when fp8 is switched on.
//   scale[m] = max_k |x[m,k]| / 448   (1 for an all-zero row),   q[m,k] = e4m3(x[m,k] / scale[m])   round-to-nearest-even, saturating
// One wave per row, two passes over the row (the second one hits L2): rows up to 14336 wide do not fit the register file.
// HBM-bound: 2 bytes read + 1 written per element.
#include "common.h"

#define FP8_E4M3_MAX 448.0f

__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, float* __restrict__ scale,
                                                               int64_t M, int K) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* xr = x + (size_t)row * ldx;
    float amax = 0.f;
    for (int i0 = lane * 8; i0 < K; i0 += 512) {
        float v[8];
        load8<VGPA_DTYPE_BF16>(xr, (size_t)i0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / FP8_E4M3_MAX : 1.f;
    if (lane == 0) scale[row] = sc;
    uint8_t* qr = q + (size_t)row * K;
    for (int i0 = lane * 8; i0 < K; i0 += 512) {
        float v[8];
        load8<VGPA_DTYPE_BF16>(xr, (size_t)i0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(v[j] / sc, -FP8_E4M3_MAX), FP8_E4M3_MAX);
        u32x2_t w;
        int t = 0;
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], t, false);
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], t, true);
        w[0] = (uint32_t)t;
        t = 0;
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], t, false);
        t = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], t, true);
        w[1] = (uint32_t)t;
        *reinterpret_cast<u32x2_t*>(qr + i0) = w;
    }
}

// ---- producers that write the e4m3 operand themselves (no bf16 round trip through HBM): GELU forward / backward of the feed-forward.
// One 256-thread workgroup per row, the row (<= 16384 wide) in registers as bf16-rounded values; results are bit-identical to the unfused
// chain  gelu_tanh_{fwd,bwd} -> quant_fp8_rows  (tests/test_gpu_wan_kernels.py).
#define Q8_NV 8
#define GELU_K1 (-2.302208198f)      // as csrc/norm.hip
#define GELU_K2 (-0.1029432396f)
__device__ __forceinline__ float q8_gelu_sig(float x, float x2) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (GELU_K1 + GELU_K2 * x2))); }

__device__ __forceinline__ float q8_block_max(float v, float* smem) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
}

__device__ __forceinline__ void q8_store_row(const float (*v)[8], int K, float sc, uint8_t* __restrict__ qr) {
#pragma unroll
    for (int c = 0; c < Q8_NV; ++c) {
        const int i0 = (c * 256 + (int)threadIdx.x) * 8;
        if (i0 < K) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fminf(fmaxf(v[c][j] / sc, -FP8_E4M3_MAX), FP8_E4M3_MAX);
            u32x2_t w;
            int p = 0;
            p = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], p, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], p, true);
            w[0] = (uint32_t)p;
            p = 0;
            p = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], p, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], p, true);
            w[1] = (uint32_t)p;
            *reinterpret_cast<u32x2_t*>(qr + i0) = w;
        }
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void gelu_q8_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dy, int K, uint8_t* __restrict__ q,
                                                        float* __restrict__ scale) {
    __shared__ float smem[4];
    const int64_t row = blockIdx.x;
    const bf16_t* ur = u + (size_t)row * K;
    // all loads of the row first, in a loop of their own: as `if (i0 < K) { load; use }` per chunk the forward's loads each got an s_waitcnt vmcnt(0)
    // behind them (one HBM round trip per chunk and wave)
    const u32x4_t z4 = {0u, 0u, 0u, 0u};
    u32x4_t ua[Q8_NV], ga[BWD ? Q8_NV : 1];
#pragma unroll
    for (int c = 0; c < Q8_NV; ++c) {
        const int i0 = (c * 256 + (int)threadIdx.x) * 8;
        ua[c] = i0 < K ? *reinterpret_cast<const u32x4_t*>(ur + i0) : z4;
        if (BWD) ga[c] = i0 < K ? *reinterpret_cast<const u32x4_t*>(dy + ((size_t)row * K + i0)) : z4;
    }
    float v[Q8_NV][8];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < Q8_NV; ++c) {
        const int i0 = (c * 256 + (int)threadIdx.x) * 8;
        if (i0 < K) {
            float a[8], g[8];
            unpack8(ua[c], a);
            if (BWD) unpack8(ga[c], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = a[j], x2 = x * x, sg = q8_gelu_sig(x, x2);
                float r;
                if (BWD) {
                    const float dz2 = 1.5957691216057308f + 0.2140644488f * x2;
                    r = g[j] * (sg + x * (sg - sg * sg) * dz2);
                } else {
                    r = x * sg;
                }
                v[c][j] = round_bf16(r);
                amax = fmaxf(amax, fabsf(v[c][j]));
            }
        }
    }
    amax = q8_block_max(amax, smem);
    const float sc = amax > 0.f ? amax / FP8_E4M3_MAX : 1.f;
    if (threadIdx.x == 0) scale[row] = sc;
    q8_store_row(v, K, sc, q + (size_t)row * K);
}

extern "C" int32_t vgpa_gelu_tanh_fwd_q8(const void* u, int64_t rows, int64_t K, void* q8, float* q8_scale, hipStream_t stream) {
    if (!u || !q8 || !q8_scale || rows <= 0 || K <= 0 || K % 8 || K > 256 * 8 * Q8_NV) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(gelu_q8_kernel<false>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)u, (const bf16_t*)nullptr, (int)K, (uint8_t*)q8, q8_scale);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_gelu_tanh_bwd_q8(const void* u, const void* dy, int64_t rows, int64_t K, void* q8, float* q8_scale, hipStream_t stream) {
    if (!u || !dy || !q8 || !q8_scale || rows <= 0 || K <= 0 || K % 8 || K > 256 * 8 * Q8_NV) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(gelu_q8_kernel<true>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)u, (const bf16_t*)dy, (int)K, (uint8_t*)q8, q8_scale);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_quant_fp8_rows(const void* x, int64_t ldx, void* q, float* scale, int64_t M, int64_t K, hipStream_t stream) {
    if (!x || !q || !scale || M <= 0 || K <= 0 || K % 8 || ldx < K || ldx % 8 || K > (1 << 24)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(quant_fp8_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)q, scale, M, (int)K);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
