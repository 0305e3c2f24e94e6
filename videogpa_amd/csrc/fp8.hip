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

extern "C" int32_t vgpa_quant_fp8_rows(const void* x, int64_t ldx, void* q, float* scale, int64_t M, int64_t K, hipStream_t stream) {
    if (!x || !q || !scale || M <= 0 || K <= 0 || K % 8 || ldx < K || ldx % 8 || K > (1 << 24)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(quant_fp8_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)q, scale, M, (int)K);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
