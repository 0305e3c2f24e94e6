// Fused forward-diffusion noising + v-prediction target over the paired latent layout.
// Replaces scheduler.add_noise x2 + scheduler.get_velocity x2 at train/CogVideoX-5B/03_train.py:129-130,154-155
// (diffusers CogVideoXDPMScheduler.add_noise/get_velocity).  One pass: reads x[B,2,N] and the shared
// noise[B,N] once, writes x_t[B,2,N] and v[B,2,N].
//   x_t = sa[t] * x + sb[t] * eps        v = sa[t] * eps - sb[t] * x
// Arithmetic mirrors torch elementwise semantics exactly: every product / sum is rounded to the storage dtype
// (bf16: round-to-nearest-even after each op; fp32: separate mul/add, no FMA contraction), so the result is
// bit-identical to the reference expression evaluated by PyTorch in that dtype.
#include "common.h"

#pragma clang fp contract(off)  // separate mul / add roundings, exactly like torch elementwise ops

template <int DT>
__device__ __forceinline__ float rnd(float v) { return DT == VGPA_DTYPE_BF16 ? round_bf16(v) : v; }

template <int DT>
__global__ __launch_bounds__(256) void noise_velocity_kernel(const void* __restrict__ x, const void* __restrict__ noise,
                                                               const int64_t* __restrict__ t, const float* __restrict__ sa_tab,
                                                               const float* __restrict__ sb_tab, int64_t N, int T,
                                                               void* __restrict__ xt, void* __restrict__ v) {
    const int b = blockIdx.y;
    int64_t ti = t[b];
    ti = ti < 0 ? 0 : (ti >= T ? T - 1 : ti);
    const float sa = sa_tab[ti], sb = sb_tab[ti];
    const size_t on = (size_t)b * N, ox = (size_t)b * 2 * N;
    const int64_t n8 = N >> 3;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n8; c += (int64_t)gridDim.x * 256) {
        const size_t i = (size_t)c << 3;
        float e[8], a[8], o1[8], o2[8];
        load8<DT>(noise, on + i, e);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load8<DT>(x, ox + (size_t)p * N + i, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o1[j] = rnd<DT>(rnd<DT>(sa * a[j]) + rnd<DT>(sb * e[j]));
                o2[j] = rnd<DT>(rnd<DT>(sa * e[j]) - rnd<DT>(sb * a[j]));
            }
            store8<DT>(xt, ox + (size_t)p * N + i, o1);
            store8<DT>(v, ox + (size_t)p * N + i, o2);
        }
    }
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const float e = load1<DT>(noise, on + i);
        for (int p = 0; p < 2; ++p) {
            const float a = load1<DT>(x, ox + (size_t)p * N + i);
            store1<DT>(xt, ox + (size_t)p * N + i, rnd<DT>(rnd<DT>(sa * a) + rnd<DT>(sb * e)));
            store1<DT>(v, ox + (size_t)p * N + i, rnd<DT>(rnd<DT>(sa * e) - rnd<DT>(sb * a)));
        }
    }
}

// Flow-matching form (train/Wan2.2-TI2V-5B/03_train.py:103-116,203-207,235-236): sigma[b] fp32,
//   x_t = (1 - sigma) * x + sigma * eps   evaluated in fp32 as torch's type promotion does (fp32 sigma times bf16 latent),
//   v   = eps - x                         in the latent dtype.
// XT_F32 selects an fp32 x_t (the reference's result dtype) or one rounded to the latent dtype.
template <int DT, bool XT_F32>
__global__ __launch_bounds__(256) void flow_noise_velocity_kernel(const void* __restrict__ x, const void* __restrict__ noise,
                                                                    const float* __restrict__ sigma, int64_t N, void* __restrict__ xt,
                                                                    void* __restrict__ v) {
    const int b = blockIdx.y;
    const float sg = sigma[b];
    const float om = 1.0f - sg;
    const size_t on = (size_t)b * N, ox = (size_t)b * 2 * N;
    const int64_t n8 = N >> 3;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n8; c += (int64_t)gridDim.x * 256) {
        const size_t i = (size_t)c << 3;
        float e[8], a[8], o1[8], o2[8];
        load8<DT>(noise, on + i, e);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load8<DT>(x, ox + (size_t)p * N + i, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o1[j] = om * a[j] + sg * e[j];
                o2[j] = rnd<DT>(e[j] - a[j]);
            }
            if (XT_F32) store8<VGPA_DTYPE_F32>(xt, ox + (size_t)p * N + i, o1);
            else store8<DT>(xt, ox + (size_t)p * N + i, o1);
            store8<DT>(v, ox + (size_t)p * N + i, o2);
        }
    }
}

extern "C" int32_t vgpa_flow_noise_velocity_paired(const void* x_pair, const void* noise, const float* sigma, int64_t B, int64_t N, int32_t dtype,
                                                   int32_t xt_f32, void* x_noisy_pair, void* v_target_pair, hipStream_t stream) {
    if (!x_pair || !noise || !sigma || !x_noisy_pair || !v_target_pair || B <= 0 || N <= 0 || B > 65535 || N % 8 != 0) return VGPA_ERR_INVALID;
    int64_t nb = (N / 8 + 255) / 256;
    if (nb > 2048) nb = 2048;
    dim3 grid((unsigned)nb, (unsigned)B);
#define FL(DT, F) VGPA_LAUNCH((flow_noise_velocity_kernel<DT, F>), grid, dim3(256), 0, stream, x_pair, noise, sigma, N, x_noisy_pair, v_target_pair)
    if (dtype == VGPA_DTYPE_BF16) { if (xt_f32) FL(VGPA_DTYPE_BF16, true); else FL(VGPA_DTYPE_BF16, false); }
    else if (dtype == VGPA_DTYPE_F32) FL(VGPA_DTYPE_F32, true);
    else return VGPA_ERR_INVALID;
#undef FL
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_noise_velocity_paired(const void* x_pair, const void* noise, const int64_t* t, const float* sqrt_abar,
                                              const float* sqrt_1m_abar, int64_t B, int64_t N, int32_t num_train_timesteps,
                                              int32_t dtype, void* x_noisy_pair, void* v_target_pair, hipStream_t stream) {
    if (!x_pair || !noise || !t || !sqrt_abar || !sqrt_1m_abar || !x_noisy_pair || !v_target_pair) return VGPA_ERR_INVALID;
    if (B <= 0 || N <= 0 || B > 65535 || num_train_timesteps <= 0) return VGPA_ERR_INVALID;
    if (N % 8 != 0) return VGPA_ERR_INVALID;  // latent volumes are multiples of 8 (C = 16)
    int64_t nb = (N / 8 + 255) / 256;
    if (nb > 2048) nb = 2048;
    dim3 grid((unsigned)nb, (unsigned)B);
    if (dtype == VGPA_DTYPE_BF16)
        VGPA_LAUNCH((noise_velocity_kernel<VGPA_DTYPE_BF16>), grid, dim3(256), 0, stream, x_pair, noise, t, sqrt_abar, sqrt_1m_abar, N,
                           num_train_timesteps, x_noisy_pair, v_target_pair);
    else if (dtype == VGPA_DTYPE_F32)
        VGPA_LAUNCH((noise_velocity_kernel<VGPA_DTYPE_F32>), grid, dim3(256), 0, stream, x_pair, noise, t, sqrt_abar, sqrt_1m_abar, N,
                           num_train_timesteps, x_noisy_pair, v_target_pair);
    else
        return VGPA_ERR_INVALID;
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
