// AdaLN-Zero normalisation kernels for the mixed [text | video] token stream.
// Replaces diffusers CogVideoXLayerNormZero / AdaLayerNorm / nn.LayerNorm as reached from
// train/CogVideoX-5B/03_train.py:134-151 (transformer forward) -- see oracle/cogvideox.py::block_forward.
//
// Layout: one residual stream x[B,S,D] bf16, text tokens first (rows < text_len), video tokens after.  The
// per-range modulation vectors (shift, 1+scale, gate) are tiny fp32 [B,D] arrays, so a row only has to pick
// the pointer for its range: no torch.cat / split of the two streams is ever materialised.
//
// HBM-bound: one wave per token row, the whole row lives in registers (D <= 4096), 16-byte loads, exact
// two-pass mean/variance, wave-level reductions only (no LDS, no barriers).
#include "common.h"

#define WAVES_PER_BLOCK 4
#define ROWS_PER_WAVE 4
#define ROWS_PER_BLOCK (WAVES_PER_BLOCK * ROWS_PER_WAVE)

// A wave walks ROWS_PER_WAVE consecutive token rows and keeps the per-column affine of its lanes in registers:
//   y = xhat * alpha + beta,  alpha = w * (1+scale),  beta = b * (1+scale) + shift
// so the four fp32 parameter vectors are fetched once per wave (and again only if the rows cross the text/video
// boundary or a batch boundary), not once per element.
template <int NV>
__device__ __forceinline__ void load_affine(const float* __restrict__ w, const float* __restrict__ bias, const float* sh,
                                            const float* sc, int D, int lane, float (&alpha)[NV][8], float (&beta)[NV][8]) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 wv = *reinterpret_cast<const float4*>(w + i0 + 4 * h);
                const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + i0 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 sv = make_float4(1.f, 1.f, 1.f, 1.f), hv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sc) sv = *reinterpret_cast<const float4*>(sc + i0 + 4 * h);
                if (sh) hv = *reinterpret_cast<const float4*>(sh + i0 + 4 * h);
                alpha[c][4 * h + 0] = wv.x * sv.x; alpha[c][4 * h + 1] = wv.y * sv.y;
                alpha[c][4 * h + 2] = wv.z * sv.z; alpha[c][4 * h + 3] = wv.w * sv.w;
                beta[c][4 * h + 0] = bv.x * sv.x + hv.x; beta[c][4 * h + 1] = bv.y * sv.y + hv.y;
                beta[c][4 * h + 2] = bv.z * sv.z + hv.z; beta[c][4 * h + 3] = bv.w * sv.w + hv.w;
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void ln_modulate_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ shift_v, const float* __restrict__ scale1p_v, const float* __restrict__ shift_t,
    const float* __restrict__ scale1p_t, int64_t mod_stride, int text_len, int S, int D, int64_t rows, float eps,
    bf16_t* __restrict__ out, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    float alpha[NV][8], beta[NV][8];
    int cur_key = -1;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= rows) return;
        const int b = (int)(row / S), s = (int)(row % S);
        const int key = b * 2 + (s < text_len ? 1 : 0);
        if (key != cur_key) {
            cur_key = key;
            const float* sh = (s < text_len) ? shift_t : shift_v;
            const float* sc = (s < text_len) ? scale1p_t : scale1p_v;
            if (shift_v) { sh += (size_t)b * mod_stride; sc += (size_t)b * mod_stride; } else { sh = nullptr; sc = nullptr; }
            load_affine<NV>(w, bias, sh, sc, D, lane, alpha, beta);
        }
        const bf16_t* xr = x + (size_t)row * D;
        float v[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                unpack8(*reinterpret_cast<const u32x4_t*>(xr + i0), v[c]);
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[c][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
        if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
        bf16_t* orow = out + (size_t)row * D;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * alpha[c][j] + beta[c][j];
                *reinterpret_cast<u32x4_t*>(orow + i0) = pack8(o);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * (1+scale) * w ;  optionally dx += dres
template <int NV>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void ln_modulate_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, const float* __restrict__ w, const float* __restrict__ scale1p_v,
    const float* __restrict__ scale1p_t, int64_t mod_stride, int text_len, int S, int D, int64_t rows,
    const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    float alpha[NV][8], beta_unused[NV][8];
    int cur_key = -1;
    for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= rows) return;
        const int b = (int)(row / S), s = (int)(row % S);
        const int key = b * 2 + (s < text_len ? 1 : 0);
        if (key != cur_key) {
            cur_key = key;
            const float* sc = (s < text_len) ? scale1p_t : scale1p_v;
            if (scale1p_v) sc += (size_t)b * mod_stride; else sc = nullptr;
            load_affine<NV>(w, nullptr, nullptr, sc, D, lane, alpha, beta_unused);
        }
        const float mean = mean_in[row], rstd = rstd_in[row];
        float g[NV][8], xh[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float a[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(dy + (size_t)row * D + i0), a);
                unpack8(*reinterpret_cast<const u32x4_t*>(x + (size_t)row * D + i0), xh[c]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gg = a[j] * alpha[c][j];
                    g[c][j] = gg;
                    xh[c][j] = (xh[c][j] - mean) * rstd;
                    s1 += gg;
                    s2 += gg * xh[c][j];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float o[8];
                if (dres) unpack8(*reinterpret_cast<const u32x4_t*>(dres + (size_t)row * D + i0), o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = rstd * (g[c][j] - c1 - xh[c][j] * c2);
                    o[j] = dres ? o[j] + d : d;
                }
                *reinterpret_cast<u32x4_t*>(dx + (size_t)row * D + i0) = pack8(o);
            }
        }
    }
}

// out = (x ? x : 0) + gate[range] * y   with bf16 rounding after the product (torch bf16 elementwise semantics).
// One workgroup walks whole token rows (no per-element index division); the gate vector of the row's range is read
// as float4 and stays L1/L2 resident.
__global__ __launch_bounds__(256) void gate_residual_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                              const float* __restrict__ gate_v, const float* __restrict__ gate_t,
                                                              int64_t mod_stride, int text_len, int S, int D, int64_t rows,
                                                              bf16_t* __restrict__ out) {
    const int d8 = D >> 3;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
        const float* gp = ((s < text_len) ? gate_t : gate_v) + (size_t)b * mod_stride;
        const size_t base = (size_t)row * D;
        for (int c = threadIdx.x; c < d8; c += 256) {
            float yy[8], xx[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(y + base + (size_t)c * 8), yy);
            if (x) unpack8(*reinterpret_cast<const u32x4_t*>(x + base + (size_t)c * 8), xx);
            const float4 g0 = *reinterpret_cast<const float4*>(gp + c * 8), g1 = *reinterpret_cast<const float4*>(gp + c * 8 + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = round_bf16(g[j] * yy[j]);
                xx[j] = x ? xx[j] + p : p;
            }
            *reinterpret_cast<u32x4_t*>(out + base + (size_t)c * 8) = pack8(xx);
        }
    }
}

__device__ __forceinline__ float tanh_fast(float z) {
    // tanh z = 1 - 2/(1 + e^{2z}) with the hardware exp2 / rcp (1 ulp-class; the result is rounded to bf16 anyway).
    // e^{2z} -> inf gives 1, -> 0 gives -1: saturates cleanly.
    const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);   // 2 * log2(e)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}

// gelu_tanh(x) = 0.5 x (1 + tanh z), z = c (x + 0.044715 x^3)  ==  x * sigmoid(2z) = x / (1 + 2^(x (k1 + k2 x^2))),
//   k1 = -2 c log2(e), k2 = 0.044715 k1.  One exp2 + one rcp + 5 full-rate VALU operations per element (the tanh form needs 10):
// at 4.8 TB/s these kernels spent two thirds of their time in VALU issue.  Saturates cleanly (2^+inf -> x * 0, 2^-inf -> x).
// Two independent 16-byte chunks per thread and iteration keep twice the bytes in flight.
#define GELU_K1 (-2.302208198f)      // -2 * 0.7978845608028654 * log2(e)
#define GELU_K2 (-0.1029432396f)     // 0.044715 * K1
__device__ __forceinline__ float gelu_sig(float x, float x2) {   // sigmoid(2z)
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (GELU_K1 + GELU_K2 * x2)));
}

__global__ __launch_bounds__(256) void gelu_tanh_fwd_kernel(const bf16_t* __restrict__ u, int64_t total8, bf16_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total8; c += 2 * stride) {
        const int64_t c1 = c + stride;
        const bool two = c1 < total8;
        const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(u + (size_t)c * 8);
        u32x4_t r1 = r0;
        if (two) r1 = *reinterpret_cast<const u32x4_t*>(u + (size_t)c1 * 8);
        float a[8], b[8];
        unpack8(r0, a);
        unpack8(r1, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = a[j] * gelu_sig(a[j], a[j] * a[j]);
            b[j] = b[j] * gelu_sig(b[j], b[j] * b[j]);
        }
        *reinterpret_cast<u32x4_t*>(out + (size_t)c * 8) = pack8(a);
        if (two) *reinterpret_cast<u32x4_t*>(out + (size_t)c1 * 8) = pack8(b);
    }
}

// d/dx [x s(x)] with s = sigmoid(2z):  s + x s (1 - s) 2 z'(x),  2 z' = 2c (1 + 3*0.044715 x^2)
__global__ __launch_bounds__(256) void gelu_tanh_bwd_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dy, int64_t total8,
                                                              bf16_t* __restrict__ du) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total8; c += 2 * stride) {
        const int64_t c1 = c + stride;
        const bool two = c1 < total8;
        const u32x4_t ru0 = *reinterpret_cast<const u32x4_t*>(u + (size_t)c * 8);
        const u32x4_t rg0 = *reinterpret_cast<const u32x4_t*>(dy + (size_t)c * 8);
        u32x4_t ru1 = ru0, rg1 = rg0;
        if (two) {
            ru1 = *reinterpret_cast<const u32x4_t*>(u + (size_t)c1 * 8);
            rg1 = *reinterpret_cast<const u32x4_t*>(dy + (size_t)c1 * 8);
        }
        float a[8], g[8], b[8], h[8];
        unpack8(ru0, a); unpack8(rg0, g); unpack8(ru1, b); unpack8(rg1, h);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            {
                const float x = a[j], x2 = x * x, sg = gelu_sig(x, x2);
                const float dz2 = 1.5957691216057308f + 0.2140644488f * x2;          // 2c, 2c * 3 * 0.044715
                a[j] = g[j] * (sg + x * (sg - sg * sg) * dz2);
            }
            {
                const float x = b[j], x2 = x * x, sg = gelu_sig(x, x2);
                const float dz2 = 1.5957691216057308f + 0.2140644488f * x2;
                b[j] = h[j] * (sg + x * (sg - sg * sg) * dz2);
            }
        }
        *reinterpret_cast<u32x4_t*>(du + (size_t)c * 8) = pack8(a);
        if (two) *reinterpret_cast<u32x4_t*>(du + (size_t)c1 * 8) = pack8(b);
    }
}

static inline unsigned ew_grid(int64_t total8) {
    int64_t nb = (total8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

#define DISPATCH_NV(D, CALL)                       \
    switch (((D) + 511) / 512) {                   \
        case 1: { constexpr int NV = 1; CALL; } break; \
        case 2: { constexpr int NV = 2; CALL; } break; \
        case 3: { constexpr int NV = 3; CALL; } break; \
        case 4: { constexpr int NV = 4; CALL; } break; \
        case 5: { constexpr int NV = 5; CALL; } break; \
        case 6: { constexpr int NV = 6; CALL; } break; \
        case 7: { constexpr int NV = 7; CALL; } break; \
        case 8: { constexpr int NV = 8; CALL; } break; \
        default: return VGPA_ERR_INVALID;          \
    }

extern "C" {

// y = LayerNorm(x; w, b, eps) [* scale1p + shift per token range].  Modulation pointers may all be NULL (plain LN).
int32_t vgpa_ln_modulate_fwd(const void* x, const float* ln_w, const float* ln_b, const float* shift_v, const float* scale1p_v,
                             const float* shift_t, const float* scale1p_t, int64_t mod_stride, int64_t B, int64_t S, int64_t D,
                             int64_t text_len, float eps, void* out, float* mean, float* rstd, hipStream_t stream) {
    if (!x || !ln_w || !ln_b || !out) return VGPA_ERR_INVALID;
    if (B <= 0 || S <= 0 || D <= 0 || D % 8 != 0 || D > 4096 || text_len < 0 || text_len > S) return VGPA_ERR_INVALID;
    const bool mod = shift_v != nullptr;
    if (mod && (!scale1p_v || (text_len > 0 && (!shift_t || !scale1p_t)))) return VGPA_ERR_INVALID;
    if ((mean == nullptr) != (rstd == nullptr)) return VGPA_ERR_INVALID;
    const int64_t rows = B * S;
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    DISPATCH_NV(D, VGPA_LAUNCH((ln_modulate_fwd_kernel<NV>), grid, dim3(64 * WAVES_PER_BLOCK), 0, stream, (const bf16_t*)x, ln_w, ln_b,
                                      shift_v, scale1p_v, shift_t, scale1p_t, mod_stride, (int)text_len, (int)S, (int)D, rows, eps,
                                      (bf16_t*)out, mean, rstd));
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// dx = LN-backward(dy * scale1p * w) [+ dres].  No grads for w, b, shift, scale (frozen base / timestep-only path).
int32_t vgpa_ln_modulate_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* ln_w,
                             const float* scale1p_v, const float* scale1p_t, int64_t mod_stride, int64_t B, int64_t S, int64_t D,
                             int64_t text_len, const void* dres, void* dx, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !ln_w || !dx) return VGPA_ERR_INVALID;
    if (B <= 0 || S <= 0 || D <= 0 || D % 8 != 0 || D > 4096 || text_len < 0 || text_len > S) return VGPA_ERR_INVALID;
    if (scale1p_v && text_len > 0 && !scale1p_t) return VGPA_ERR_INVALID;
    const int64_t rows = B * S;
    dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
    DISPATCH_NV(D, VGPA_LAUNCH((ln_modulate_bwd_kernel<NV>), grid, dim3(64 * WAVES_PER_BLOCK), 0, stream, (const bf16_t*)dy,
                                      (const bf16_t*)x, mean, rstd, ln_w, scale1p_v, scale1p_t, mod_stride, (int)text_len, (int)S, (int)D, rows,
                                      (const bf16_t*)dres, (bf16_t*)dx));
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// out = x + gate[range] * y   (x may be NULL: out = gate * y, which is also the backward dy = gate * dout)
int32_t vgpa_gate_residual(const void* x, const void* y, const float* gate_v, const float* gate_t, int64_t mod_stride, int64_t B,
                           int64_t S, int64_t D, int64_t text_len, void* out, hipStream_t stream) {
    if (!y || !gate_v || !out) return VGPA_ERR_INVALID;
    if (B <= 0 || S <= 0 || D <= 0 || D % 8 != 0 || text_len < 0 || text_len > S || (text_len > 0 && !gate_t)) return VGPA_ERR_INVALID;
    const int64_t rows = B * S;
    const unsigned nb = (unsigned)(rows < 16384 ? rows : 16384);
    VGPA_LAUNCH(gate_residual_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)y, gate_v, gate_t,
                       mod_stride, (int)text_len, (int)S, (int)D, rows, (bf16_t*)out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

int32_t vgpa_gelu_tanh_fwd(const void* u, int64_t n, void* out, hipStream_t stream) {
    if (!u || !out || n <= 0 || n % 8 != 0) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(gelu_tanh_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16_t*)u, n / 8, (bf16_t*)out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

int32_t vgpa_gelu_tanh_bwd(const void* u, const void* dy, int64_t n, void* du, hipStream_t stream) {
    if (!u || !dy || !du || n <= 0 || n % 8 != 0) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(gelu_tanh_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, stream, (const bf16_t*)u, (const bf16_t*)dy, n / 8, (bf16_t*)du);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
