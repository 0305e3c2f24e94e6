// Gated residual add fused with the following AdaLN-Zero LayerNorm + modulation, forward and backward
// (SURVEY K4 + K9 epilogue).  In diffusers' CogVideoXBlock these are `hidden_states + gate * attn_out` followed by
// `norm2(hidden_states, ...)` (and `+ gate_ff * ff_out` followed by the NEXT block's `norm1`); reached from
// train/CogVideoX-5B/03_train.py:134-151.  Oracle: oracle/cogvideox.py::block_forward.
//
//   forward :  x' = x + bf16(gate[range] * y)            (bf16, the residual stream)
//              n  = LayerNorm(x') * (w (1+scale)) + (b (1+scale) + shift)
//   backward:  dx' = dres + LN'(dn)                      (dres = gradient reaching x' through the residual path)
//              dy  = bf16(gate[range] * dx')
//
// HBM-bound.  Fusing saves one full read of the residual stream in the forward and three full passes in the backward
// (LN-backward output, the autograd add, the gate multiply input).  One wave per token row, row held in registers;
// the three per-column fp32 parameter vectors (gate, alpha, beta) of the workgroup's (batch, text|video) range live in
// LDS, so a lane needs no parameter registers.  Workgroups are aligned to the (batch, text | video) ranges -- blockIdx maps to
// (batch, range, 32-row block inside the range) -- so every row of a workgroup uses the LDS copy: the earlier layout (blocks of
// 32 consecutive rows, a global-memory slow path for rows of another range) cost the forward 58 VGPRs for a path that 2 of 1111
// workgroups ever took (166 -> 110 VGPRs, 3 -> 4 waves per SIMD).
#include "common.h"

#define RL_WAVES 4
#define RL_ROWS_PER_WAVE 8
#define RL_ROWS (RL_WAVES * RL_ROWS_PER_WAVE)

struct RLParams {
    const float* ln_w; const float* ln_b;
    const float* shift_v; const float* scale1p_v; const float* shift_t; const float* scale1p_t; int64_t mod_stride;
    const float* gate_v; const float* gate_t; int64_t gate_stride;
};

__device__ __forceinline__ void ld8(const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// alpha / beta / gate for 8 columns starting at i0, straight from global memory (slow path and LDS fill)
__device__ __forceinline__ void params_global(const RLParams& P, int b, bool is_text, int i0, float* alpha, float* beta, float* gate) {
    float w[8], bb[8], sc[8], sh[8];
    ld8(P.ln_w + i0, w);
    if (P.ln_b) ld8(P.ln_b + i0, bb);
    const float* scp = is_text ? P.scale1p_t : P.scale1p_v;
    const float* shp = is_text ? P.shift_t : P.shift_v;
    if (P.scale1p_v) ld8(scp + (size_t)b * P.mod_stride + i0, sc);
    if (P.shift_v) ld8(shp + (size_t)b * P.mod_stride + i0, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float s = P.scale1p_v ? sc[j] : 1.f;
        alpha[j] = w[j] * s;
        if (beta) beta[j] = (P.ln_b ? bb[j] * s : 0.f) + (P.shift_v ? sh[j] : 0.f);
    }
    if (gate && P.gate_v) ld8((is_text ? P.gate_t : P.gate_v) + (size_t)b * P.gate_stride + i0, gate);
}

// All global loads of a row are issued back to back into packed registers BEFORE anything depends on them (branches
// inside the chunk loop would otherwise serialise them behind s_waitcnt vmcnt(0)); HAS_Y / HAS_DRES are compile-time and
// the home / slow-path choice is hoisted to one wave-uniform branch per row.
template <int NV, bool HAS_Y>
__global__ __launch_bounds__(64 * RL_WAVES) void residual_ln_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, RLParams P,
                                                                         int text_len, int S, int D, int64_t rows, float eps,
                                                                         bf16_t* __restrict__ x_new, bf16_t* __restrict__ n_out, int64_t n_stride,
                                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    extern __shared__ __attribute__((aligned(16))) float sp[];   // [3][D]: alpha, beta, gate of the home range
    float* s_alpha = sp; float* s_beta = sp + D; float* s_gate = sp + 2 * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nT = (text_len + RL_ROWS - 1) / RL_ROWS, per = nT + (S - text_len + RL_ROWS - 1) / RL_ROWS;
    const int hb = (int)blockIdx.x / per, rblk = (int)blockIdx.x % per;
    const bool htext = rblk < nT;
    const int tok0 = htext ? rblk * RL_ROWS : text_len + (rblk - nT) * RL_ROWS, tok_end = htext ? text_len : S;
    for (int c = threadIdx.x; c * 8 < D; c += 64 * RL_WAVES) {
        float a[8], be[8], g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        params_global(P, hb, htext, c * 8, a, be, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s_alpha[c * 8 + j] = a[j]; s_beta[c * 8 + j] = be[j]; s_gate[c * 8 + j] = g[j]; }
    }
    __syncthreads();
    for (int rr = 0; rr < RL_ROWS_PER_WAVE; ++rr) {
        const int tok = tok0 + wave * RL_ROWS_PER_WAVE + rr;
        if (tok >= tok_end) return;
        const int64_t row = (int64_t)hb * S + tok;
        u32x4_t xp[NV], yp[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            const size_t off = (size_t)row * D + (i0 < D ? i0 : 0);
            xp[c] = *reinterpret_cast<const u32x4_t*>(x + off);
            if (HAS_Y) yp[c] = *reinterpret_cast<const u32x4_t*>(y + off);
        }
        float v[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            unpack8(xp[c], v[c]);
            if (HAS_Y) {
                float yy[8], g[8];
                unpack8(yp[c], yy);
                if (i0 < D) ld8(s_gate + i0, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = round_bf16(v[c][j] + round_bf16(g[j] * yy[j]));
                if (i0 < D) *reinterpret_cast<u32x4_t*>(x_new + (size_t)row * D + i0) = pack8(v[c]);
            }
            if (i0 < D) {
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
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float a[8], be[8], o[8];
                ld8(s_alpha + i0, a);
                ld8(s_beta + i0, be);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * a[j] + be[j];
                *reinterpret_cast<u32x4_t*>(n_out + (size_t)row * n_stride + i0) = pack8(o);
            }
        }
    }
}

template <int NV, bool HAS_DRES, bool HAS_DY>
__global__ __launch_bounds__(64 * RL_WAVES) void residual_ln_bwd_kernel(const bf16_t* __restrict__ dn, const bf16_t* __restrict__ xn,
                                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                         RLParams P, int text_len, int S, int D, int64_t rows,
                                                                         const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                                         bf16_t* __restrict__ dy, int64_t dy_stride) {
    extern __shared__ __attribute__((aligned(16))) float sp[];   // [2][D]: alpha, gate
    float* s_alpha = sp; float* s_gate = sp + D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nT = (text_len + RL_ROWS - 1) / RL_ROWS, per = nT + (S - text_len + RL_ROWS - 1) / RL_ROWS;
    const int hb = (int)blockIdx.x / per, rblk = (int)blockIdx.x % per;
    const bool htext = rblk < nT;
    const int tok0 = htext ? rblk * RL_ROWS : text_len + (rblk - nT) * RL_ROWS, tok_end = htext ? text_len : S;
    for (int c = threadIdx.x; c * 8 < D; c += 64 * RL_WAVES) {
        float a[8], g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        params_global(P, hb, htext, c * 8, a, nullptr, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s_alpha[c * 8 + j] = a[j]; s_gate[c * 8 + j] = g[j]; }
    }
    __syncthreads();
    for (int rr = 0; rr < RL_ROWS_PER_WAVE; ++rr) {
        const int tok = tok0 + wave * RL_ROWS_PER_WAVE + rr;
        if (tok >= tok_end) return;
        const int64_t row = (int64_t)hb * S + tok;
        const float mean = mean_in[row], rstd = rstd_in[row];
        u32x4_t dnp[NV], xp[NV], rp[NV];   // packed bf16: g and xhat are recomputed in the second pass instead of held as fp32
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            const size_t off = (size_t)row * D + (i0 < D ? i0 : 0);
            dnp[c] = *reinterpret_cast<const u32x4_t*>(dn + off);
            xp[c] = *reinterpret_cast<const u32x4_t*>(xn + off);
            if (HAS_DRES) rp[c] = *reinterpret_cast<const u32x4_t*>(dres + off);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float g[8], xh[8], a[8];
                unpack8(dnp[c], g);
                unpack8(xp[c], xh);
                ld8(s_alpha + i0, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gg = g[j] * a[j];
                    s1 += gg;
                    s2 += gg * ((xh[j] - mean) * rstd);
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            if (i0 < D) {
                float g[8], xh[8], a[8], gt[8], o[8], r[8];
                unpack8(dnp[c], g);
                unpack8(xp[c], xh);
                ld8(s_alpha + i0, a);
                if (HAS_DY) ld8(s_gate + i0, gt);
                if (HAS_DRES) unpack8(rp[c], r);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = rstd * (g[j] * a[j] - c1 - ((xh[j] - mean) * rstd) * c2);
                    o[j] = round_bf16(HAS_DRES ? r[j] + d : d);
                }
                *reinterpret_cast<u32x4_t*>(dx + (size_t)row * D + i0) = pack8(o);
                if (HAS_DY) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = gt[j] * o[j];
                    *reinterpret_cast<u32x4_t*>(dy + (size_t)row * dy_stride + i0) = pack8(o);
                }
            }
        }
    }
}

#define RL_DISPATCH(D, CALL)                           \
    switch (((D) + 511) / 512) {                       \
        case 1: { constexpr int NV = 1; CALL; } break; \
        case 2: { constexpr int NV = 2; CALL; } break; \
        case 3: { constexpr int NV = 3; CALL; } break; \
        case 4: { constexpr int NV = 4; CALL; } break; \
        case 5: { constexpr int NV = 5; CALL; } break; \
        case 6: { constexpr int NV = 6; CALL; } break; \
        case 7: { constexpr int NV = 7; CALL; } break; \
        case 8: { constexpr int NV = 8; CALL; } break; \
        default: return VGPA_ERR_INVALID;              \
    }

extern "C" {

// x_new = x + gate[range]*y ; n = LN(x_new)*alpha + beta.  y (and gates, x_new) may be NULL: n = LN-modulate(x) only.
// Modulation pointers may all be NULL (plain LayerNorm).  mean/rstd: fp32 [B,S] (both or neither).
// n_stride: row stride of n_out in elements (>= D, multiple of 8): the consumer GEMM reads n as the first D columns of a wider
// [rows, D + R] buffer whose tail carries the LoRA down-projections (the adapters ride the projection as extra K).
int32_t vgpa_residual_ln_fwd(const void* x, const void* y, const float* gate_v, const float* gate_t, int64_t gate_stride, const float* ln_w,
                             const float* ln_b, const float* shift_v, const float* scale1p_v, const float* shift_t, const float* scale1p_t,
                             int64_t mod_stride, int64_t B, int64_t S, int64_t D, int64_t text_len, float eps, void* x_new, void* n_out,
                             int64_t n_stride, float* mean, float* rstd, hipStream_t stream) {
    if (!x || !ln_w || !ln_b || !n_out || n_stride < D || n_stride % 8 != 0) return VGPA_ERR_INVALID;
    if (B <= 0 || S <= 0 || D <= 0 || D % 8 != 0 || D > 4096 || text_len < 0 || text_len > S) return VGPA_ERR_INVALID;
    if (y && (!gate_v || !x_new || (text_len > 0 && !gate_t))) return VGPA_ERR_INVALID;
    if (shift_v && (!scale1p_v || (text_len > 0 && (!shift_t || !scale1p_t)))) return VGPA_ERR_INVALID;
    if ((mean == nullptr) != (rstd == nullptr)) return VGPA_ERR_INVALID;
    RLParams P = {ln_w, ln_b, shift_v, scale1p_v, shift_t, scale1p_t, mod_stride, y ? gate_v : nullptr, y ? gate_t : nullptr, gate_stride};
    const int64_t rows = B * S;
    const dim3 grid((unsigned)(B * ((text_len + RL_ROWS - 1) / RL_ROWS + (S - text_len + RL_ROWS - 1) / RL_ROWS)));   // range-aligned blocks
    const size_t shmem = (size_t)3 * D * sizeof(float);
#define FWD_ARGS grid, dim3(64 * RL_WAVES), shmem, stream, (const bf16_t*)x, (const bf16_t*)y, P, (int)text_len, (int)S, (int)D, rows, eps, \
                 (bf16_t*)x_new, (bf16_t*)n_out, n_stride, mean, rstd
    if (y) { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_fwd_kernel<NV, true>), FWD_ARGS)); }
    else { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_fwd_kernel<NV, false>), FWD_ARGS)); }
#undef FWD_ARGS
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// dx = (dres ? dres : 0) + LN-backward(dn) ; dy = gate[range]*dx (dy, gates may be NULL).  x_new is the LN input.
// dy_stride: row stride of dy in elements (>= D, multiple of 8; see n_stride above).
int32_t vgpa_residual_ln_bwd(const void* dn, const void* x_new, const float* mean, const float* rstd, const float* ln_w, const float* scale1p_v,
                             const float* scale1p_t, int64_t mod_stride, const float* gate_v, const float* gate_t, int64_t gate_stride,
                             const void* dres, int64_t B, int64_t S, int64_t D, int64_t text_len, void* dx, void* dy, int64_t dy_stride,
                             hipStream_t stream) {
    if (!dn || !x_new || !mean || !rstd || !ln_w || !dx || (dy && (dy_stride < D || dy_stride % 8 != 0))) return VGPA_ERR_INVALID;
    if (B <= 0 || S <= 0 || D <= 0 || D % 8 != 0 || D > 4096 || text_len < 0 || text_len > S) return VGPA_ERR_INVALID;
    if (dy && (!gate_v || (text_len > 0 && !gate_t))) return VGPA_ERR_INVALID;
    if (scale1p_v && text_len > 0 && !scale1p_t) return VGPA_ERR_INVALID;
    RLParams P = {ln_w, nullptr, nullptr, scale1p_v, nullptr, scale1p_t, mod_stride, dy ? gate_v : nullptr, dy ? gate_t : nullptr, gate_stride};
    const int64_t rows = B * S;
    const dim3 grid((unsigned)(B * ((text_len + RL_ROWS - 1) / RL_ROWS + (S - text_len + RL_ROWS - 1) / RL_ROWS)));   // range-aligned blocks
    const size_t shmem = (size_t)2 * D * sizeof(float);
#define BWD_ARGS grid, dim3(64 * RL_WAVES), shmem, stream, (const bf16_t*)dn, (const bf16_t*)x_new, mean, rstd, P, (int)text_len, (int)S, (int)D, \
                 rows, (const bf16_t*)dres, (bf16_t*)dx, (bf16_t*)dy, dy_stride
    if (dres && dy) { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_bwd_kernel<NV, true, true>), BWD_ARGS)); }
    else if (dres) { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_bwd_kernel<NV, true, false>), BWD_ARGS)); }
    else if (dy) { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_bwd_kernel<NV, false, true>), BWD_ARGS)); }
    else { RL_DISPATCH(D, VGPA_LAUNCH((residual_ln_bwd_kernel<NV, false, false>), BWD_ARGS)); }
#undef BWD_ARGS
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
