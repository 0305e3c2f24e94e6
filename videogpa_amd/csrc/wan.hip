// Row kernels of the Wan2.2-TI2V-5B denoiser block (wan/modules/model.py of the Wan2.2 checkout the reference imports at
// train/Wan2.2-TI2V-5B/03_train.py:43-48; not vendored: restated from its published architecture, see oracle/wan.py):
//   * the residual stream is fp32 (x + y * e with fp32 e promotes it in the first block), every matmul input is bf16;
//   * modulation e = modulation + time_projection(t) is PER TOKEN in Wan2.2 ([B, L, 6, C]).  Here it is a small table
//     [groups, n_chunk, C] (one row per distinct timestep of a sample: 2 for TI2V -- first-frame tokens at t = 0, the rest at t)
//     plus an int32 group id per token, 8 KB instead of 1.4 GB per sample at 18480 tokens;
//   * q / k go through WanRMSNorm over the FULL projection width (all heads), times a bf16 weight, then the 3-axis RoPE on
//     interleaved pairs of each 128-wide head.
// One wave per row, rows of C <= 4096 elements held in registers; all HBM-bound (algorithmic bytes in DESIGN.md section 4).
#include "common.h"

#define WAN_NV 8                 // 64 lanes x 8 elements x WAN_NV >= C
#define WAN_WAVES 4

__device__ __forceinline__ int64_t wan_row() { return (int64_t)blockIdx.x * WAN_WAVES + (threadIdx.x >> 6); }

// one row held by a wave as v[c][j] (already rounded to bf16) -> e4m3 + the row's scale, the arithmetic of quant_fp8_rows_kernel (csrc/fp8.hip)
__device__ __forceinline__ void wan_quant_row(const float (*v)[8], int D, int lane, uint8_t* __restrict__ qr, float* __restrict__ scale) {
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c)
        if ((c * 64 + lane) * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[c][j]));
        }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.f;
    if (lane == 0) *scale = sc;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fminf(fmaxf(v[c][j] / sc, -448.0f), 448.0f);
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

// y = [round_bf16](LN(x)) * w + b, then * (1 + scale[g]) + shift[g]  ->  bf16 (row stride out_ld) and / or e4m3 + per-row scale   (w, b, scale / shift optional)
// mod: table row stride `mod_stride` floats, shift at column offset 0 of `shift`, scale of `scale` (pointers into the same table)
// GR: the row is first x + yres(bf16) * gate[g] (the gated residual add that precedes this LayerNorm in the block, gate NULL = 1) and that sum is ALSO
// stored to xo (fp32): one pass of 12 B per element where wan_gate_residual + wan_ln_mod_fwd moved 16.
template <int XDT, int ODT = VGPA_DTYPE_BF16, bool GR = false>
__global__ __launch_bounds__(64 * WAN_WAVES) void wan_ln_mod_fwd_kernel(const void* __restrict__ x, const int* __restrict__ gid, const float* __restrict__ w,
                                                                          const float* __restrict__ b, const float* __restrict__ shift,
                                                                          const float* __restrict__ scale, int64_t mod_stride, int D, int64_t rows, float eps,
                                                                          int round_xhat, void* __restrict__ out, int64_t out_ld,
                                                                          uint8_t* __restrict__ q8, float* __restrict__ q8_scale,
                                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                          const bf16_t* __restrict__ yres = nullptr, const float* __restrict__ gate = nullptr,
                                                                          float* __restrict__ xo = nullptr) {
    const int64_t row = wan_row();
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float v[WAN_NV][8];
    float sum = 0.f;
    // the row's HBM loads first, in a loop of their own (common.h Raw8); the modulation / affine tables further down are L2 hits shared by every row.
    // Measured per form (tools/wan_row_bench.py, one session): the plain form 145 -> 136 us at 36 960 x 3072; the GR form, which also holds the branch output
    // and writes x', LOSES (286 -> 297 us: 174 registers, two waves per SIMD) and keeps its loads next to their use.
    constexpr bool HOIST = !GR;
    Raw8<XDT> xr[HOIST ? WAN_NV : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int c = 0; c < WAN_NV; ++c) {
            const int i0 = (c * 64 + lane) * 8;
            xr[c].load(x, (size_t)row * D + i0, i0 < D);
        }
    }
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            if constexpr (HOIST) xr[c].get(v[c]);
            else load8<XDT>(x, (size_t)row * D + i0, v[c]);
            if (GR) {
                float a[8], gt[8];
                load8<VGPA_DTYPE_BF16>(yres, (size_t)row * D + i0, a);
                if (gate) load8<VGPA_DTYPE_F32>(gate, (gid ? (size_t)gid[row] * mod_stride : 0) + i0, gt);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = v[c][j] + a[j] * (gate ? gt[j] : 1.f);       // the arithmetic of wan_gate_residual_kernel
                store8<VGPA_DTYPE_F32>(xo, (size_t)row * D + i0, v[c]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[c][j];
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c)
        if ((c * 64 + lane) * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const size_t g = gid ? (size_t)gid[row] * mod_stride : 0;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float o[8], t0[8], t1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (v[c][j] - mean) * rstd;
                if (round_xhat) o[j] = round_bf16(o[j]);
            }
            if (w) {
                load8<VGPA_DTYPE_F32>(w, (size_t)i0, t0);
                load8<VGPA_DTYPE_F32>(b, (size_t)i0, t1);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] * t0[j] + t1[j];
            }
            if (scale) {
                load8<VGPA_DTYPE_F32>(scale, g + i0, t0);
                load8<VGPA_DTYPE_F32>(shift, g + i0, t1);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] * (1.f + t0[j]) + t1[j];
            }
            if (out) store8<ODT>(out, (size_t)row * out_ld + i0, o);
            if (q8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = round_bf16(o[j]);       // the e4m3 operand is made from the bf16 value, as quant_fp8_rows does
            }
        }
    }
    if (q8) wan_quant_row(v, D, lane, q8 + (size_t)row * D, q8_scale + row);
}

// dx = [dres +] LN-backward(dy * (1 + scale[g]) * w)           fp32 out (may alias dres)
// GB: the result is also handed to the gated residual add in front of this LayerNorm: dyp(bf16, row stride ld_dyp) = dx * gate_prev[g] (NULL = 1), the
// arithmetic of wan_gate_bwd_kernel, without reading dx back.
template <int XDT, int DYDT = VGPA_DTYPE_BF16, bool GB = false>
__global__ __launch_bounds__(64 * WAN_WAVES) void wan_ln_mod_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x, const float* __restrict__ mean,
                                                                          const float* __restrict__ rstd, const int* __restrict__ gid, const float* __restrict__ w,
                                                                          const float* __restrict__ scale, int64_t mod_stride, int D, int64_t rows,
                                                                          const float* dres, float* dx, const float* __restrict__ gate_prev = nullptr,
                                                                          bf16_t* __restrict__ dyp = nullptr, int64_t ld_dyp = 0) {
    const int64_t row = wan_row();
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float mu = mean[row], rs = rstd[row];
    const size_t g = gid ? (size_t)gid[row] * mod_stride : 0;
    float gy[WAN_NV][8], xh[WAN_NV][8];
    float s1 = 0.f, s2 = 0.f;
    // (loads next to their use: hoisting both operands of the row -- common.h Raw8 -- measured SLOWER here, 359 -> 379 us at 36 960 x 3072: 178 registers)
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            load8<DYDT>(dy, (size_t)row * D + i0, gy[c]);
            load8<XDT>(x, (size_t)row * D + i0, xh[c]);
            float t0[8];
            if (scale) {
                load8<VGPA_DTYPE_F32>(scale, g + i0, t0);
#pragma unroll
                for (int j = 0; j < 8; ++j) gy[c][j] *= 1.f + t0[j];
            }
            if (w) {
                load8<VGPA_DTYPE_F32>(w, (size_t)i0, t0);
#pragma unroll
                for (int j = 0; j < 8; ++j) gy[c][j] *= t0[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[c][j] = (xh[c][j] - mu) * rs;
                s1 += gy[c][j];
                s2 += gy[c][j] * xh[c][j];
            }
        }
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float o[8];
            if (dres) load8<VGPA_DTYPE_F32>(dres, (size_t)row * D + i0, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (dres ? o[j] : 0.f) + rs * (gy[c][j] - c1 - xh[c][j] * c2);
            store8<VGPA_DTYPE_F32>(dx, (size_t)row * D + i0, o);
            if (GB) {
                float gt[8];
                if (gate_prev) load8<VGPA_DTYPE_F32>(gate_prev, g + i0, gt);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] * (gate_prev ? gt[j] : 1.f);
                store8<VGPA_DTYPE_BF16>(dyp, (size_t)row * ld_dyp + i0, o);
            }
        }
    }
}

// out(fp32) = x + y(bf16) * gate[g]           (x NULL: out = y * gate; gate NULL: 1)          out may alias x
__global__ __launch_bounds__(256) void wan_gate_residual_kernel(const float* x, const bf16_t* __restrict__ y, const int* __restrict__ gid,
                                                                  const float* __restrict__ gate, int64_t mod_stride, int D, int64_t rows, float* out) {
    const int per_row = D / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / per_row;
        const int i0 = (int)(i % per_row) * 8;
        float a[8], o[8];
        load8<VGPA_DTYPE_BF16>(y, (size_t)row * D + i0, a);
        if (x) load8<VGPA_DTYPE_F32>(x, (size_t)row * D + i0, o);
        const float* gp = gate ? gate + (gid ? (size_t)gid[row] * mod_stride : 0) + i0 : nullptr;
        float gt[8];
        if (gp) load8<VGPA_DTYPE_F32>(gp, 0, gt);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (x ? o[j] : 0.f) + a[j] * (gp ? gt[j] : 1.f);
        store8<VGPA_DTYPE_F32>(out, (size_t)row * D + i0, o);
    }
}
// dy(bf16) = dout(fp32) * gate[g]
__global__ __launch_bounds__(256) void wan_gate_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ gid, const float* __restrict__ gate,
                                                             int64_t mod_stride, int D, int64_t rows, bf16_t* __restrict__ dy, int64_t ld_dy) {
    const int per_row = D / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / per_row;
        const int i0 = (int)(i % per_row) * 8;
        float o[8];
        load8<VGPA_DTYPE_F32>(dout, (size_t)row * D + i0, o);
        const float* gp = gate ? gate + (gid ? (size_t)gid[row] * mod_stride : 0) + i0 : nullptr;
        if (gp) {
            float gt[8];
            load8<VGPA_DTYPE_F32>(gp, 0, gt);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] *= gt[j];
        }
        store8<VGPA_DTYPE_BF16>(dy, (size_t)row * ld_dy + i0, o);
    }
}
// the same product written as the e4m3 operand of the feed-forward's fp8 dX GEMM: q8 = e4m3(bf16(dout * gate[g]) / scale), one wave per row
__global__ __launch_bounds__(64 * WAN_WAVES) void wan_gate_bwd_q8_kernel(const float* __restrict__ dout, const int* __restrict__ gid, const float* __restrict__ gate,
                                                                           int64_t mod_stride, int D, int64_t rows, uint8_t* __restrict__ q8,
                                                                           float* __restrict__ q8_scale) {
    const int64_t row = wan_row();
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* gp = gate ? gate + (gid ? (size_t)gid[row] * mod_stride : 0) : nullptr;
    float v[WAN_NV][8];
    Raw8<VGPA_DTYPE_F32> dr[WAN_NV];          // the row's HBM loads first (common.h Raw8)
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) dr[c].load(dout, (size_t)row * D + (c * 64 + lane) * 8, (c * 64 + lane) * 8 < D);
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            dr[c].get(v[c]);
            float gt[8];
            if (gp) load8<VGPA_DTYPE_F32>(gp, (size_t)i0, gt);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = round_bf16(gp ? v[c][j] * gt[j] : v[c][j]);
        }
    }
    wan_quant_row(v, D, lane, q8 + (size_t)row * D, q8_scale + row);
}

// WanRMSNorm over the whole row + bf16 weight + RoPE on interleaved pairs of each head (head_dim = 2 * half, pair p of every head
// turned by the angle whose cos / sin sit at rope[(row % L) * half + p]); rope NULL: no rotation (cross-attention q / k).
//   n = bf16(u * rsqrt(mean(u^2) + eps));  y = bf16(n * w);  (a, b) -> (a cos - b sin, a sin + b cos)
__global__ __launch_bounds__(64 * WAN_WAVES) void wan_rms_rope_fwd_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ w, const float* __restrict__ rope_cos,
                                                                            const float* __restrict__ rope_sin, int L, int half, int D, int64_t rows, float eps,
                                                                            int64_t ld_u, bf16_t* __restrict__ out, int64_t ld_out, float* __restrict__ rstd_out) {
    const int64_t row = wan_row();
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    // Every load of the row is issued before anything is used.  Written as `if (i0 < D) { load; use }` per chunk, hipcc put an s_waitcnt vmcnt(0) behind
    // each load (one HBM round trip per 1 KiB chunk and wave: 3.1 TB/s at 36 960 x 3072); with the loads in a loop of their own they leave back to back.
    const u32x4_t z4 = {0u, 0u, 0u, 0u};
    u32x4_t raw[WAN_NV];
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        raw[c] = i0 < D ? *reinterpret_cast<const u32x4_t*>(u + ((size_t)row * ld_u + i0)) : z4;
    }
    const size_t tr = (size_t)(row % L) * half;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float v[8];
            unpack8(raw[c], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += v[j] * v[j];
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rs;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float v[8], wv[8], o[8];
            unpack8(raw[c], v);
            load8<VGPA_DTYPE_BF16>(w, (size_t)i0, wv);          // the weight and the table rows are L2 hits, shared by every row
            f32x4_t cs4, sn4;
            if (rope_cos) {
                const int p0 = (i0 % (2 * half)) / 2;     // first pair of these 8 elements inside its head: a multiple of 4 -> 16-byte aligned table reads
                cs4 = *reinterpret_cast<const f32x4_t*>(rope_cos + tr + p0);
                sn4 = *reinterpret_cast<const f32x4_t*>(rope_sin + tr + p0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = round_bf16(round_bf16(v[j] * rs) * wv[j]);
            if (rope_cos) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float cs = cs4[j / 2], sn = sn4[j / 2];
                    const float a = o[j], bb = o[j + 1];
                    o[j] = a * cs - bb * sn;
                    o[j + 1] = a * sn + bb * cs;
                }
            }
            store8<VGPA_DTYPE_BF16>(out, (size_t)row * ld_out + i0, o);
        }
    }
}
// du = rs * (dn - n sum(dn n) / D),  dn = w * R^T dout,  n = u * rs                       (the weight is frozen: no dw)
__global__ __launch_bounds__(64 * WAN_WAVES) void wan_rms_rope_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ u, const float* __restrict__ rstd,
                                                                            const bf16_t* __restrict__ w, const float* __restrict__ rope_cos,
                                                                            const float* __restrict__ rope_sin, int L, int half, int D, int64_t rows,
                                                                            int64_t ld_dout, int64_t ld_u, bf16_t* __restrict__ du, int64_t ld_du) {
    const int64_t row = wan_row();
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float rs = rstd[row];
    const size_t tr = (size_t)(row % L) * half;
    float dn[WAN_NV][8], n[WAN_NV][8];
    float s = 0.f;
    Raw8<VGPA_DTYPE_BF16> dr[WAN_NV], ur[WAN_NV];          // the row's HBM loads first (common.h Raw8)
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        dr[c].load(dout, (size_t)row * ld_dout + i0, i0 < D);
        ur[c].load(u, (size_t)row * ld_u + i0, i0 < D);
    }
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float wv[8];
            dr[c].get(dn[c]);
            ur[c].get(n[c]);
            load8<VGPA_DTYPE_BF16>(w, (size_t)i0, wv);
            if (rope_cos) {
                const int p0 = (i0 % (2 * half)) / 2;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float cs = rope_cos[tr + p0 + j / 2], sn = rope_sin[tr + p0 + j / 2];
                    const float a = dn[c][j], bb = dn[c][j + 1];
                    dn[c][j] = a * cs + bb * sn;
                    dn[c][j + 1] = -a * sn + bb * cs;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dn[c][j] *= wv[j];
                n[c][j] *= rs;
                s += dn[c][j] * n[c][j];
            }
        }
    }
    const float m = wave_sum(s) / (float)D;
#pragma unroll
    for (int c = 0; c < WAN_NV; ++c) {
        const int i0 = (c * 64 + lane) * 8;
        if (i0 < D) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs * (dn[c][j] - n[c][j] * m);
            store8<VGPA_DTYPE_BF16>(du, (size_t)row * ld_du + i0, o);
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------- host
static inline bool wan_dims_ok(int64_t rows, int64_t D) { return rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * WAN_NV; }
static inline dim3 wan_grid(int64_t rows) { return dim3((unsigned)((rows + WAN_WAVES - 1) / WAN_WAVES)); }
static inline dim3 wan_ew_grid(int64_t n8) { const int64_t b = (n8 + 255) / 256; return dim3((unsigned)(b < 65536 ? b : 65536)); }

extern "C" int32_t vgpa_wan_ln_mod_fwd(const void* x, int32_t x_dtype, const int32_t* gid, const float* ln_w, const float* ln_b, const float* shift,
                                       const float* scale, int64_t mod_stride, int64_t rows, int64_t D, float eps, int32_t round_xhat, void* out, int64_t out_ld,
                                       void* q8, float* q8_scale, float* mean, float* rstd, hipStream_t stream) {
    if (!x || (!out && !q8) || !wan_dims_ok(rows, D) || (ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale == nullptr) ||
        (mean == nullptr) != (rstd == nullptr) || (q8 == nullptr) != (q8_scale == nullptr) || (out && (out_ld < D || out_ld % 8)))
        return VGPA_ERR_INVALID;
    if (x_dtype == VGPA_DTYPE_F32)
        VGPA_LAUNCH((wan_ln_mod_fwd_kernel<VGPA_DTYPE_F32>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, x, gid, ln_w, ln_b, shift, scale, mod_stride, (int)D, rows,
                    eps, round_xhat, out, out_ld, (uint8_t*)q8, q8_scale, mean, rstd);
    else if (x_dtype == VGPA_DTYPE_BF16)
        VGPA_LAUNCH((wan_ln_mod_fwd_kernel<VGPA_DTYPE_BF16>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, x, gid, ln_w, ln_b, shift, scale, mod_stride, (int)D, rows,
                    eps, round_xhat, out, out_ld, (uint8_t*)q8, q8_scale, mean, rstd);
    else
        return VGPA_ERR_INVALID;
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_ln_mod_bwd(const void* dy, const void* x, int32_t x_dtype, const float* mean, const float* rstd, const int32_t* gid, const float* ln_w,
                                       const float* scale, int64_t mod_stride, int64_t rows, int64_t D, const float* dres, float* dx, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !dx || !wan_dims_ok(rows, D)) return VGPA_ERR_INVALID;
    if (x_dtype == VGPA_DTYPE_F32)
        VGPA_LAUNCH((wan_ln_mod_bwd_kernel<VGPA_DTYPE_F32>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, dy, x, mean, rstd, gid, ln_w, scale, mod_stride,
                    (int)D, rows, dres, dx);
    else if (x_dtype == VGPA_DTYPE_BF16)
        VGPA_LAUNCH((wan_ln_mod_bwd_kernel<VGPA_DTYPE_BF16>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, dy, x, mean, rstd, gid, ln_w, scale, mod_stride,
                    (int)D, rows, dres, dx);
    else
        return VGPA_ERR_INVALID;
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// The output head of WanModel runs in fp32 upstream (LN, modulation and the projection under autocast(float32)): the same row kernels with an fp32
// result / an fp32 incoming gradient, x fp32, no affine.
extern "C" int32_t vgpa_wan_ln_mod_fwd_f32(const float* x, const int32_t* gid, const float* shift, const float* scale, int64_t mod_stride, int64_t rows, int64_t D,
                                           float eps, float* out, float* mean, float* rstd, hipStream_t stream) {
    if (!x || !out || !wan_dims_ok(rows, D) || (shift == nullptr) != (scale == nullptr) || (mean == nullptr) != (rstd == nullptr)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH((wan_ln_mod_fwd_kernel<VGPA_DTYPE_F32, VGPA_DTYPE_F32>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, (const void*)x, gid, (const float*)nullptr,
                (const float*)nullptr, shift, scale, mod_stride, (int)D, rows, eps, 0, (void*)out, D, (uint8_t*)nullptr, (float*)nullptr, mean, rstd);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_ln_mod_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const int32_t* gid, const float* scale,
                                           int64_t mod_stride, int64_t rows, int64_t D, float* dx, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !dx || !wan_dims_ok(rows, D)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH((wan_ln_mod_bwd_kernel<VGPA_DTYPE_F32, VGPA_DTYPE_F32>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, (const void*)dy, (const void*)x, mean, rstd, gid,
                (const float*)nullptr, scale, mod_stride, (int)D, rows, (const float*)nullptr, dx);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// Fused forms of a block's [gated residual add -> LayerNorm] pairs (x' = x + y * gate[g]; h = LN(x') ...): see the GR / GB notes at the kernels.
extern "C" int32_t vgpa_wan_gate_ln_mod_fwd(const float* x, const void* y, const int32_t* gid, const float* gate, const float* ln_w, const float* ln_b, const float* shift,
                                            const float* scale, int64_t mod_stride, int64_t rows, int64_t D, float eps, float* xo, void* out, int64_t out_ld, void* q8,
                                            float* q8_scale, float* mean, float* rstd, hipStream_t stream) {
    if (!x || !y || !xo || (!out && !q8) || !wan_dims_ok(rows, D) || (ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale == nullptr) ||
        (mean == nullptr) != (rstd == nullptr) || (q8 == nullptr) != (q8_scale == nullptr) || (out && (out_ld < D || out_ld % 8)))
        return VGPA_ERR_INVALID;
    VGPA_LAUNCH((wan_ln_mod_fwd_kernel<VGPA_DTYPE_F32, VGPA_DTYPE_BF16, true>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, (const void*)x, gid, ln_w, ln_b, shift, scale,
                mod_stride, (int)D, rows, eps, 0, out, out_ld, (uint8_t*)q8, q8_scale, mean, rstd, (const bf16_t*)y, gate, xo);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_ln_mod_bwd_gate(const void* dy, const float* x, const float* mean, const float* rstd, const int32_t* gid, const float* ln_w, const float* scale,
                                            int64_t mod_stride, int64_t rows, int64_t D, const float* dres, float* dx, const float* gate_prev, void* dy_prev,
                                            int64_t ld_dy_prev, hipStream_t stream) {
    if (!dy || !x || !mean || !rstd || !dx || !dy_prev || !wan_dims_ok(rows, D) || ld_dy_prev < D || ld_dy_prev % 8) return VGPA_ERR_INVALID;
    VGPA_LAUNCH((wan_ln_mod_bwd_kernel<VGPA_DTYPE_F32, VGPA_DTYPE_BF16, true>), wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, dy, (const void*)x, mean, rstd, gid, ln_w,
                scale, mod_stride, (int)D, rows, dres, dx, gate_prev, (bf16_t*)dy_prev, ld_dy_prev);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_gate_residual(const float* x, const void* y, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, float* out,
                                          hipStream_t stream) {
    if (!y || !out || rows <= 0 || D <= 0 || D % 8) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(wan_gate_residual_kernel, wan_ew_grid(rows * (D / 8)), dim3(256), 0, stream, x, (const bf16_t*)y, gid, gate, mod_stride, (int)D, rows, out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_gate_bwd(const float* dout, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, void* dy,
                                     int64_t ld_dy, hipStream_t stream) {
    if (!dout || !dy || rows <= 0 || D <= 0 || D % 8 || ld_dy < D || ld_dy % 8) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(wan_gate_bwd_kernel, wan_ew_grid(rows * (D / 8)), dim3(256), 0, stream, dout, gid, gate, mod_stride, (int)D, rows, (bf16_t*)dy, ld_dy);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_gate_bwd_q8(const float* dout, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, void* q8,
                                        float* q8_scale, hipStream_t stream) {
    if (!dout || !q8 || !q8_scale || !wan_dims_ok(rows, D)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(wan_gate_bwd_q8_kernel, wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, dout, gid, gate, mod_stride, (int)D, rows, (uint8_t*)q8, q8_scale);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_rms_rope_fwd(const void* u, int64_t ld_u, const void* w, const float* rope_cos, const float* rope_sin, int64_t L, int64_t head_dim,
                                         int64_t rows, int64_t D, float eps, void* out, int64_t ld_out, float* rstd, hipStream_t stream) {
    if (!u || !w || !out || !wan_dims_ok(rows, D) || (rope_cos == nullptr) != (rope_sin == nullptr)) return VGPA_ERR_INVALID;
    if (ld_u < D || ld_u % 8 || ld_out < D || ld_out % 8) return VGPA_ERR_INVALID;
    if (rope_cos && (L <= 0 || head_dim <= 0 || head_dim % 8 || D % head_dim)) return VGPA_ERR_INVALID;
    if (rope_cos && (((uintptr_t)rope_cos | (uintptr_t)rope_sin) & 15)) return VGPA_ERR_INVALID;      // the forward reads the tables as 16-byte vectors
    VGPA_LAUNCH(wan_rms_rope_fwd_kernel, wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, (const bf16_t*)u, (const bf16_t*)w, rope_cos, rope_sin, (int)(rope_cos ? L : 1),
                (int)(rope_cos ? head_dim / 2 : 1), (int)D, rows, eps, ld_u, (bf16_t*)out, ld_out, rstd);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_wan_rms_rope_bwd(const void* dout, int64_t ld_dout, const void* u, int64_t ld_u, const float* rstd, const void* w, const float* rope_cos,
                                         const float* rope_sin, int64_t L, int64_t head_dim, int64_t rows, int64_t D, void* du, int64_t ld_du, hipStream_t stream) {
    if (!dout || !u || !rstd || !w || !du || !wan_dims_ok(rows, D) || (rope_cos == nullptr) != (rope_sin == nullptr)) return VGPA_ERR_INVALID;
    if (ld_dout < D || ld_dout % 8 || ld_u < D || ld_u % 8 || ld_du < D || ld_du % 8) return VGPA_ERR_INVALID;
    if (rope_cos && (L <= 0 || head_dim <= 0 || head_dim % 8 || D % head_dim)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(wan_rms_rope_bwd_kernel, wan_grid(rows), dim3(64 * WAN_WAVES), 0, stream, (const bf16_t*)dout, (const bf16_t*)u, rstd, (const bf16_t*)w, rope_cos, rope_sin,
                (int)(rope_cos ? L : 1), (int)(rope_cos ? head_dim / 2 : 1), (int)D, rows, ld_dout, ld_u, (bf16_t*)du, ld_du);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
