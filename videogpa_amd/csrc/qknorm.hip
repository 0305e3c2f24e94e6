// QK-norm (LayerNorm over head_dim = 64, eps 1e-6, affine) fused with 3D RoPE and the token-major -> attention
// layout change, forward and backward.  Replaces attn.norm_q / attn.norm_k / apply_rotary_emb in
// diffusers' CogVideoXAttnProcessor2_0 (reached from train/CogVideoX-5B/03_train.py:134-151; RoPE only when
// image_rotary_emb is given, i.e. the generate path generate/CogVideoX-5B.py:72-77); oracle:
// oracle/cogvideox.py::block_forward / apply_rotary_emb.
//
// The same kernels serve the VGGT aggregator's attention (vggt/layers/attention.py:50-72: LayerNorm(64) on q and k, eps 1e-5, then
// RotaryPositionEmbedding2D, vggt/layers/rope.py:154-188), whose rotation pairs features (i, i + 16) inside each 32-feature half
// (vertical / horizontal) instead of (2j, 2j + 1): rope_mode 1.
//
// HBM-bound: 8 lanes own one (token, head) vector of 64 (16 B per lane), statistics by 3 xor-shuffles, fp32 math,
// one bf16 rounding at the store.  RoPE pairs (2j, 2j+1) are lane-local; the half-split partner (i +- 16) sits two lanes over.  Input is read where the fused QKV GEMM
// left it ([B,S,3,H,64], any strides); output strides are free as well, so no separate permute kernel exists.
#include "common.h"

struct QStride { int64_t b, h, s; };

__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const bf16_t* __restrict__ q_in, const bf16_t* __restrict__ k_in,
                                                                bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_out, QStride si_q,
                                                                QStride si_k, QStride so_q, QStride so_k, const float* __restrict__ wq,
                                                                const float* __restrict__ bq, const float* __restrict__ wk,
                                                                const float* __restrict__ bk, const float* __restrict__ rope_cos,
                                                                const float* __restrict__ rope_sin, int text_len, int B, int H, int S,
                                                                float eps, float q_out_scale, int rope_mode) {
    const int64_t nvec = (int64_t)B * S * H;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t vid = gid >> 3;
    const int c8 = (int)(gid & 7);
    if (vid >= 2 * nvec) return;  // whole 8-lane groups exit together (2*nvec*8 is a multiple of 8)
    const int which = vid >= nvec;
    // (token, head) index split in 32-bit arithmetic (the launcher guarantees 2 * nvec < 2^31): a 64-bit divide is ~100 VALU instructions per thread.
    // Measured (tools/qknorm_bench.py, one session): backward 243 -> 238 us at the cfg2 shape, forward unchanged -- the kernels are bandwidth-bound
    uint32_t r = (uint32_t)(which ? vid - nvec : vid);
    const int h = (int)(r % (uint32_t)H); r /= (uint32_t)H;
    const int s = (int)(r % (uint32_t)S);
    const int b = (int)(r / (uint32_t)S);
    const bf16_t* ip = which ? (k_in + b * si_k.b + h * si_k.h + (int64_t)s * si_k.s) : (q_in + b * si_q.b + h * si_q.h + (int64_t)s * si_q.s);
    bf16_t* op = which ? (k_out + b * so_k.b + h * so_k.h + (int64_t)s * so_k.s) : (q_out + b * so_q.b + h * so_q.h + (int64_t)s * so_q.s);
    const float* w = which ? wk : wq;
    const float* bb = which ? bk : bq;
    float x[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(ip + c8 * 8), x);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += x[j];
    const float mean = group8_sum(sum) * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = x[j] - mean; sq += d * d; }
    const float rstd = rsqrtf(group8_sum(sq) * (1.f / 64.f) + eps);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (x[j] - mean) * rstd * w[c8 * 8 + j] + bb[c8 * 8 + j];
    if (rope_cos && rope_mode == 1) {   // half-split pairs (i, i + 16) of each 32-feature half: out = y cos + rotate_half(y) sin
        const bool on = s >= text_len;   // the partner exchange is executed by every lane of the 8-lane group
        const float* cp = rope_cos + (int64_t)(on ? s - text_len : 0) * 64 + c8 * 8;
        const float* sp = rope_sin + (int64_t)(on ? s - text_len : 0) * 64 + c8 * 8;
        const float sgn = (c8 & 2) ? 1.f : -1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float partner = __shfl_xor(y[j], 2, 64);
            if (on) y[j] = y[j] * cp[j] + sgn * partner * sp[j];
        }
    } else if (rope_cos && s >= text_len) {
        const float* cp = rope_cos + (int64_t)(s - text_len) * 64 + c8 * 8;
        const float* sp = rope_sin + (int64_t)(s - text_len) * 64 + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float a = y[j], bq2 = y[j + 1];
            y[j] = a * cp[j] - bq2 * sp[j];
            y[j + 1] = bq2 * cp[j + 1] + a * sp[j + 1];
        }
    }
    if (!which) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] *= q_out_scale;   // attention wants q * scale * log2(e): fold it here, one rounding
    }
    *reinterpret_cast<u32x4_t*>(op + c8 * 8) = pack8(y);
}

__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16_t* __restrict__ dq_out, const bf16_t* __restrict__ dk_out,
                                                                const bf16_t* __restrict__ q_in, const bf16_t* __restrict__ k_in,
                                                                bf16_t* __restrict__ dq_in, bf16_t* __restrict__ dk_in, QStride sg_q,
                                                                QStride sg_k, QStride si_q, QStride si_k, QStride sd_q, QStride sd_k,
                                                                const float* __restrict__ wq, const float* __restrict__ wk,
                                                                const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                                int text_len, int B, int H, int S, float eps, int rope_mode) {
    const int64_t nvec = (int64_t)B * S * H;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t vid = gid >> 3;
    const int c8 = (int)(gid & 7);
    if (vid >= 2 * nvec) return;
    const int which = vid >= nvec;
    // (token, head) index split in 32-bit arithmetic (the launcher guarantees 2 * nvec < 2^31): a 64-bit divide is ~100 VALU instructions per thread.
    // Measured (tools/qknorm_bench.py, one session): backward 243 -> 238 us at the cfg2 shape, forward unchanged -- the kernels are bandwidth-bound
    uint32_t r = (uint32_t)(which ? vid - nvec : vid);
    const int h = (int)(r % (uint32_t)H); r /= (uint32_t)H;
    const int s = (int)(r % (uint32_t)S);
    const int b = (int)(r / (uint32_t)S);
    const QStride sg = which ? sg_k : sg_q, si = which ? si_k : si_q, sd = which ? sd_k : sd_q;
    const bf16_t* gp = (which ? dk_out : dq_out) + b * sg.b + h * sg.h + (int64_t)s * sg.s;
    const bf16_t* ip = (which ? k_in : q_in) + b * si.b + h * si.h + (int64_t)s * si.s;
    bf16_t* op = (which ? dk_in : dq_in) + b * sd.b + h * sd.h + (int64_t)s * sd.s;
    const float* w = which ? wk : wq;
    float g[8], x[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(gp + c8 * 8), g);
    unpack8(*reinterpret_cast<const u32x4_t*>(ip + c8 * 8), x);
    if (rope_cos && rope_mode == 1) {   // transpose of the half-split rotation: dy[d] = g[d] cos[d] -+ g[partner] sin[partner]
        const bool on = s >= text_len;
        const int64_t row = (int64_t)(on ? s - text_len : 0) * 64;
        const float* cp = rope_cos + row + c8 * 8;
        const float* spp = rope_sin + row + (c8 ^ 2) * 8;
        const float sgn = (c8 & 2) ? -1.f : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float partner = __shfl_xor(g[j], 2, 64);
            if (on) g[j] = g[j] * cp[j] + sgn * partner * spp[j];
        }
    } else if (rope_cos && s >= text_len) {  // transpose of the rotation
        const float* cp = rope_cos + (int64_t)(s - text_len) * 64 + c8 * 8;
        const float* sp = rope_sin + (int64_t)(s - text_len) * 64 + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float d0 = g[j], d1 = g[j + 1];
            g[j] = d0 * cp[j] + d1 * sp[j + 1];
            g[j + 1] = d1 * cp[j + 1] - d0 * sp[j];
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += x[j];
    const float mean = group8_sum(sum) * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] -= mean; sq += x[j] * x[j]; }
    const float rstd = rsqrtf(group8_sum(sq) * (1.f / 64.f) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] *= rstd;               // xhat
        g[j] *= w[c8 * 8 + j];      // dL/dxhat
        s1 += g[j];
        s2 += g[j] * x[j];
    }
    const float c1 = group8_sum(s1) * (1.f / 64.f), c2 = group8_sum(s2) * (1.f / 64.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = rstd * (g[j] - c1 - x[j] * c2);
    *reinterpret_cast<u32x4_t*>(op + c8 * 8) = pack8(g);
}

static inline bool qs_ok(const int64_t* st) { return st && (st[0] % 8 == 0) && (st[1] % 8 == 0) && (st[2] % 8 == 0); }
static inline QStride qmk(const int64_t* st) { QStride t; t.b = st[0]; t.h = st[1]; t.s = st[2]; return t; }

extern "C" {

// q_out = q_out_scale * RoPE(LayerNorm_64(q_in)), k_out = RoPE(LayerNorm_64(k_in)); every tensor is a bf16 [B,H,S,64] view given by element strides
// {batch, head, token}.  rope_cos/rope_sin: fp32 [S - text_len, 64] or NULL; rope_mode 0: interleaved pairs (2j, 2j+1), tables
// pair-repeated (diffusers apply_rotary_emb, SURVEY A-2); rope_mode 1: pairs (i, i+16) inside each 32-feature half, tables = the
// [cos(y-angles) x2 | cos(x-angles) x2] rows RotaryPositionEmbedding2D builds (vggt/layers/rope.py:103-112,154-188).
int32_t vgpa_qknorm_rope_fwd(const void* q_in, const void* k_in, void* q_out, void* k_out, const int64_t* qin_strides,
                             const int64_t* kin_strides, const int64_t* qout_strides, const int64_t* kout_strides, const float* wq,
                             const float* bq, const float* wk, const float* bk, const float* rope_cos, const float* rope_sin,
                             int64_t text_len, int64_t B, int64_t H, int64_t S, int64_t head_dim, float eps, float q_out_scale,
                             int32_t rope_mode, hipStream_t stream) {
    if (!q_in || !k_in || !q_out || !k_out || !wq || !bq || !wk || !bk || head_dim != 64 || B <= 0 || H <= 0 || S <= 0) return VGPA_ERR_INVALID;
    if (!qs_ok(qin_strides) || !qs_ok(kin_strides) || !qs_ok(qout_strides) || !qs_ok(kout_strides)) return VGPA_ERR_INVALID;
    if ((rope_cos == nullptr) != (rope_sin == nullptr) || text_len < 0 || text_len > S || (rope_mode != 0 && rope_mode != 1)) return VGPA_ERR_INVALID;
    const int64_t threads = 2 * B * S * H * 8;
    if (2 * B * S * H >= ((int64_t)1 << 31) || threads / 256 >= ((int64_t)1 << 31)) return VGPA_ERR_INVALID;      // the kernels split the vector index in 32 bits
    VGPA_LAUNCH(qknorm_rope_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)q_in,
                       (const bf16_t*)k_in, (bf16_t*)q_out, (bf16_t*)k_out, qmk(qin_strides), qmk(kin_strides), qmk(qout_strides),
                       qmk(kout_strides), wq, bq, wk, bk, rope_cos, rope_sin, (int)text_len, (int)B, (int)H, (int)S, eps, q_out_scale, (int)rope_mode);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// d(q_in/k_in) from d(q_out/k_out); the pre-norm inputs are re-read (statistics recomputed, nothing saved).
int32_t vgpa_qknorm_rope_bwd(const void* dq_out, const void* dk_out, const void* q_in, const void* k_in, void* dq_in, void* dk_in,
                             const int64_t* dqout_strides, const int64_t* dkout_strides, const int64_t* qin_strides,
                             const int64_t* kin_strides, const int64_t* dqin_strides, const int64_t* dkin_strides, const float* wq,
                             const float* wk, const float* rope_cos, const float* rope_sin, int64_t text_len, int64_t B, int64_t H,
                             int64_t S, int64_t head_dim, float eps, int32_t rope_mode, hipStream_t stream) {
    if (!dq_out || !dk_out || !q_in || !k_in || !dq_in || !dk_in || !wq || !wk || head_dim != 64 || B <= 0 || H <= 0 || S <= 0) return VGPA_ERR_INVALID;
    if (!qs_ok(dqout_strides) || !qs_ok(dkout_strides) || !qs_ok(qin_strides) || !qs_ok(kin_strides) || !qs_ok(dqin_strides) || !qs_ok(dkin_strides))
        return VGPA_ERR_INVALID;
    if ((rope_cos == nullptr) != (rope_sin == nullptr) || text_len < 0 || text_len > S || (rope_mode != 0 && rope_mode != 1)) return VGPA_ERR_INVALID;
    const int64_t threads = 2 * B * S * H * 8;
    if (2 * B * S * H >= ((int64_t)1 << 31) || threads / 256 >= ((int64_t)1 << 31)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(qknorm_rope_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dq_out,
                       (const bf16_t*)dk_out, (const bf16_t*)q_in, (const bf16_t*)k_in, (bf16_t*)dq_in, (bf16_t*)dk_in, qmk(dqout_strides),
                       qmk(dkout_strides), qmk(qin_strides), qmk(kin_strides), qmk(dqin_strides), qmk(dkin_strides), wq, wk, rope_cos,
                       rope_sin, (int)text_len, (int)B, (int)H, (int)S, eps, (int)rope_mode);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
