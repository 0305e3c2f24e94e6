// Attention for head_dim 128 with separate query / key lengths: the two attention shapes of the Wan2.2-TI2V-5B denoiser
// (train/Wan2.2-TI2V-5B/03_train.py:150-163 builds it; self-attention 24 heads x 128 over the video tokens, cross-attention of the
// video tokens over the 512 text tokens), forward and backward.  Same mathematics and operand conventions as attention.hip
// (softmax in the exp2 domain, fp32 statistics, lse2 = m + log2(l) kept for the backward, delta = rowsum(dO o O)).  Two kernel
// families behind the same entry points:
//   * the w1 kernels (one wave per SIMD, LDS-DMA ring, generated main loops -- DESIGN.md sections 4.1 / 4.4) for long sweeps:
//     attn128_fwd_w1_kernel, attn128_dkv_w1_kernel, attn128_dq_w1_kernel;
//   * compiler-scheduled kernels for short sweeps (cross-attention over 512 text tokens), the forward's flagged strips and as the
//     reference implementation: 4 waves x 32 stationary rows per workgroup, the streamed operand as [64 x 128] tiles through registers
//     into a double-buffered padded LDS image (pitch 136: the 16-byte row-fragment reads are conflict-free).
//   forward :  S^T = K Q^T (q on the lanes) -> online softmax per lane -> O^T += V^T P^T
//   dQ      :  dQ^T += K^T dS^T,  dS^T = P^T o (dP^T - delta),  dP^T = V dO^T
//   dK, dV  :  key on the lanes:  S = Q K^T,  dV^T += dO^T P,  dK^T += Q^T dS          (statistics per row from an LDS tile)
// Rows past the end of a ragged tile read as zeros through the buffer descriptor; keys past Skv are masked in the forward only
// (in the backward their K / V rows are zero or their results are not stored).
#include "mfma_tiles.h"

#define D128 128
#define P128 136
#define T128 (64 * P128)   // elements of one [64 x 128] LDS tile

// consecutive workgroup ids go round-robin over the 8 XCDs: give every XCD a contiguous range of tasks (its L2 then sees one (b, h) at a time)
__device__ __forceinline__ int xcd_remap128(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ rsrc_t rsrc128(const bf16_t* base /* uniform */, uint32_t row_stride, int S) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, ((uint32_t)(S - 1) * row_stride + (uint32_t)D128) * 2u, 0x00020000);
}
// 256 threads x 4 chunks of 16 B: chunk c = tid + 256 j -> row c >> 4, column chunk c & 15.  The whole offset rides in the VGPR:
// the descriptor's range check does not see the scalar offset.
__device__ __forceinline__ void tile128_load(rsrc_t rs, uint32_t row_stride, int row0, u32x4_t (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = threadIdx.x + 256u * j;
        r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (((uint32_t)row0 + (c >> 4)) * row_stride + (c & 15u) * 8u) * 2u, 0, 0);
    }
}
__device__ __forceinline__ void tile128_store(bf16_t* lds, const u32x4_t (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = threadIdx.x + 256u * j;
        *reinterpret_cast<u32x4_t*>(lds + (c >> 4) * P128 + (c & 15u) * 8u) = r[j];
    }
}
__device__ __forceinline__ bf16x8_t frag_row128(const bf16_t* lds, int rowbase, int ks, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(lds + (rowbase + (lane & 31)) * P128 + ks * 16 + (lane >> 5) * 8);
}
__device__ __forceinline__ bf16x8_t frag_tr128(const bf16_t* lds, int rowbase, int colbase, int lane) {   // see frag_tr in mfma_tiles.h
    const int hi = lane >> 5;
    const bf16_t* p = lds + (rowbase + 4 * hi + ((lane & 15) >> 2)) * P128 + colbase + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr_t;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p));
    const bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p + 8 * P128));
    bf16x8_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo[i]; r[i + 4] = hi4[i]; }
    return r;
}
__device__ __forceinline__ void load_row_frags128(const bf16_t* base, uint32_t row_stride, int row, int S, int lane, bf16x8_t (&f)[8]) {
    int r = row + (lane & 31);
    r = r < S ? r : S - 1;
    const bf16_t* p = base + ((size_t)r * row_stride + (size_t)((lane >> 5) * 8));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) f[ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 16);
}
// store this lane's column of four transposed accumulator blocks (rows d = 32 db + acc_row(r, hi)) as bf16, scaled: eight 16-byte stores per lane
// (common.h pair_rows8: the lower lane of a pair writes rows 8 g .. 8 g + 7 of the even groups, the upper lane those of the odd groups)
__device__ __forceinline__ void store_col128(bf16_t* row_ptr, const f32x16_t (&a)[4], float scale, int hi) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            u32x2_t w[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int g = 2 * gp + e;
                w[e][0] = pack_bf16x2(a[db][4 * g] * scale, a[db][4 * g + 1] * scale);
                w[e][1] = pack_bf16x2(a[db][4 * g + 2] * scale, a[db][4 * g + 3] * scale);
            }
            *reinterpret_cast<u32x4_t*>(row_ptr + db * 32 + 8 * (2 * gp + hi)) = pair_rows8(w[0], w[1]);
        }
}
// the same, and (res_row != NULL) the eight further mantissa bits of every stored value (common.h res8) for the backward's delta
__device__ __forceinline__ void store_col128_res8(bf16_t* row_ptr, uint8_t* res_row, const f32x16_t (&a)[4], float scale, int hi) {
    if (!res_row) return store_col128(row_ptr, a, scale, hi);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            u32x2_t w[2];
            uint32_t rb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int g = 2 * gp + e;
                float x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = a[db][4 * g + i] * scale;
                w[e][0] = pack_bf16x2(x[0], x[1]);
                w[e][1] = pack_bf16x2(x[2], x[3]);
                rb[e] = res8_pack4(x, w[e]);
            }
            *reinterpret_cast<u32x4_t*>(row_ptr + db * 32 + 8 * (2 * gp + hi)) = pair_rows8(w[0], w[1]);
            *reinterpret_cast<u32x2_t*>(res_row + db * 32 + 8 * (2 * gp + hi)) = pair_rows8_dword(rb[0], rb[1]);
        }
}
#define RES_ROW(ORES, sor, b, h, q) ((ORES) ? (ORES) + ((size_t)(b) * (sor).b + (size_t)(h) * (sor).h + (size_t)(q) * (sor).s) : (uint8_t*)nullptr)

// ===================================================================================================== forward
__global__ __launch_bounds__(256, 2) void attn128_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               bf16_t* __restrict__ O, float* __restrict__ LSE2, TStride sq, TStride sk, TStride sv, TStride so,
                                                               int Sq, int Skv, int H, int n_qt, float c, const int* __restrict__ only_flagged,
                                                               uint8_t* __restrict__ ORES, TStride sor) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][T128];
    const int bh = blockIdx.x / n_qt, qt = blockIdx.x % n_qt;
    // redo pass of the w1 forward: flags are per 256-row strip (= two of this kernel's 128-row tasks)
    if (only_flagged && !only_flagged[bh * ((n_qt + 1) / 2) + (qt >> 1)]) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5, wave = threadIdx.x >> 6;
    const int q0 = qt * 128 + wave * 32;
    bf16x8_t qf[8];
    load_row_frags128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0, Sq, lane, qf);
    const rsrc_t krs = rsrc128(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, Skv);
    const rsrc_t vrs = rsrc128(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, Skv);
    f32x16_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int nt = (Skv + 63) / 64;
    u32x4_t kr[4], vr[4];
    tile128_load(krs, sk.s, 0, kr);
    tile128_load(vrs, sv.s, 0, vr);
    tile128_store(lds[0][0], kr);
    tile128_store(lds[0][1], vr);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            tile128_load(krs, sk.s, (t + 1) * 64, kr);
            tile128_load(vrs, sv.s, (t + 1) * 64, vr);
        }
        const bf16_t* Kt = lds[cur][0];
        const bf16_t* Vt = lds[cur][1];
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s[kb] = mfma32(frag_row128(Kt, 32 * kb, ks, lane), qf[ks], s[kb]);
        }
        const int krem = Skv - t * 64;   // valid keys of this tile
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float x = s[kb][i] * c;
                if (krem < 64 && 32 * kb + acc_row(i, hi) >= krem) x = -INFINITY;
                s[kb][i] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, other_half(mx));
        const float mn = fmaxf(m, mx);           // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(s[kb][i] - mn);
                s[kb][i] = p;
                ls += p;
            }
        l = l * alpha + ls;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const bf16x8_t pk = pack_frag(s[kb], 8 * cc);
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] = mfma32(frag_tr128(Vt, 32 * kb + 16 * cc, 32 * db, lane), pk, o[db]);
            }
        if (t + 1 < nt) {
            tile128_store(lds[cur ^ 1][0], kr);
            tile128_store(lds[cur ^ 1][1], vr);
        }
        __syncthreads();
    }
    l += other_half(l);
    const int q = q0 + (lane & 31);
    if (q < Sq) {
        store_col128_res8(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s), RES_ROW(ORES, sor, b, h, q), o, 1.f / l, hi);
        if (hi == 0) LSE2[(size_t)bh * Sq + q] = m + __builtin_amdgcn_logf(l);   // v_log_f32 is log2
    }
}

// ----------------------------------------------------------------------------------------------------- forward, w1 structure
// The forward on the one-wave-per-SIMD structure of attention_w1.hip at head_dim 128: 4 waves x 2 q-blocks
// = 256 query rows per workgroup, K/V tiles [64 x 128] by LDS-DMA into a 4-slot ring of 32 KiB slots, main loop from
// tools/gen_w1_asm.py::Fwd128Loop (w1_fwd128_loop.inc: pipeline, LDS image and register map in its docstring).  Softmax shift = the
// row bound M[q] = c |q| max|k| (attn128_kmax_kernel), strips that underflow / overflow / are not finite are flagged and redone by
// attn128_fwd_kernel.
#include "attn_w1.h"

#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) uint32_t u32x16_t;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;
#define W1H_TILE_BYTES 16384
#define W1H_SLOT_BYTES 32768
#define W1H_RING_BYTES (4 * W1H_SLOT_BYTES)
#define W1H_L_MIN 7.888609052210118e-31f   // 2^-100
#define W1H_L_MAX 3.3230699e35f            // 2^118: as W1_L_MAX (attention_w1.hip) -- a finite row sum next to overflowed O accumulators must flag the strip too
#define W1H_M_MAX 1024.0f      // as W1_M_MAX (attention_w1.hip): the fp32 accumulator's ulp at |M'| = 1024 is a fiftieth of the weight's bf16 rounding
#define W1H_SAMPLE_KEYS 64     // as W1_SAMPLE_KEYS / W1_SAMPLE_UP (attention_w1.hip): the shift follows a sampled lower bound of the row maximum
#define W1H_SAMPLE_UP 64.0f

__device__ __forceinline__ uint32_t w1h_swz(uint32_t r) { return ((r & 3u) << 2) | ((r >> 2) & 3u); }

__global__ __launch_bounds__(256) void attn128_kmax_kernel(const bf16_t* __restrict__ K, TStride sk, int S, int H, unsigned* __restrict__ kmax2) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)S * 16; i += (int64_t)gridDim.x * 256) {   // 16 lanes per row
        const int row = (int)(i >> 4), c16 = (int)(i & 15);
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(Kb + ((size_t)row * sk.s + c16 * 8)), f);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += f[j] * f[j];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 8, 64);
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(kmax2 + bh, __float_as_uint(mx));
}

__device__ __forceinline__ u32x16_t pack4h(const bf16x8_t& a, const bf16x8_t& b, const bf16x8_t& c, const bf16x8_t& d) {
    const u32x4_t w[4] = {__builtin_bit_cast(u32x4_t, a), __builtin_bit_cast(u32x4_t, b), __builtin_bit_cast(u32x4_t, c), __builtin_bit_cast(u32x4_t, d)};
    u32x16_t r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = w[i >> 2][i & 3];
    return r;
}

__global__ __launch_bounds__(256, 1) void attn128_fwd_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                  bf16_t* __restrict__ O, float* __restrict__ LSE2, const unsigned* __restrict__ KMAX2,
                                                                  int* __restrict__ flags, TStride sq, TStride sk, TStride sv, TStride so, int Sq, int Skv,
                                                                  int H, int n_qt, float c, uint8_t* __restrict__ ORES, TStride sor) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1H_RING_BYTES];   // slot = [K tile | V tile]
    const int vid = blockIdx.x;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * 64;

    bf16x8_t qf[2][8];
    float nmc[2];    // -M[q] / c: the srcC of the score chains (the loop multiplies by c)
    const float kmax = sqrtf(__uint_as_float(KMAX2[bh]));
#pragma unroll
    for (int j = 0; j < 2; ++j) load_row_frags128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0 + 32 * j, Sq, lane, qf[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float a = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("" ::"v"(qf[j][ks]));
            float f[8];
            unpack8(__builtin_bit_cast(u32x4_t, qf[j][ks]), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) a += f[i] * f[i];
        }
        a += other_half(a);
        nmc[j] = sqrtf(a) * kmax * 1.0009765625f;                          // b[q] = |q| max|k| (unscaled: the loop multiplies by c)
    }
    if (Skv >= 2 * W1H_SAMPLE_KEYS) {   // M'[q] = min(b, m_s + 64 log2 units), m_s = the row's maximum over 64 keys spread over the sweep (attention_w1.hip W1_SAMPLE_UP)
        const bf16_t* Ks = K + ((size_t)b * sk.b + (size_t)h * sk.h);
        const uint32_t step = (uint32_t)Skv / W1H_SAMPLE_KEYS;
        float ms[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int kb = 0; kb < W1H_SAMPLE_KEYS / 32; ++kb) {
            bf16x8_t kf[8];
            load_row_frags128(Ks, sk.s * step, 32 * kb, W1H_SAMPLE_KEYS, lane, kf);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(kf[ks]));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = mfma32(kf[ks], qf[j][ks], acc);
                float m = acc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[i]);
                ms[j] = fmaxf(ms[j], fmaxf(m, other_half(m)));
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) nmc[j] = -fminf(nmc[j], ms[j] + W1H_SAMPLE_UP / c);      // unscaled units: the loop multiplies by c
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) nmc[j] = -nmc[j];
    }
    const int nt = (Skv + 63) / 64;
    {   // the pipeline's first transposed reads hit the V tile of ring slot 3: make it finite
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(lds + 3 * W1H_SLOT_BYTES + W1H_TILE_BYTES + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    const bf16_t* Vb = V + ((size_t)b * sv.b + (size_t)h * sv.h);
    const W1Rsrc krs = w1_rsrc(Kb, ((uint32_t)(Skv - 1) * sk.s + (uint32_t)D128) * 2u);
    const W1Rsrc vrs = w1_rsrc(Vb, ((uint32_t)(Skv - 1) * sv.s + (uint32_t)D128) * 2u);
    // this wave moves pieces 4 wave .. 4 wave + 3 of a tile: piece p = rows 4p .. 4p+3; lane -> (row 4p + lane / 16, LDS chunk lane % 16)
    u32x8_t voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 4u * (uint32_t)(wave * 4 + i) + (uint32_t)(lane >> 4);
        const uint32_t cl = (uint32_t)(lane & 15) ^ w1h_swz(row);
        voff[i] = (row * sk.s + cl * 8u) * 2u;
        voff[4 + i] = (row * sv.s + cl * 8u) * 2u;
    }
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * sk.s * 2u), vstep = __builtin_amdgcn_readfirstlane(64u * sv.s * 2u);
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 4096u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // tiles 0, 1 -> ring slots 0, 1
        const uint32_t dst = wbase + (uint32_t)t * W1H_SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w1_dma(dst + 1024u * i, krs, voff[i], 0u);
            w1_dma(dst + W1H_TILE_BYTES + 1024u * i, vrs, voff[4 + i], 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] += kstep; voff[4 + i] += vstep; }
    }
    // lane-constant LDS read offsets, one set per slot pair (ds offsets are 16 bit)
    u32x16_t la[2];
    {
        const uint32_t m = lane & 31;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) la[st][ks] = st * 65536u + m * 256u + ((((uint32_t)(2 * ks) + (uint32_t)hi) ^ w1h_swz(m)) << 4);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r3 = 0; r3 < 2; ++r3) {
                    const uint32_t rr = 4u * hi + ((uint32_t)(lane & 15) >> 2) + 8u * r3;
                    const uint32_t cc = 4u * db + 2u * ((uint32_t)(lane >> 4) & 1u) + (((uint32_t)lane & 3u) >> 1);
                    la[st][8 + 2 * db + r3] = st * 65536u + rr * 256u + ((cc ^ w1h_swz(rr)) << 4) + ((uint32_t)lane & 1u) * 8u;
                }
        }
    }
    const u32x16_t q00 = pack4h(qf[0][0], qf[0][1], qf[0][2], qf[0][3]), q01 = pack4h(qf[0][4], qf[0][5], qf[0][6], qf[0][7]);
    const u32x16_t q10 = pack4h(qf[1][0], qf[1][1], qf[1][2], qf[1][3]), q11 = pack4h(qf[1][4], qf[1][5], qf[1][6], qf[1][7]);
    const uint32_t niter = (uint32_t)(nt + 1);     // one extra tile step drains the pipeline
    const uint32_t krem = (uint32_t)Skv;
    const uint32_t hi4 = 4u * (uint32_t)hi;
    const uint32_t cs = __builtin_amdgcn_readfirstlane(__float_as_uint(c));
    f32x16_t o[2][4];
    u32x8_t lv;
    uint32_t t0, t1, t2, t3;
    asm volatile(
#include "w1_fwd128_loop.inc"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "={a[0:15]}"(o[0][0]), "={a[16:31]}"(o[0][1]), "={a[32:47]}"(o[0][2]), "={a[48:63]}"(o[0][3]),
          "={a[64:79]}"(o[1][0]), "={a[80:95]}"(o[1][1]), "={a[96:111]}"(o[1][2]), "={a[112:127]}"(o[1][3]), "={v[128:135]}"(lv), "+{v[176:183]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), [krem] "s"(krem), [cs] "s"(cs),
          "{a[128:143]}"(q00), "{a[144:159]}"(q01), "{a[160:175]}"(q10), "{a[176:191]}"(q11), "{v136}"(nmc[0]), "{v137}"(nmc[1]), "{v[144:159]}"(la[0]),
          "{v[160:175]}"(la[1]), "{v184}"(hi4)
        : "memory", "scc", "vcc",
#include "w1_fwd128_clobbers.inc"
    );
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < 4; ++db) asm volatile("" : "+v"(o[j][db]));

    bool bad = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float a = (__uint_as_float(lv[4 * j]) + __uint_as_float(lv[4 * j + 1])) + (__uint_as_float(lv[4 * j + 2]) + __uint_as_float(lv[4 * j + 3]));
        const float l = a + other_half(a);
        const float M = -nmc[j] * c;
        const int q = q0 + 32 * j + (lane & 31);
        if (q < Sq) {
            float oabs = 0.f;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) oabs += fabsf(o[j][db][i]);
            bad = bad || !(l >= W1H_L_MIN && l < W1H_L_MAX) || !(M <= W1H_M_MAX) || !(oabs < INFINITY);
            store_col128_res8(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s), RES_ROW(ORES, sor, b, h, q), o[j], 1.f / l, hi);
            if (hi == 0) LSE2[(size_t)bh * Sq + q] = M + __builtin_amdgcn_logf(l);
        }
    }
    if (__any(bad) && lane == 0) flags[vid] = 1;
}

// ===================================================================================================== backward
// delta[b,h,q] = sum_d dO O : 16 lanes per row, 16 B each; with `stats` also the planes [B, H, 2, S] = {-lse2 / c, -delta} the w1 dK/dV kernel
// takes as the srcC of its score chains
__global__ __launch_bounds__(256) void attn128_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, TStride sdo, TStride so, int S, int H,
                                                              int64_t total, float* __restrict__ delta, const float* __restrict__ LSE2, float inv_c,
                                                              float* __restrict__ stats, const uint8_t* __restrict__ ORES, TStride sor) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gid >> 4;
    const int c16 = (int)(gid & 15);
    float acc = 0.f;
    int64_t bh = 0;
    int q = 0;
    if (row < total) {
        q = (int)(row % S);
        bh = row / S;
        const int h = (int)(bh % H), b = (int)(bh / H);
        float a[8], o[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h + (size_t)q * sdo.s + c16 * 8)), a);
        const u32x4_t ob = *reinterpret_cast<const u32x4_t*>(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s + c16 * 8));
        if (ORES)     // the forward's 8 further mantissa bits: delta from the output to 2^-17 ("precise delta", csrc/attention_w1.hip w1_residual4 has the why)
            unpack8_res8(ob, *reinterpret_cast<const u32x2_t*>(ORES + ((size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s + c16 * 8)), o);
        else
            unpack8(ob, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j] * o[j];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if (row < total && c16 == 0) {
        delta[row] = acc;
        if (stats) {
            stats[bh * 2 * S + q] = -LSE2[row] * inv_c;
            stats[bh * 2 * S + S + q] = -acc;
        }
    }
}

__global__ __launch_bounds__(256, 1) void attn128_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                              const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ DELTA,
                                                              bf16_t* __restrict__ dQ, TStride sq, TStride sk, TStride sv, TStride sdo, TStride sdq, int Sq,
                                                              int Skv, int H, int n_qt, float c, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][T128];
    const int bh = blockIdx.x / n_qt, qt = blockIdx.x % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5, wave = threadIdx.x >> 6;
    const int q0 = qt * 128 + wave * 32;
    bf16x8_t qf[8], dof[8];
    load_row_frags128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0, Sq, lane, qf);
    load_row_frags128(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, q0, Sq, lane, dof);
    int qc = q0 + (lane & 31);
    qc = qc < Sq ? qc : Sq - 1;
    const float lse = LSE2[(size_t)bh * Sq + qc], dl = DELTA[(size_t)bh * Sq + qc];
    const rsrc_t krs = rsrc128(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, Skv);
    const rsrc_t vrs = rsrc128(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, Skv);
    f32x16_t dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;
    const int nt = (Skv + 63) / 64;
    u32x4_t kr[4], vr[4];
    tile128_load(krs, sk.s, 0, kr);
    tile128_load(vrs, sv.s, 0, vr);
    tile128_store(lds[0][0], kr);
    tile128_store(lds[0][1], vr);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            tile128_load(krs, sk.s, (t + 1) * 64, kr);
            tile128_load(vrs, sv.s, (t + 1) * 64, vr);
        }
        const bf16_t* Kt = lds[cur][0];
        const bf16_t* Vt = lds[cur][1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s = mfma32(frag_row128(Kt, 32 * kb, ks, lane), qf[ks], s);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) dp = mfma32(frag_row128(Vt, 32 * kb, ks, lane), dof[ks], dp);
            // keys past Skv: their K rows are zero, so whatever dS they get adds nothing below
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] * c - lse) * (dp[i] - dl);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const bf16x8_t dsk = pack_frag(s, 8 * cc);
#pragma unroll
                for (int db = 0; db < 4; ++db) dq[db] = mfma32(frag_tr128(Kt, 32 * kb + 16 * cc, 32 * db, lane), dsk, dq[db]);
            }
        }
        if (t + 1 < nt) {
            tile128_store(lds[cur ^ 1][0], kr);
            tile128_store(lds[cur ^ 1][1], vr);
        }
        __syncthreads();
    }
    const int q = q0 + (lane & 31);
    if (q < Sq) store_col128(dQ + ((size_t)b * sdq.b + (size_t)h * sdq.h + (size_t)q * sdq.s), dq, scale, hi);
}

__global__ __launch_bounds__(256, 1) void attn128_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ DELTA,
                                                               bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, TStride sq, TStride sk, TStride sv,
                                                               TStride sdo, TStride sdk, TStride sdv, int Sq, int Skv, int H, int n_kt, float c, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][T128];      // [buffer][Q | dO]
    __shared__ __attribute__((aligned(16))) float stats[2][2][64];       // [buffer][lse2 | delta]
    const int bh = blockIdx.x / n_kt, kt = blockIdx.x % n_kt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5, wave = threadIdx.x >> 6;
    const int k0 = kt * 128 + wave * 32;
    bf16x8_t kf[8], vf[8];
    load_row_frags128(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, k0, Skv, lane, kf);
    load_row_frags128(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, k0, Skv, lane, vf);
    const rsrc_t qrs = rsrc128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, Sq);
    const rsrc_t drs = rsrc128(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, Sq);
    const float* lse_bh = LSE2 + (size_t)bh * Sq;
    const float* dl_bh = DELTA + (size_t)bh * Sq;
    f32x16_t dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
    const int nt = (Sq + 63) / 64;
    u32x4_t qr[4], dr[4];
    float st = 0.f;
    // rows past Sq: lse2 = +huge makes P = 0 there
    auto load_stat = [&](int row0) {
        if (threadIdx.x < 128) {
            const int r = row0 + (int)(threadIdx.x & 63);
            st = threadIdx.x < 64 ? (r < Sq ? lse_bh[r] : 1e30f) : (r < Sq ? dl_bh[r] : 0.f);
        }
    };
    auto store_stat = [&](int buf) {
        if (threadIdx.x < 128) stats[buf][threadIdx.x >> 6][threadIdx.x & 63] = st;
    };
    tile128_load(qrs, sq.s, 0, qr);
    tile128_load(drs, sdo.s, 0, dr);
    load_stat(0);
    tile128_store(lds[0][0], qr);
    tile128_store(lds[0][1], dr);
    store_stat(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            tile128_load(qrs, sq.s, (t + 1) * 64, qr);
            tile128_load(drs, sdo.s, (t + 1) * 64, dr);
            load_stat((t + 1) * 64);
        }
        const bf16_t* Qt = lds[cur][0];
        const bf16_t* Dt = lds[cur][1];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16_t s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s = mfma32(frag_row128(Qt, 32 * qb, ks, lane), kf[ks], s);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) dp = mfma32(frag_row128(Dt, 32 * qb, ks, lane), vf[ks], dp);
            f32x16_t ds;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t ls = *reinterpret_cast<const f32x4_t*>(&stats[cur][0][32 * qb + 8 * g + 4 * hi]);
                const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(&stats[cur][1][32 * qb + 8 * g + 4 * hi]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(s[4 * g + i] * c - ls[i]);
                    s[4 * g + i] = p;
                    ds[4 * g + i] = p * (dp[4 * g + i] - dl[i]);
                }
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const bf16x8_t pk = pack_frag(s, 8 * cc), dsk = pack_frag(ds, 8 * cc);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    dv[db] = mfma32(frag_tr128(Dt, 32 * qb + 16 * cc, 32 * db, lane), pk, dv[db]);
                    dk[db] = mfma32(frag_tr128(Qt, 32 * qb + 16 * cc, 32 * db, lane), dsk, dk[db]);
                }
            }
        }
        if (t + 1 < nt) {
            tile128_store(lds[cur ^ 1][0], qr);
            tile128_store(lds[cur ^ 1][1], dr);
            store_stat(cur ^ 1);
        }
        __syncthreads();
    }
    const int k = k0 + (lane & 31);
    if (k < Skv) {
        store_col128(dK + ((size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)k * sdk.s), dk, scale, hi);
        store_col128(dV + ((size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)k * sdv.s), dv, 1.f, hi);
    }
}

// ----------------------------------------------------------------------------------------------------- dK, dV, w1 structure
// One 32-key block per wave (dK^T, dV^T and the K, V fragments in AGPRs), Q | dO tiles and the statistics planes by LDS-DMA, main loop from
// tools/gen_w1_asm.py::Dkv128Loop.  LDS-bandwidth-bound by construction (one fragment read per MFMA), see the generator's docstring.
#define W1H_STAT_BYTES 1024
__global__ __launch_bounds__(256, 1) void attn128_dkv_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                  const bf16_t* __restrict__ dO, const float* __restrict__ STATS, bf16_t* __restrict__ dK,
                                                                  bf16_t* __restrict__ dV, TStride sq, TStride sk, TStride sv, TStride sdo, TStride sdk,
                                                                  TStride sdv, int Sq, int Skv, int H, int n_kt, float c, float scale) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1H_RING_BYTES + 4 * W1H_STAT_BYTES];   // slot = [Q tile | dO tile]; statistics behind the ring
    const int vid = xcd_remap128(blockIdx.x, gridDim.x);
    const int bh = vid / n_kt, kt = vid % n_kt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int k0 = (kt * 4 + wave) * 32;

    bf16x8_t kf[8], vf[8];
    load_row_frags128(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, k0, Skv, lane, kf);
    load_row_frags128(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, k0, Skv, lane, vf);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { asm volatile("" ::"v"(kf[ks])); asm volatile("" ::"v"(vf[ks])); }
    const int nt = (Sq + 63) / 64;
    {   // the pipeline's first transposed reads hit ring slot 3 (both tiles): make it finite
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4_t*>(lds + 3 * W1H_SLOT_BYTES + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const W1Rsrc qrs = w1_rsrc(Q + ((size_t)b * sq.b + (size_t)h * sq.h), ((uint32_t)(Sq - 1) * sq.s + (uint32_t)D128) * 2u);
    const W1Rsrc dors = w1_rsrc(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), ((uint32_t)(Sq - 1) * sdo.s + (uint32_t)D128) * 2u);
    const W1Rsrc strs = w1_rsrc(STATS + (int64_t)bh * 2 * Sq, (uint32_t)(2 * Sq) * 4u);
    u32x8_t voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 4u * (uint32_t)(wave * 4 + i) + (uint32_t)(lane >> 4);
        const uint32_t cl = (uint32_t)(lane & 15) ^ w1h_swz(row);
        voff[i] = (row * sq.s + cl * 8u) * 2u;
        voff[4 + i] = (row * sdo.s + cl * 8u) * 2u;
    }
    const uint32_t qstep = __builtin_amdgcn_readfirstlane(64u * sq.s * 2u), dstep = __builtin_amdgcn_readfirstlane(64u * sdo.s * 2u);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 4096u);
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)W1H_RING_BYTES + (uint32_t)wave * 256u);
    // statistics piece of this wave: lanes 0..15 fetch plane 0 (-lse2 / c) of rows 16 wave + lane, lanes 16..31 plane 1 (-delta); the upper
    // half-wave repeats the lower one (its 128 bytes of the LDS piece are never read)
    uint32_t svo = ((uint32_t)((lane >> 4) & 1) * (uint32_t)Sq + (uint32_t)(16 * wave + (lane & 15))) * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // tiles 0, 1 -> ring slots 0, 1
        const uint32_t dst = wbase + (uint32_t)t * W1H_SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w1_dma(dst + 1024u * i, qrs, voff[i], 0u);
            w1_dma(dst + W1H_TILE_BYTES + 1024u * i, dors, voff[4 + i], 0u);
        }
        w1_dma4(sbase + (uint32_t)t * W1H_STAT_BYTES, strs, svo, 0u);
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] += qstep; voff[4 + i] += dstep; }
        svo += 256u;
    }
    u32x16_t la[2];
    {
        const uint32_t m = lane & 31;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) la[st][ks] = st * 65536u + m * 256u + ((((uint32_t)(2 * ks) + (uint32_t)hi) ^ w1h_swz(m)) << 4);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r3 = 0; r3 < 2; ++r3) {
                    const uint32_t rr = 4u * hi + ((uint32_t)(lane & 15) >> 2) + 8u * r3;
                    const uint32_t cc = 4u * db + 2u * ((uint32_t)(lane >> 4) & 1u) + (((uint32_t)lane & 3u) >> 1);
                    la[st][8 + 2 * db + r3] = st * 65536u + rr * 256u + ((cc ^ w1h_swz(rr)) << 4) + ((uint32_t)lane & 1u) * 8u;
                }
        }
    }
    const uint32_t sread = lds0 + (uint32_t)W1H_RING_BYTES + 16u * (uint32_t)hi;
    const u32x16_t kf0 = pack4h(kf[0], kf[1], kf[2], kf[3]), kf1 = pack4h(kf[4], kf[5], kf[6], kf[7]);
    const u32x16_t vf0 = pack4h(vf[0], vf[1], vf[2], vf[3]), vf1 = pack4h(vf[4], vf[5], vf[6], vf[7]);
    const uint32_t niter = (uint32_t)(nt + 1);   // one extra tile step drains the pipeline
    const uint32_t cs = __builtin_amdgcn_readfirstlane(__float_as_uint(c));
    f32x16_t dk[4], dv[4];
    uint32_t t0, t1;
    asm volatile(
#include "w1_dkv128_loop.inc"
        : "=&s"(t0), "=&s"(t1), "={a[0:15]}"(dk[0]), "={a[16:31]}"(dk[1]), "={a[32:47]}"(dk[2]), "={a[48:63]}"(dk[3]), "={a[64:79]}"(dv[0]),
          "={a[80:95]}"(dv[1]), "={a[96:111]}"(dv[2]), "={a[112:127]}"(dv[3]), "+{v[160:167]}"(voff), "+{v168}"(svo)
        : [rq] "s"(qrs.w), [rdo] "s"(dors.w), [rst] "s"(strs.w), [qstep] "s"(qstep), [dstep] "s"(dstep), [wbase] "s"(wbase), [sbase] "s"(sbase),
          [niter] "s"(niter), [cs] "s"(cs), "{a[128:143]}"(kf0), "{a[144:159]}"(kf1), "{a[160:175]}"(vf0), "{a[176:191]}"(vf1), "{v[128:143]}"(la[0]),
          "{v[144:159]}"(la[1]), "{v169}"(sread)
        : "memory", "scc",
#include "w1_dkv128_clobbers.inc"
    );
#pragma unroll
    for (int db = 0; db < 4; ++db) { asm volatile("" : "+v"(dk[db])); asm volatile("" : "+v"(dv[db])); }
    const int k = k0 + (lane & 31);
    if (k < Skv) {
        store_col128(dK + ((size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)k * sdk.s), dk, scale, hi);
        store_col128(dV + ((size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)k * sdv.s), dv, 1.f, hi);
    }
}

// ----------------------------------------------------------------------------------------------------- dQ, w1 structure
// One 32-row q-block per wave (dQ^T, Q and dO fragments in AGPRs), K | V tiles by LDS-DMA, main loop from tools/gen_w1_asm.py::Dq128Loop.
__global__ __launch_bounds__(256, 1) void attn128_dq_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                 const bf16_t* __restrict__ dO, const float* __restrict__ STATS, bf16_t* __restrict__ dQ,
                                                                 TStride sq, TStride sk, TStride sv, TStride sdo, TStride sdq, int Sq, int Skv, int H, int n_qt,
                                                                 float c, float scale) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1H_RING_BYTES];   // slot = [K tile | V tile]
    const int vid = xcd_remap128(blockIdx.x, gridDim.x);
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * 32;

    bf16x8_t qf[8], dof[8];
    load_row_frags128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0, Sq, lane, qf);
    load_row_frags128(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, q0, Sq, lane, dof);
    int qc = q0 + (lane & 31);
    qc = qc < Sq ? qc : Sq - 1;
    const float nl = STATS[(size_t)bh * 2 * Sq + qc], nd = STATS[(size_t)bh * 2 * Sq + Sq + qc];    // -lse2 / c, -delta
    f32x16_t cs_t, cd_t;
#pragma unroll
    for (int i = 0; i < 16; ++i) { cs_t[i] = nl; cd_t[i] = nd; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { asm volatile("" ::"v"(qf[ks])); asm volatile("" ::"v"(dof[ks])); }
    const int nt = (Skv + 63) / 64;
    {   // the pipeline's first transposed reads hit the K tile of ring slot 3: make it finite
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(lds + 3 * W1H_SLOT_BYTES + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const W1Rsrc krs = w1_rsrc(K + ((size_t)b * sk.b + (size_t)h * sk.h), ((uint32_t)(Skv - 1) * sk.s + (uint32_t)D128) * 2u);
    const W1Rsrc vrs = w1_rsrc(V + ((size_t)b * sv.b + (size_t)h * sv.h), ((uint32_t)(Skv - 1) * sv.s + (uint32_t)D128) * 2u);
    u32x8_t voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 4u * (uint32_t)(wave * 4 + i) + (uint32_t)(lane >> 4);
        const uint32_t cl = (uint32_t)(lane & 15) ^ w1h_swz(row);
        voff[i] = (row * sk.s + cl * 8u) * 2u;
        voff[4 + i] = (row * sv.s + cl * 8u) * 2u;
    }
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * sk.s * 2u), vstep = __builtin_amdgcn_readfirstlane(64u * sv.s * 2u);
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 4096u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t dst = wbase + (uint32_t)t * W1H_SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w1_dma(dst + 1024u * i, krs, voff[i], 0u);
            w1_dma(dst + W1H_TILE_BYTES + 1024u * i, vrs, voff[4 + i], 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] += kstep; voff[4 + i] += vstep; }
    }
    u32x16_t la[2];
    {
        const uint32_t m = lane & 31;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) la[st][ks] = st * 65536u + m * 256u + ((((uint32_t)(2 * ks) + (uint32_t)hi) ^ w1h_swz(m)) << 4);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r3 = 0; r3 < 2; ++r3) {
                    const uint32_t rr = 4u * hi + ((uint32_t)(lane & 15) >> 2) + 8u * r3;
                    const uint32_t cc = 4u * db + 2u * ((uint32_t)(lane >> 4) & 1u) + (((uint32_t)lane & 3u) >> 1);
                    la[st][8 + 2 * db + r3] = st * 65536u + rr * 256u + ((cc ^ w1h_swz(rr)) << 4) + ((uint32_t)lane & 1u) * 8u;
                }
        }
    }
    const u32x16_t qf0 = pack4h(qf[0], qf[1], qf[2], qf[3]), qf1 = pack4h(qf[4], qf[5], qf[6], qf[7]);
    const u32x16_t do0 = pack4h(dof[0], dof[1], dof[2], dof[3]), do1 = pack4h(dof[4], dof[5], dof[6], dof[7]);
    const uint32_t niter = (uint32_t)(nt + 1);
    const uint32_t cs = __builtin_amdgcn_readfirstlane(__float_as_uint(c));
    f32x16_t dq[4];
    uint32_t t0, t1;
    asm volatile(
#include "w1_dq128_loop.inc"
        : "=&s"(t0), "=&s"(t1), "={a[0:15]}"(dq[0]), "={a[16:31]}"(dq[1]), "={a[32:47]}"(dq[2]), "={a[48:63]}"(dq[3]), "+{v[144:151]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), [cs] "s"(cs),
          "{a[64:79]}"(qf0), "{a[80:95]}"(qf1), "{a[96:111]}"(do0), "{a[112:127]}"(do1), "{v[80:95]}"(cs_t), "{v[96:111]}"(cd_t), "{v[112:127]}"(la[0]),
          "{v[128:143]}"(la[1])
        : "memory", "scc",
#include "w1_dq128_clobbers.inc"
    );
#pragma unroll
    for (int db = 0; db < 4; ++db) asm volatile("" : "+v"(dq[db]));
    const int q = q0 + (lane & 31);
    if (q < Sq) store_col128(dQ + ((size_t)b * sdq.b + (size_t)h * sdq.h + (size_t)q * sdq.s), dq, scale, hi);
}

// ----------------------------------------------------------------------------------------------------- dQ, w1 structure, two q-blocks per wave
// Two 32-row q-blocks per wave: every streamed K / V fragment feeds two MFMAs (tools/gen_w1_asm.py::Dq128x2Loop has the register map and the reason:
// one fragment read per MFMA costs a quarter of the matrix pipe's own energy on this part).  256 query rows per workgroup.
__global__ __launch_bounds__(256, 1) void attn128_dq_w1x2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                                   const bf16_t* __restrict__ dO, const float* __restrict__ STATS, bf16_t* __restrict__ dQ,
                                                                   TStride sq, TStride sk, TStride sv, TStride sdo, TStride sdq, int Sq, int Skv, int H, int n_qt,
                                                                   float c, float scale) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1H_RING_BYTES];   // slot = [K tile | V tile]
    const int vid = xcd_remap128(blockIdx.x, gridDim.x);
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * 64;

    bf16x8_t qf[2][8], dof[2][8];
    u32x4_t st4;        // -lse2[q_0], -lse2[q_1], -delta[q_0], -delta[q_1]   (the statistics plane holds -lse2 / c: the bf16-kernel convention)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        load_row_frags128(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0 + 32 * j, Sq, lane, qf[j]);
        load_row_frags128(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, q0 + 32 * j, Sq, lane, dof[j]);
        int qc = q0 + 32 * j + (lane & 31);
        qc = qc < Sq ? qc : Sq - 1;
        st4[j] = __float_as_uint(STATS[(size_t)bh * 2 * Sq + qc] * c);
        st4[2 + j] = __float_as_uint(STATS[(size_t)bh * 2 * Sq + Sq + qc]);
    }
    const int nt = (Skv + 63) / 64;
    {   // the pipeline's first transposed reads hit the K tile of ring slot 3: make it finite
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(lds + 3 * W1H_SLOT_BYTES + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const W1Rsrc krs = w1_rsrc(K + ((size_t)b * sk.b + (size_t)h * sk.h), ((uint32_t)(Skv - 1) * sk.s + (uint32_t)D128) * 2u);
    const W1Rsrc vrs = w1_rsrc(V + ((size_t)b * sv.b + (size_t)h * sv.h), ((uint32_t)(Skv - 1) * sv.s + (uint32_t)D128) * 2u);
    u32x8_t voff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 4u * (uint32_t)(wave * 4 + i) + (uint32_t)(lane >> 4);
        const uint32_t cl = (uint32_t)(lane & 15) ^ w1h_swz(row);
        voff[i] = (row * sk.s + cl * 8u) * 2u;
        voff[4 + i] = (row * sv.s + cl * 8u) * 2u;
    }
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * sk.s * 2u), vstep = __builtin_amdgcn_readfirstlane(64u * sv.s * 2u);
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 4096u);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t dst = wbase + (uint32_t)t * W1H_SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w1_dma(dst + 1024u * i, krs, voff[i], 0u);
            w1_dma(dst + W1H_TILE_BYTES + 1024u * i, vrs, voff[4 + i], 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { voff[i] += kstep; voff[4 + i] += vstep; }
    }
    u32x16_t la[2];
    {
        const uint32_t m = lane & 31;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) la[st][ks] = st * 65536u + m * 256u + ((((uint32_t)(2 * ks) + (uint32_t)hi) ^ w1h_swz(m)) << 4);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r3 = 0; r3 < 2; ++r3) {
                    const uint32_t rr = 4u * hi + ((uint32_t)(lane & 15) >> 2) + 8u * r3;
                    const uint32_t cc = 4u * db + 2u * ((uint32_t)(lane >> 4) & 1u) + (((uint32_t)lane & 3u) >> 1);
                    la[st][8 + 2 * db + r3] = st * 65536u + rr * 256u + ((cc ^ w1h_swz(rr)) << 4) + ((uint32_t)lane & 1u) * 8u;
                }
        }
    }
    const u32x16_t q00 = pack4h(qf[0][0], qf[0][1], qf[0][2], qf[0][3]), q01 = pack4h(qf[0][4], qf[0][5], qf[0][6], qf[0][7]);
    const u32x16_t q10 = pack4h(qf[1][0], qf[1][1], qf[1][2], qf[1][3]), q11 = pack4h(qf[1][4], qf[1][5], qf[1][6], qf[1][7]);
    const u32x16_t d00 = pack4h(dof[0][0], dof[0][1], dof[0][2], dof[0][3]), d01 = pack4h(dof[0][4], dof[0][5], dof[0][6], dof[0][7]);
    const u32x16_t d10 = pack4h(dof[1][0], dof[1][1], dof[1][2], dof[1][3]), d11 = pack4h(dof[1][4], dof[1][5], dof[1][6], dof[1][7]);
    const uint32_t niter = (uint32_t)(nt + 1);
    const uint32_t cs = __builtin_amdgcn_readfirstlane(__float_as_uint(c));
    f32x16_t dq[2][4];
    uint32_t t0, t1;
    asm volatile(
#include "w1_dq128x2_loop.inc"
        : "=&s"(t0), "=&s"(t1), "={a[0:15]}"(dq[0][0]), "={a[16:31]}"(dq[0][1]), "={a[32:47]}"(dq[0][2]), "={a[48:63]}"(dq[0][3]), "={a[64:79]}"(dq[1][0]),
          "={a[80:95]}"(dq[1][1]), "={a[96:111]}"(dq[1][2]), "={a[112:127]}"(dq[1][3]), "+{v[224:231]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), [cs] "s"(cs),
          "{a[128:143]}"(q00), "{a[144:159]}"(q01), "{a[160:175]}"(q10), "{a[176:191]}"(q11), "{a[192:207]}"(d00), "{a[208:223]}"(d01), "{a[224:239]}"(d10),
          "{a[240:255]}"(d11), "{v[192:207]}"(la[0]), "{v[208:223]}"(la[1]), "{v[232:235]}"(st4)
        : "memory", "scc",
#include "w1_dq128x2_clobbers.inc"
    );
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int db = 0; db < 4; ++db) asm volatile("" : "+v"(dq[j][db]));
        const int q = q0 + 32 * j + (lane & 31);
        if (q < Sq) store_col128(dQ + ((size_t)b * sdq.b + (size_t)h * sdq.h + (size_t)q * sdq.s), dq[j], scale, hi);
    }
}

// ===================================================================================================== host
static inline bool sok128(const int64_t* st) { return st && st[0] >= 0 && st[1] >= 0 && st[2] >= D128 && st[0] % 8 == 0 && st[1] % 8 == 0 && st[2] % 8 == 0; }
static inline bool rok128(const int64_t* st, int64_t B, int64_t H, int64_t S) { return (B - 1) * st[0] + (H - 1) * st[1] + (S - 1) * st[2] + D128 < ((int64_t)1 << 31); }
static inline TStride mk128(const int64_t* st) { TStride t; t.b = (uint32_t)st[0]; t.h = (uint32_t)st[1]; t.s = (uint32_t)st[2]; return t; }
static inline bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }
#define LOG2E_F 1.4426950408889634f

// workspace of the w1 forward: per (batch, head) max |k|^2 and one redo flag per 256-row strip
extern "C" size_t vgpa_attn128_fwd_workspace_bytes(int64_t B, int64_t H, int64_t Sq) {
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    return (size_t)(B * H) * 4 + (size_t)(B * H * ((Sq + 255) / 256)) * 4;
}

// workspace NULL (or too few keys for the pipeline to pay): the compiler-scheduled kernel
#ifndef ATTN128_W1_MIN_KEYS
#define ATTN128_W1_MIN_KEYS 1024
#endif
// the sweep length from which the one-wave-per-SIMD kernels are used (-DATTN128_W1_MIN_KEYS=... in a variant build for measurements; no environment is read)
static constexpr int64_t attn128_min_sweep() { return ATTN128_W1_MIN_KEYS; }
// o_res8 (optional: uint8 [B, H, Sq, 128] view with its own element strides; NULL = not written): eight further mantissa bits of every output value
// (common.h res8) for the backward's delta -- "precise delta", see vgpa_attn_fwd_w1_res
static inline bool res8_ok(const void* o_res8, const int64_t* st, int64_t B, int64_t H, int64_t S) {
    return !o_res8 || (st && st[0] >= 0 && st[1] >= 0 && st[2] >= D128 && st[0] % 8 == 0 && st[1] % 8 == 0 && st[2] % 8 == 0 && ((uintptr_t)o_res8 & 7) == 0 &&
                       (B - 1) * st[0] + (H - 1) * st[1] + (S - 1) * st[2] + D128 < ((int64_t)1 << 32));
}
extern "C" int32_t vgpa_attn128_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides, const int64_t* k_strides,
                                    const int64_t* v_strides, const int64_t* o_strides, void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H,
                                    int64_t Sq, int64_t Skv, float scale, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !o || !lse2 || B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || !res8_ok(o_res8, ores_strides, B, H, Sq)) return VGPA_ERR_INVALID;
    uint8_t* ores = (uint8_t*)o_res8;
    const TStride sor = o_res8 ? mk128(ores_strides) : TStride{0, 0, 0};
    if (!sok128(q_strides) || !sok128(k_strides) || !sok128(v_strides) || !sok128(o_strides) || !a16(q) || !a16(k) || !a16(v) || !a16(o)) return VGPA_ERR_INVALID;
    if (!rok128(q_strides, B, H, Sq) || !rok128(k_strides, B, H, Skv) || !rok128(v_strides, B, H, Skv) || !rok128(o_strides, B, H, Sq)) return VGPA_ERR_INVALID;
    const int64_t n_qt = (Sq + 127) / 128, tasks = B * H * n_qt;
    if (tasks >= ((int64_t)1 << 31)) return VGPA_ERR_INVALID;
    const float c = scale * LOG2E_F;
    if (workspace && Skv >= attn128_min_sweep()) {
        if (ws_bytes < vgpa_attn128_fwd_workspace_bytes(B, H, Sq) || ((uintptr_t)workspace & 3)) return VGPA_ERR_WORKSPACE;
        const int64_t n_q256 = (Sq + 255) / 256, tasks256 = B * H * n_q256;
        unsigned* kmax2 = (unsigned*)workspace;
        int* flags = (int*)workspace + B * H;
        if (hipMemsetAsync(workspace, 0, vgpa_attn128_fwd_workspace_bytes(B, H, Sq), stream) != hipSuccess) return VGPA_ERR_LAUNCH;
        VGPA_LAUNCH(attn128_kmax_kernel, dim3(16, (unsigned)(B * H)), dim3(256), 0, stream, (const bf16_t*)k, mk128(k_strides), (int)Skv, (int)H, kmax2);
        VGPA_LAUNCH(attn128_fwd_w1_kernel, dim3((unsigned)tasks256), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2,
                    (const unsigned*)kmax2, flags, mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(o_strides), (int)Sq, (int)Skv, (int)H, (int)n_q256, c,
                    ores, sor);
        // flagged strips again, with the running-max kernel (exits at once for unflagged ones)
        VGPA_LAUNCH(attn128_fwd_kernel, dim3((unsigned)tasks), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2,
                    mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(o_strides), (int)Sq, (int)Skv, (int)H, (int)n_qt, c, (const int*)flags, ores, sor);
        VGPA_CHECK_LAUNCH();
        return VGPA_OK;
    }
    // Short key sweeps (the cross-attention over the 512 text tokens) stay on the per-task kernel.  Measured in round 5: a persistent workgroup per CU with
    // ONE LDS-DMA ring of K / V tiles kept running across (batch-head, q-tile) tasks and the next task's Q fragments loaded a task ahead -- no prologue,
    // no per-tile round trip -- around the SAME compiler-scheduled loop body ran 0.540-0.554 ms against this kernel's 0.510-0.513 ms at 18 480 x 512 x 48
    // heads: two independent 4-wave workgroups per CU already hide the latencies, and the loop itself (every MFMA behind its own ds_read + lgkmcnt(0))
    // is the limit.  What would help is the generated one-wave-per-SIMD loop made persistent; the ring kernel was removed again.
    VGPA_LAUNCH(attn128_fwd_kernel, dim3((unsigned)tasks), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2,
                mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(o_strides), (int)Sq, (int)Skv, (int)H, (int)n_qt, c, (const int*)nullptr, ores, sor);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// ===================================================================================================== forward on OCP-e4m3 operands
// BASELINE configs[4] "fp8 MFMA path" (train/Wan2.2-TI2V-5B/03_train.py:189-242 calls the denoiser four times per pair; its self-attention is
// 58 % of the cfg5 step).  Both products of the forward run as v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 matrix rate):
//   prep 1  attn128_f8_amax_kernel : max |q|, |k|, |v| per (batch, head)                                   -> one power-of-two scale per tensor and head
//   prep 2  attn128_f8_quant_kernel: q8 = e4m3(q c 2^-eq) [B,H,Sq,128], k8 = e4m3(k 2^-ek) [B,H,Skv,128], v8t = e4m3(v 2^-ev) TRANSPOSED [B,H,128,Lp],
//                                     |q8 row|^2 (in the units of the scores) and max_k |k8 row|^2 for the row bound M[q] >= every score of the row
//   main    attn128_fwd_f8_kernel  : tools/gen_w1_asm.py::Fwd128F8Loop (w1_fwd128f8_loop.inc) -- operand layout, key order, per-tile power-of-two P scale
//                                     and the LDS image are documented there; this wrapper owns prologue, epilogue and the redo flags
// e4m3 is a floating-point format (4 exponent bits: 17 binades), so one scale per tensor and head keeps every element's RELATIVE error at 2^-4; the scales
// are powers of two because then they ride the instruction's E8M0 scale operands for free.  c = scale * log2(e) is folded into q8: the accumulators hold
// the scores in log2 units.  Row sums stay fp32 of the unquantised weights; lse2 = M + log2(l) as in the bf16 kernels (the backward runs on bf16 operands).
#define F8_SLOT_BYTES 16384
#define F8_RING_BYTES (4 * F8_SLOT_BYTES)
__device__ __forceinline__ int f8_exp_of(float amax) {   // the smallest e with amax * 2^-e <= 448 (e4m3's largest finite value)
    if (!(amax > 0.f) || !(amax < INFINITY)) return 0;
    int e;
    (void)frexpf(amax * (1.f / 448.f), &e);
    return e;
}
__device__ __forceinline__ uint32_t f8_pack4(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ float f8_norm2_4(uint32_t w) {   // sum of squares of the four e4m3 values of a dword
    const float a = __builtin_amdgcn_cvt_f32_fp8((int)w, 0), b = __builtin_amdgcn_cvt_f32_fp8((int)w, 1), c = __builtin_amdgcn_cvt_f32_fp8((int)w, 2),
                d = __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
    return (a * a + b * b) + (c * c + d * d);
}

// stats[bh * 4 + {0, 1, 2}] = max |q|, |k|, |v| (bit patterns of non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void attn128_f8_amax_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, TStride sq,
                                                                TStride sk, TStride sv, int Sq, int Skv, int H, unsigned* __restrict__ stats) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    float m[3] = {0.f, 0.f, 0.f};
    for (int which = 0; which < 3; ++which) {
        const bf16_t* base = which == 0 ? Q + ((size_t)b * sq.b + (size_t)h * sq.h) : which == 1 ? K + ((size_t)b * sk.b + (size_t)h * sk.h) : V + ((size_t)b * sv.b + (size_t)h * sv.h);
        const uint32_t rs = which == 0 ? sq.s : which == 1 ? sk.s : sv.s;
        const int S = which == 0 ? Sq : Skv;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)S * 16; i += (int64_t)gridDim.x * 256) {   // 16 lanes per row, 16 B each
            float f[8];
            unpack8(*reinterpret_cast<const u32x4_t*>(base + ((size_t)(i >> 4) * rs + (size_t)(i & 15) * 8)), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[which] = fmaxf(m[which], fabsf(f[j]));
        }
    }
    __shared__ float mw[3][4];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const float w = wave_max(m[which]);
        if ((threadIdx.x & 63) == 0) mw[which][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < 3) {      // one conditional atomic per workgroup and tensor (same-address atomics serialise at the memory side)
        const float w = fmaxf(fmaxf(mw[threadIdx.x][0], mw[threadIdx.x][1]), fmaxf(mw[threadIdx.x][2], mw[threadIdx.x][3]));
        unsigned* dst = stats + bh * 4 + threadIdx.x;
        if (__float_as_uint(w) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, __float_as_uint(w));
    }
}

// one workgroup = 64 tokens of one (batch, head): thread t -> token t / 4, 32 features from 32 (t % 4).  All twelve 16-byte loads of the thread (q, k, v) are
// issued before any use.  V leaves transposed: the thread's eight e4m3 dwords (4 features each) go into an LDS image [feature quad][token] (pitch 66 dwords),
// then thread (quad = t / 8, token group = t % 8) reads its 4 features x 8 tokens as four 8-byte pieces, transposes the two 4 x 4 byte blocks in registers
// (v_perm_b32) and stores 8 contiguous token bytes to each of its four v8t rows.
__global__ __launch_bounds__(256) void attn128_f8_quant_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, TStride sq,
                                                                 TStride sk, TStride sv, int Sq, int Skv, int Lp, int H, float c, const unsigned* __restrict__ stats,
                                                                 uint8_t* __restrict__ q8, uint8_t* __restrict__ k8, uint8_t* __restrict__ v8t,
                                                                 float* __restrict__ qn2, unsigned* __restrict__ kmax2, bf16_t* __restrict__ QD,
                                                                 bf16_t* __restrict__ KD, bf16_t* __restrict__ VD, TStride sqd, TStride skd, TStride svd) {
    __shared__ __attribute__((aligned(16))) uint32_t vt[32 * 66];
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int tl = (int)threadIdx.x >> 2, tok = blockIdx.x * 64 + tl, d0 = 32 * ((int)threadIdx.x & 3);
    const int eq = f8_exp_of(__uint_as_float(stats[bh * 4]) * c), ek = f8_exp_of(__uint_as_float(stats[bh * 4 + 1])), ev = f8_exp_of(__uint_as_float(stats[bh * 4 + 2]));
    const u32x4_t z4 = {0u, 0u, 0u, 0u};
    u32x4_t raw[3][4];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const bf16_t* base = which == 0 ? Q + ((size_t)b * sq.b + (size_t)h * sq.h) : which == 1 ? K + ((size_t)b * sk.b + (size_t)h * sk.h) : V + ((size_t)b * sv.b + (size_t)h * sv.h);
        const uint32_t rs = which == 0 ? sq.s : which == 1 ? sk.s : sv.s;
        const bool in = tok < (which == 0 ? Sq : Skv);
#pragma unroll
        for (int g = 0; g < 4; ++g) raw[which][g] = in ? *reinterpret_cast<const u32x4_t*>(base + ((size_t)tok * rs + (size_t)(d0 + 8 * g))) : z4;
    }
    float kn = 0.f;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        const int S = which == 0 ? Sq : Skv;
        const float mul = which == 0 ? ldexpf(c, -eq) : ldexpf(1.f, which == 1 ? -ek : -ev);
        uint32_t w[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float f[8];
            unpack8(raw[which][g], f);
            w[2 * g] = f8_pack4(f[0] * mul, f[1] * mul, f[2] * mul, f[3] * mul);
            w[2 * g + 1] = f8_pack4(f[4] * mul, f[5] * mul, f[6] * mul, f[7] * mul);
        }
        // the operands the matrix pipe will multiply, DEquantised to bf16 for the backward (vgpa_attn128_fwd_f8, q_deq / k_deq / v_deq): an e4m3 value times a
        // power of two IS a bf16 number, so all three are exact -- q_deq = q8 2^eq is q PRE-SCALED by c = scale log2(e) (round 5 wrote q8 2^eq / c, rounded to
        // bf16: a 2^-9 relative error per element that moves a score of +-100 log2 units by several per cent of a weight; vgpa_attn128_bwd_prescaled takes it as is)
        bf16_t* dq_base = which == 0 ? QD : which == 1 ? KD : VD;
        if (dq_base && tok < S) {
            const TStride sd = which == 0 ? sqd : which == 1 ? skd : svd;
            const float inv = ldexpf(1.f, which == 0 ? eq : which == 1 ? ek : ev);
            bf16_t* dst = dq_base + ((size_t)b * sd.b + (size_t)h * sd.h + (size_t)tok * sd.s + (size_t)d0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float f[8];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int x = (int)w[2 * g + i];
                    f[4 * i] = __builtin_amdgcn_cvt_f32_fp8(x, 0) * inv;
                    f[4 * i + 1] = __builtin_amdgcn_cvt_f32_fp8(x, 1) * inv;
                    f[4 * i + 2] = __builtin_amdgcn_cvt_f32_fp8(x, 2) * inv;
                    f[4 * i + 3] = __builtin_amdgcn_cvt_f32_fp8(x, 3) * inv;
                }
                *reinterpret_cast<u32x4_t*>(dst + 8 * g) = pack8(f);
            }
        }
        if (which < 2) {
            float n2 = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) n2 += f8_norm2_4(w[g]);
            n2 += __shfl_xor(n2, 1, 64);
            n2 += __shfl_xor(n2, 2, 64);
            n2 = ldexpf(n2, 2 * (which == 0 ? eq : ek));           // back in the tensor's own units (q: incl. c)
            if (tok < S) {
                uint8_t* dst = (which == 0 ? q8 : k8) + (((size_t)bh * S + tok) * 128 + d0);
                *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{w[0], w[1], w[2], w[3]};
                *reinterpret_cast<u32x4_t*>(dst + 16) = u32x4_t{w[4], w[5], w[6], w[7]};
                if (which == 0 && d0 == 0) qn2[(size_t)bh * Sq + tok] = n2;
                if (which == 1) kn = n2;
            }
        } else {   // V: feature quads into the LDS image (zeros for tokens past Skv: the padded columns of v8t must be finite)
#pragma unroll
            for (int g = 0; g < 8; ++g) vt[((d0 >> 2) + g) * 66 + tl] = w[g];
        }
    }
    // one atomic per workgroup at most, and only when it would raise the value: thousands of same-address atomics per (batch, head) serialise at the
    // memory side (they were the whole cost of this kernel: 475 us against 230 for the traffic).  A stale read only errs towards issuing the atomic.
    __shared__ float knw[4];
    kn = wave_max(kn);
    if ((threadIdx.x & 63) == 0) knw[threadIdx.x >> 6] = kn;
    __syncthreads();
    if (threadIdx.x == 0) {
        kn = fmaxf(fmaxf(knw[0], knw[1]), fmaxf(knw[2], knw[3]));
        if (kn > 0.f && __float_as_uint(kn) > __hip_atomic_load(kmax2 + bh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(kmax2 + bh, __float_as_uint(kn));
    }
    if (blockIdx.x * 64 < Lp) {
        const int dq = (int)threadIdx.x >> 3, tg = (int)threadIdx.x & 7;
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x2_t t2 = *reinterpret_cast<const u32x2_t*>(vt + dq * 66 + 8 * tg + 2 * i);
            x[2 * i] = t2[0];
            x[2 * i + 1] = t2[1];
        }
        uint32_t o[4][2];          // [feature 4 dq + r][tokens 0..3 | 4..7 of the group]
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const uint32_t x0 = x[4 * hb], x1 = x[4 * hb + 1], x2 = x[4 * hb + 2], x3 = x[4 * hb + 3];
            const uint32_t a = __builtin_amdgcn_perm(x1, x0, 0x05010400u), bq = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
            const uint32_t cq = __builtin_amdgcn_perm(x3, x2, 0x05010400u), dd = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
            o[0][hb] = __builtin_amdgcn_perm(cq, a, 0x05040100u);
            o[1][hb] = __builtin_amdgcn_perm(cq, a, 0x07060302u);
            o[2][hb] = __builtin_amdgcn_perm(dd, bq, 0x05040100u);
            o[3][hb] = __builtin_amdgcn_perm(dd, bq, 0x07060302u);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<u32x2_t*>(v8t + (((size_t)bh * 128 + 4 * dq + r) * Lp + (size_t)blockIdx.x * 64 + 8 * tg)) = u32x2_t{o[r][0], o[r][1]};
    }
}

__global__ __launch_bounds__(256, 1) void attn128_fwd_f8_kernel(const uint8_t* __restrict__ Q8, const uint8_t* __restrict__ K8, const uint8_t* __restrict__ V8T,
                                                                  const float* __restrict__ QN2, const unsigned* __restrict__ KMAX2, const unsigned* __restrict__ STATS,
                                                                  bf16_t* __restrict__ O, float* __restrict__ LSE2, int* __restrict__ flags, TStride so, int Sq, int Skv,
                                                                  int Lp, int H, int n_qt, float c, uint8_t* __restrict__ ORES, TStride sor) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[F8_RING_BYTES];   // slot = [K8 tile 64 x 128 B | V8^T tile 128 x 64 B]
    const int vid = blockIdx.x;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * 64;
    const int eq = f8_exp_of(__uint_as_float(STATS[bh * 4]) * c), ek = f8_exp_of(__uint_as_float(STATS[bh * 4 + 1])), ev = f8_exp_of(__uint_as_float(STATS[bh * 4 + 2]));

    // Q8 fragments: B operand of S^T = K8 Q8^T -- column q = lane % 32, bytes 0..15 = d 64 ks + 16 hi .., bytes 16..31 = d 64 ks + 32 + 16 hi ..
    u32x4_t qf[2][2][2];
    float nm[2];
    const float kmax = sqrtf(__uint_as_float(KMAX2[bh]));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int q = q0 + 32 * j + m;
        q = q < Sq ? q : Sq - 1;
        const uint8_t* row = Q8 + ((size_t)bh * Sq + q) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) qf[j][ks][u] = *reinterpret_cast<const u32x4_t*>(row + 64 * ks + 32 * u + 16 * hi);
        nm[j] = -(sqrtf(QN2[(size_t)bh * Sq + q]) * kmax * 1.0009765625f);      // -b[q]: a hair above |q8 row| max |k8 row| >= every score of the row
    }
    if (Skv >= 2 * W1H_SAMPLE_KEYS) {
        // Round 6: the shift follows the data here too (attention_w1.hip W1_SAMPLE_UP).  M'[q] = b[q] - n, n = floor(max(0, b - (m_s + 64))) with m_s the row's maximum over
        // 64 keys spread evenly over the sweep (8 scaled MFMAs per wave on the e4m3 operands themselves).  n is an INTEGER: every p = exp2(s - M') is the bound-shifted
        // p times 2^n exactly, the per-tile exponent x moves by n with it, so P8 = e4m3(p / 2^x) keeps the bits oracle/wan.py::_F8Attn models -- except that rows whose
        // scores lie > 100 log2 units under the bound (QK-norm gains >= 2.5: every strip) no longer underflow into the redo pass (measured 14.3 ms per launch there
        // against 4.1: profiles/r06_bench_cfg5_trained_like.json).  Flags as in the bf16 kernels: l outside [2^-100, 2^118), M' > 1024, a non-finite accumulator.
        typedef int v8i_t __attribute__((ext_vector_type(8)));
        const uint32_t step = (uint32_t)Skv / W1H_SAMPLE_KEYS;
        const int sa = 127 + ek, sb = 127 + eq;
        float ms[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int kb = 0; kb < W1H_SAMPLE_KEYS / 32; ++kb) {
            const uint8_t* krow = K8 + ((size_t)bh * Skv + (size_t)(32 * kb + m) * step) * 128;
            u32x4_t kf[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u) kf[ks][u] = *reinterpret_cast<const u32x4_t*>(krow + 64 * ks + 32 * u + 16 * hi);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const v8i_t a = {(int)kf[ks][0][0], (int)kf[ks][0][1], (int)kf[ks][0][2], (int)kf[ks][0][3], (int)kf[ks][1][0], (int)kf[ks][1][1], (int)kf[ks][1][2], (int)kf[ks][1][3]};
                    const v8i_t bq = {(int)qf[j][ks][0][0], (int)qf[j][ks][0][1], (int)qf[j][ks][0][2], (int)qf[j][ks][0][3],
                                      (int)qf[j][ks][1][0], (int)qf[j][ks][1][1], (int)qf[j][ks][1][2], (int)qf[j][ks][1][3]};
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq, acc, 0, 0, 0, sa, 0, sb);      // S^T[key][q] in log2 units: this lane holds 16 keys of column q = m
                }
                float mx = acc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = fmaxf(mx, acc[i]);
                ms[j] = fmaxf(ms[j], fmaxf(mx, other_half(mx)));
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) nm[j] += floorf(fmaxf(0.f, -nm[j] - (ms[j] + W1H_SAMPLE_UP)));      // -M' = -b + n
    }
    {   // C of the first two iterations reads the V8^T halves of ring slots 2 and 3: make them finite (P8 = 0 there)
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int sl = 2; sl < 4; ++sl)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(lds + sl * F8_SLOT_BYTES + 8192 + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const W1Rsrc krs = w1_rsrc(K8 + (size_t)bh * Skv * 128, (uint32_t)Skv * 128u);                 // rows >= Skv read as zeros
    const W1Rsrc vrs = w1_rsrc(V8T + (size_t)bh * 128 * Lp, 128u * (uint32_t)Lp);
    // this wave moves pieces 2 wave, 2 wave + 1 of both tiles.  K8 piece p = rows 8p .. 8p+7 (lane -> row 8p + lane / 8, LDS chunk lane % 8 holds the
    // row's 16-byte chunk (lane % 8) ^ ((row >> 1) & 7));  V8^T piece p = rows d 16p .. 16p+15 (lane -> d 16p + lane / 4, chunk (lane % 4) ^ ((d >> 2) & 3))
    u32x4_t voff;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t kr = 8u * (uint32_t)(2 * wave + i) + (uint32_t)(lane >> 3);
        voff[i] = kr * 128u + ((((uint32_t)lane & 7u) ^ ((kr >> 1) & 7u)) << 4);
        const uint32_t vd = 16u * (uint32_t)(2 * wave + i) + (uint32_t)(lane >> 2);
        voff[2 + i] = vd * (uint32_t)Lp + ((((uint32_t)lane & 3u) ^ ((vd >> 2) & 3u)) << 4);
    }
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * 128u), vstep = __builtin_amdgcn_readfirstlane(64u);
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 2048u);
    w1_dma(wbase, krs, voff[0], 0u);                 // tile 0 -> ring slot 0
    w1_dma(wbase + 1024u, krs, voff[1], 0u);
    w1_dma(wbase + 8192u, vrs, voff[2], 0u);
    w1_dma(wbase + 8192u + 1024u, vrs, voff[3], 0u);
    voff[0] += kstep; voff[1] += kstep; voff[2] += vstep; voff[3] += vstep;

    // lane-constant LDS read offsets: K8 row of key block b: 16 ((m >> 2) & 1) + (m & 3) + 4 (m >> 3)  (+ 32 b rows as an immediate), chunk 4 ks + 2 u + hi;
    //                                 V8^T row d = m (+ 32 db as an immediate), chunk 2 u + hi
    const uint32_t krow = 16u * (((uint32_t)m >> 2) & 1u) + ((uint32_t)m & 3u) + 4u * ((uint32_t)m >> 3);
    u32x8_t la;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) la[2 * ks + u] = krow * 128u + ((((uint32_t)(4 * ks + 2 * u) + (uint32_t)hi) ^ ((krow >> 1) & 7u)) << 4);
#pragma unroll
    for (int u = 0; u < 2; ++u) la[4 + u] = (uint32_t)m * 64u + ((((uint32_t)(2 * u) + (uint32_t)hi) ^ (((uint32_t)m >> 2) & 3u)) << 4);
    la[6] = la[7] = 0u;

    const int nt = (Skv + 63) / 64;
    const uint32_t niter = (uint32_t)(nt + 2);       // two extra steps drain the pipeline (A on zero-filled tiles with every key masked)
    const uint32_t krem = (uint32_t)Skv;
    const uint32_t hi16 = 16u * (uint32_t)hi;
    const uint32_t e8q = (uint32_t)(127 + eq), e8k = (uint32_t)(127 + ek), e8v = (uint32_t)(127 + ev);
    u32x16_t q0p, q1p;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q0p[8 * ks + 4 * u + i] = qf[0][ks][u][i];
                q1p[8 * ks + 4 * u + i] = qf[1][ks][u][i];
            }
    f32x16_t o[2][4];
    u32x2_t lv;
    uint32_t t0, t1, t2, t3;
    asm volatile(
#include "w1_fwd128f8_loop.inc"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "={a[0:15]}"(o[0][0]), "={a[16:31]}"(o[0][1]), "={a[32:47]}"(o[0][2]), "={a[48:63]}"(o[0][3]),
          "={a[64:79]}"(o[1][0]), "={a[80:95]}"(o[1][1]), "={a[96:111]}"(o[1][2]), "={a[112:127]}"(o[1][3]), "={v[204:205]}"(lv), "+{v[200:203]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), [krem] "s"(krem),
          "{a[128:143]}"(q0p), "{a[144:159]}"(q1p), "{v[192:199]}"(la), "{v220}"(e8q), "{v221}"(e8k), "{v222}"(e8v), "{v224}"(hi16), "{v225}"(nm[0]), "{v226}"(nm[1])
        : "memory", "scc", "vcc",
#include "w1_fwd128f8_clobbers.inc"
    );
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < 4; ++db) asm volatile("" : "+v"(o[j][db]));

    bool bad = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float l = __uint_as_float(lv[j]);      // the loop adds the sum over all 64 keys of a tile in both lane halves
        const float M = -nm[j];
        const int q = q0 + 32 * j + m;
        if (q < Sq) {
            float oabs = 0.f;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) oabs += fabsf(o[j][db][i]);
            bad = bad || !(l >= W1H_L_MIN && l < W1H_L_MAX) || !(M <= W1H_M_MAX) || !(oabs < INFINITY);
            store_col128_res8(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s), RES_ROW(ORES, sor, b, h, q), o[j], 1.f / l, hi);
            if (hi == 0) LSE2[(size_t)bh * Sq + q] = M + __builtin_amdgcn_logf(l);
        }
    }
    if (__any(bad) && lane == 0) flags[vid] = 1;
}

static inline size_t f8_align(size_t x) { return (x + 255) / 256 * 256; }
// workspace of vgpa_attn128_fwd_f8: [stats B H 4 | kmax2 B H | flags per 256-row strip] [|q8 row|^2] [q8] [k8] [v8t]
extern "C" size_t vgpa_attn128_fwd_f8_workspace_bytes(int64_t B, int64_t H, int64_t Sq, int64_t Skv) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0) return 0;
    const int64_t Lp = (Skv + 63) / 64 * 64;
    return f8_align((size_t)(B * H * 5 + B * H * ((Sq + 255) / 256)) * 4) + f8_align((size_t)(B * H * Sq) * 4) + f8_align((size_t)(B * H * Sq) * 128) +
           f8_align((size_t)(B * H * Skv) * 128) + f8_align((size_t)(B * H * 128 * Lp));
}
// softmax(scale q k^T) v with e4m3 matrix operands (forward only: same arguments and results as vgpa_attn128_fwd; the workspace holds the quantised
// copies and is scratch).  Strips the kernel flags (row sum near underflow, bound too large) are redone by the bf16 running-max kernel -- from the dequantised
// operands when those are asked for (so that every row's output is the softmax of what the backward gets), from q, k, v otherwise.
// q_deq / k_deq / v_deq (optional, all or none): bf16 copies of the operands the products really ran on -- hand THEM to vgpa_attn128_bwd in place of q, k, v
// and its recomputed P = exp2(c q k^T - lse2) is this forward's p / l (rows sum to one), dP is formed from the v the forward used and delta = rowsum(dO o O)
// matches both: the backward is then the straight-through gradient of THIS forward instead of the gradient of a neighbouring bf16 one.
extern "C" int32_t vgpa_attn128_fwd_f8(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides, const int64_t* k_strides,
                                       const int64_t* v_strides, const int64_t* o_strides, void* o_res8, const int64_t* ores_strides, void* q_deq, void* k_deq,
                                       void* v_deq, const int64_t* qd_strides, const int64_t* kd_strides, const int64_t* vd_strides, int64_t B, int64_t H,
                                       int64_t Sq, int64_t Skv, float scale, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !o || !lse2 || !workspace || B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || !res8_ok(o_res8, ores_strides, B, H, Sq)) return VGPA_ERR_INVALID;
    // the dequantised operands are all there or all absent; each a bf16 [B, H, S, 128] view like its source
    const bool deq = q_deq || k_deq || v_deq;
    if (deq && (!q_deq || !k_deq || !v_deq || !qd_strides || !kd_strides || !vd_strides || !sok128(qd_strides) || !sok128(kd_strides) || !sok128(vd_strides) ||
                !a16(q_deq) || !a16(k_deq) || !a16(v_deq) || !rok128(qd_strides, B, H, Sq) || !rok128(kd_strides, B, H, Skv) || !rok128(vd_strides, B, H, Skv)))
        return VGPA_ERR_INVALID;
    const TStride sqd = deq ? mk128(qd_strides) : TStride{0, 0, 0}, skd = deq ? mk128(kd_strides) : TStride{0, 0, 0}, svd = deq ? mk128(vd_strides) : TStride{0, 0, 0};
    uint8_t* ores = (uint8_t*)o_res8;
    const TStride sor = o_res8 ? mk128(ores_strides) : TStride{0, 0, 0};
    if (!sok128(q_strides) || !sok128(k_strides) || !sok128(v_strides) || !sok128(o_strides) || !a16(q) || !a16(k) || !a16(v) || !a16(o)) return VGPA_ERR_INVALID;
    if (!rok128(q_strides, B, H, Sq) || !rok128(k_strides, B, H, Skv) || !rok128(v_strides, B, H, Skv) || !rok128(o_strides, B, H, Sq)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_attn128_fwd_f8_workspace_bytes(B, H, Sq, Skv) || ((uintptr_t)workspace & 255)) return VGPA_ERR_WORKSPACE;
    const int64_t Lp = (Skv + 63) / 64 * 64, n_q256 = (Sq + 255) / 256, tasks256 = B * H * n_q256, n_qt = (Sq + 127) / 128, tasks = B * H * n_qt;
    if (tasks >= ((int64_t)1 << 31) || B * H > 65535 || Skv * 128 >= ((int64_t)1 << 31) || 128 * Lp >= ((int64_t)1 << 31)) return VGPA_ERR_INVALID;
    const float c = scale * LOG2E_F;
    char* w = (char*)workspace;
    const size_t head = f8_align((size_t)(B * H * 5 + tasks256) * 4);
    unsigned* stats = (unsigned*)w;
    unsigned* kmax2 = stats + B * H * 4;
    int* flags = (int*)(kmax2 + B * H);
    float* qn2 = (float*)(w + head);
    uint8_t* q8 = (uint8_t*)qn2 + f8_align((size_t)(B * H * Sq) * 4);
    uint8_t* k8 = q8 + f8_align((size_t)(B * H * Sq) * 128);
    uint8_t* v8t = k8 + f8_align((size_t)(B * H * Skv) * 128);
    if (hipMemsetAsync(workspace, 0, head, stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    VGPA_LAUNCH(attn128_f8_amax_kernel, dim3(32, (unsigned)(B * H)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, mk128(q_strides),
                mk128(k_strides), mk128(v_strides), (int)Sq, (int)Skv, (int)H, stats);
    const int64_t nblk = (Sq > Lp ? (Sq + 63) / 64 : Lp / 64);
    VGPA_LAUNCH(attn128_f8_quant_kernel, dim3((unsigned)nblk, (unsigned)(B * H)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                mk128(q_strides), mk128(k_strides), mk128(v_strides), (int)Sq, (int)Skv, (int)Lp, (int)H, c, (const unsigned*)stats, q8, k8, v8t, qn2, kmax2,
                (bf16_t*)q_deq, (bf16_t*)k_deq, (bf16_t*)v_deq, sqd, skd, svd);
    VGPA_LAUNCH(attn128_fwd_f8_kernel, dim3((unsigned)tasks256), dim3(256), 0, stream, (const uint8_t*)q8, (const uint8_t*)k8, (const uint8_t*)v8t, (const float*)qn2,
                (const unsigned*)kmax2, (const unsigned*)stats, (bf16_t*)o, lse2, flags, mk128(o_strides), (int)Sq, (int)Skv, (int)Lp, (int)H, (int)n_q256, c, ores, sor);
    // redo pass over the flagged strips (bf16 running-max kernel).  With dequantised operands supplied it runs on THEM: the rows it rewrites (o, lse2, res8) are then
    // the softmax of exactly the q / k / v the backward will be handed, as everywhere else (round 5 redid from the original bf16 operands, which left those rows'
    // recomputed P = exp2(c q_deq k_deq - lse2) summing to 1 +- several %: ADVICE r5)
    VGPA_LAUNCH(attn128_fwd_kernel, dim3((unsigned)tasks), dim3(256), 0, stream, (const bf16_t*)(deq ? q_deq : q), (const bf16_t*)(deq ? k_deq : k),
                (const bf16_t*)(deq ? v_deq : v), (bf16_t*)o, lse2, deq ? sqd : mk128(q_strides), deq ? skd : mk128(k_strides), deq ? svd : mk128(v_strides),
                mk128(o_strides), (int)Sq, (int)Skv, (int)H, (int)n_qt, deq ? 1.f : c, (const int*)flags, ores, sor);      // q_deq carries c already
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// -DATTN128_DQ_X2=0 (variant build) selects the one-q-block-per-wave dQ kernel (A/B measurements)
#ifndef ATTN128_DQ_X2
#define ATTN128_DQ_X2 1
#endif
static constexpr bool attn128_dq_x2() { return ATTN128_DQ_X2 != 0; }

// workspace: delta [B*H*Sq] + the statistics planes [B, H, 2, Sq] (fp32)
extern "C" size_t vgpa_attn128_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Sq) {
    if (B <= 0 || H <= 0 || Sq <= 0) return 0;
    return (size_t)(3 * B * H * Sq) * 4;
}

// dkv_mode: 0 = the compiler-scheduled dQ and dK/dV kernels, 1 = the w1 kernels, -1 = automatic (w1 dK/dV from 1024 queries on, w1 dQ from 1024 keys on)
// c: log2-domain multiplier of q.k in the score chains; dq_mul / dk_mul: what dS k and dS^T q are multiplied by on the way out
static int32_t attn128_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                                void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                                const int64_t* dv_strides, const void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H, int64_t Sq, int64_t Skv,
                                float c, float dq_mul, float dk_mul, int32_t dkv_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !o || !d_o || !lse2 || !dq || !dk || !dv || B <= 0 || H <= 0 || Sq <= 0 || Skv <= 0 || !res8_ok(o_res8, ores_strides, B, H, Sq))
        return VGPA_ERR_INVALID;
    const int64_t* qs[] = {q_strides, o_strides, do_strides, dq_strides};
    const int64_t* ks[] = {k_strides, v_strides, dk_strides, dv_strides};
    for (const int64_t* s : qs) if (!sok128(s) || !rok128(s, B, H, Sq)) return VGPA_ERR_INVALID;
    for (const int64_t* s : ks) if (!sok128(s) || !rok128(s, B, H, Skv)) return VGPA_ERR_INVALID;
    if (!a16(q) || !a16(k) || !a16(v) || !a16(o) || !a16(d_o) || !a16(dq) || !a16(dk) || !a16(dv)) return VGPA_ERR_INVALID;
    if (!workspace || ((uintptr_t)workspace & 15) || ws_bytes < vgpa_attn128_bwd_workspace_bytes(B, H, Sq)) return VGPA_ERR_WORKSPACE;
    const int64_t n_qt = (Sq + 127) / 128, n_kt = (Skv + 127) / 128, total = B * H * Sq;
    if (B * H * n_qt >= ((int64_t)1 << 31) || B * H * n_kt >= ((int64_t)1 << 31) || total * 16 >= ((int64_t)1 << 39) || 2 * Sq * 4 >= ((int64_t)1 << 31))
        return VGPA_ERR_INVALID;
    float* delta = (float*)workspace;
    float* stats = delta + total;
    const bool w1 = dkv_mode == 1 || (dkv_mode < 0 && Sq >= attn128_min_sweep());
    const bool w1q = dkv_mode == 1 || (dkv_mode < 0 && Skv >= attn128_min_sweep());     // the dQ kernel sweeps the keys
    VGPA_LAUNCH(attn128_delta_kernel, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)d_o, (const bf16_t*)o, mk128(do_strides),
                mk128(o_strides), (int)Sq, (int)H, total, delta, lse2, 1.f / c, (w1 || w1q) ? stats : (float*)nullptr, (const uint8_t*)o_res8,
                o_res8 ? mk128(ores_strides) : TStride{0, 0, 0});
    if (w1q && attn128_dq_x2()) {
        const int64_t n_q256 = (Sq + 255) / 256;
        VGPA_LAUNCH(attn128_dq_w1x2_kernel, dim3((unsigned)(B * H * n_q256)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, (const float*)stats, (bf16_t*)dq, mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(do_strides),
                    mk128(dq_strides), (int)Sq, (int)Skv, (int)H, (int)n_q256, c, dq_mul);
    } else if (w1q)
        VGPA_LAUNCH(attn128_dq_w1_kernel, dim3((unsigned)(B * H * n_qt)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, (const float*)stats, (bf16_t*)dq, mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(do_strides),
                    mk128(dq_strides), (int)Sq, (int)Skv, (int)H, (int)n_qt, c, dq_mul);
    else
        VGPA_LAUNCH(attn128_dq_kernel, dim3((unsigned)(B * H * n_qt)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o,
                    lse2, (const float*)delta, (bf16_t*)dq, mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(do_strides), mk128(dq_strides), (int)Sq,
                    (int)Skv, (int)H, (int)n_qt, c, dq_mul);
    if (w1)
        VGPA_LAUNCH(attn128_dkv_w1_kernel, dim3((unsigned)(B * H * n_kt)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, (const float*)stats, (bf16_t*)dk, (bf16_t*)dv, mk128(q_strides), mk128(k_strides), mk128(v_strides), mk128(do_strides),
                    mk128(dk_strides), mk128(dv_strides), (int)Sq, (int)Skv, (int)H, (int)n_kt, c, dk_mul);
    else
        VGPA_LAUNCH(attn128_dkv_kernel, dim3((unsigned)(B * H * n_kt)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, lse2, (const float*)delta, (bf16_t*)dk, (bf16_t*)dv, mk128(q_strides), mk128(k_strides), mk128(v_strides),
                    mk128(do_strides), mk128(dk_strides), mk128(dv_strides), (int)Sq, (int)Skv, (int)H, (int)n_kt, c, dk_mul);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" int32_t vgpa_attn128_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                                    void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                    const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                                    const int64_t* dv_strides, const void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H, int64_t Sq, int64_t Skv,
                                    float scale, int32_t dkv_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    return attn128_bwd_impl(q, k, v, o, d_o, lse2, dq, dk, dv, q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides, o_res8,
                            ores_strides, B, H, Sq, Skv, scale * LOG2E_F, scale, scale, dkv_mode, workspace, ws_bytes, stream);
}

// The backward of vgpa_attn128_fwd_f8 over ITS operands: q = that call's q_deq = q8 2^eq, i.e. the query PRE-SCALED by scale log2(e) (exact in bf16), k = k_deq,
// v = v_deq.  Scores are q.k as they stand (log2 units: bit for bit the forward's, so P = exp2(q.k - lse2) is the forward's p / l); with q'' = scale log2(e) q the
// straight-through gradients are dq = scale dS k (unchanged) and dk = ln 2 dS^T q'' (= scale dS^T q).
extern "C" int32_t vgpa_attn128_bwd_prescaled(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                                              void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                              const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                                              const int64_t* dv_strides, const void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H, int64_t Sq, int64_t Skv,
                                              float scale, int32_t dkv_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    return attn128_bwd_impl(q, k, v, o, d_o, lse2, dq, dk, dv, q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides, o_res8,
                            ores_strides, B, H, Sq, Skv, 1.f, scale, 0.6931471805599453f, dkv_mode, workspace, ws_bytes, stream);
}
