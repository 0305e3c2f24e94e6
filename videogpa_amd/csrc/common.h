// Shared device helpers for the videogpa_amd HIP kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VGPA_OK 0
#define VGPA_ERR_INVALID (-1)   // bad argument (null pointer, unsupported shape / dtype)
#define VGPA_ERR_LAUNCH (-2)    // hipGetLastError() after a launch
#define VGPA_ERR_WORKSPACE (-3) // caller-provided workspace too small

#define VGPA_DTYPE_F32 0
#define VGPA_DTYPE_BF16 1

// clear any stale (unrelated) runtime error first, so VGPA_CHECK_LAUNCH reports only this launch
#define VGPA_LAUNCH(...)                 \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

#define VGPA_CHECK_LAUNCH()                               \
    do {                                                  \
        if (hipGetLastError() != hipSuccess) return VGPA_ERR_LAUNCH; \
    } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte load/store
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// fp32 -> bf16, round-to-nearest-even (the result of torch's float -> bfloat16 cast for every non-NaN input; NaN stays NaN).
// gfx950 converts in hardware, two values per instruction (v_cvt_pk_bf16_f32): the integer-arithmetic form this replaces cost
// ~6 VALU instructions per element, which made the "HBM-bound" LN / GELU / QK-norm kernels co-limited by VALU issue.
typedef __attribute__((ext_vector_type(2))) float vgpa_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 vgpa_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const vgpa_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, vgpa_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16lo_to_f32(v[i]);
        f[2 * i + 1] = bf16hi_to_f32(v[i]);
    }
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- "res8": eight further mantissa bits of a value behind its bf16 rounding -------------------------------------------------------
// The attention forward stores its output as bf16 (the projection that follows multiplies bf16), but the backward's delta = rowsum(dO o O)
// wants the UNROUNDED output (csrc/attention_w1.hip, w1_residual4, has the why).  One byte per element carries the difference: with
// E = the biased exponent of bf16(x), ulp = 2^(E - 134), the residual x - bf16(x) lies in [-ulp / 2, ulp / 2] and is stored as
//     byte = 128 + clamp(rint((x - bf16(x)) * 2^8 / ulp), -128, 127),
// so (bf16(x), byte) together hold x to 2^-17 relative -- what a bf16 residual tensor gives (2^-18) at half its bytes.  |x| < 2^-111 stores 128.
#define VGPA_RES_NONE 0
#define VGPA_RES_BF16 1   // o_res = bf16(x - bf16(x))
#define VGPA_RES_8 2      // the byte above
__device__ __forceinline__ uint32_t res8_byte(float x, uint32_t xb /* bf16 bits of x in the low half */) {
    const uint32_t E = (xb >> 7) & 0xffu;
    const float r = x - __uint_as_float(xb << 16);
    const float t = __builtin_rintf(r * __uint_as_float((269u - E) << 23));          // 2^(142 - E) = 2^8 / ulp
    const int q = E > 15u ? (int)fminf(fmaxf(t, -128.f), 127.f) : 0;
    return (uint32_t)(q + 128);
}
// four values and their packed bf16 pairs (element 0 in the low half of packed[0]) -> four bytes, element 0 lowest
__device__ __forceinline__ uint32_t res8_pack4(const float* x, u32x2_t packed) {
    return res8_byte(x[0], packed[0] & 0xffffu) | (res8_byte(x[1], packed[0] >> 16) << 8) | (res8_byte(x[2], packed[1] & 0xffffu) << 16) |
           (res8_byte(x[3], packed[1] >> 16) << 24);
}
__device__ __forceinline__ float res8_value(uint32_t xb /* bf16 bits in the low half */, uint32_t byte) {
    const uint32_t E = (xb >> 7) & 0xffu;
    const float sc = E > 15u ? __uint_as_float((E - 15u) << 23) : 0.f;             // ulp / 2^8
    return __uint_as_float(xb << 16) + ((float)byte - 128.f) * sc;
}
// eight bf16 values (16 bytes) and their eight res8 bytes -> fp32
__device__ __forceinline__ void unpack8_res8(const u32x4_t v, const u32x2_t r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t rb = r[i >> 1] >> (16 * (i & 1));
        f[2 * i] = res8_value(v[i] & 0xffffu, rb & 0xffu);
        f[2 * i + 1] = res8_value(v[i] >> 16, (rb >> 8) & 0xffu);
    }
}

// ---- 16-byte epilogue stores ---------------------------------------------------------------------------------------------------------
// A column of a 32x32 accumulator block lives in the lane pair (l, l + 32): of every 8-row group g, lane l holds rows 8g + 0..3 and lane l + 32 rows
// 8g + 4..7 -- 8 bytes of bf16 each, which made every epilogue a string of 8-byte stores.  Given the packed rows of an EVEN group and of the ODD group after
// it, one v_permlane32_swap per dword leaves the lower lane with all eight rows of the even group and the upper lane with all eight of the odd group: one
// 16-byte row-contiguous store per lane at row 8 (g_even + hi) instead of two 8-byte ones.  The attention epilogues' store tail is ISSUE-bound
// (MI355X_MICROARCH.md, "attention epilogue store tail": half as many, twice as wide stores halve it).  Both lanes of a pair must be active.
__device__ __forceinline__ u32x4_t pair_rows8(u32x2_t even, u32x2_t odd) {
    const auto a = __builtin_amdgcn_permlane32_swap(even[0], odd[0], false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(even[1], odd[1], false, false);
    return u32x4_t{a[0], b[0], a[1], b[1]};
}
// the same for one dword per group (the four res8 bytes of its rows): 8 bytes per lane
__device__ __forceinline__ u32x2_t pair_rows8_dword(uint32_t even, uint32_t odd) {
    const auto a = __builtin_amdgcn_permlane32_swap(even, odd, false, false);
    return u32x2_t{a[0], a[1]};
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in every thread
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 entries */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    T r = 0;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}

// load 8 consecutive elements as float from f32 or bf16 storage (16-byte aligned for bf16, 32 for f32)
template <int DT>
__device__ __forceinline__ void load8(const void* base, size_t idx, float* f) {
    if (DT == VGPA_DTYPE_BF16) {
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(base) + idx);
        unpack8(v, f);
    } else {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
        float4 a = p[0], b = p[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
}
template <int DT>
__device__ __forceinline__ void store8(void* base, size_t idx, const float* f) {
    if (DT == VGPA_DTYPE_BF16) {
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(base) + idx) = pack8(f);
    } else {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx);
        p[0] = make_float4(f[0], f[1], f[2], f[3]);
        p[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
}
// Eight consecutive elements AS LOADED (no conversion): `Raw8<DT> r; r.load(base, idx)` for every chunk of a row first, `r.get(f)` when the values are
// needed.  A row kernel written as `if (in range) { load8(...); use }` per chunk gets an s_waitcnt behind every chunk's loads from hipcc -- one HBM round
// trip per chunk and wave; with the loads in a loop of their own (and the conversion, which is a USE, kept out of it) they leave back to back.
template <int DT>
struct Raw8;
template <>
struct Raw8<VGPA_DTYPE_BF16> {
    u32x4_t a;
    __device__ __forceinline__ void load(const void* base, size_t idx, bool in) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        a = in ? *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(base) + idx) : z;
    }
    __device__ __forceinline__ void get(float* f) const { unpack8(a, f); }
};
template <>
struct Raw8<VGPA_DTYPE_F32> {
    f32x4_t a, b;
    __device__ __forceinline__ void load(const void* base, size_t idx, bool in) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f32x4_t* p = reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + idx);
        a = in ? p[0] : z;
        b = in ? p[1] : z;
    }
    __device__ __forceinline__ void get(float* f) const {
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    }
};

template <int DT>
__device__ __forceinline__ float load1(const void* base, size_t idx) {
    if (DT == VGPA_DTYPE_BF16) return bf16_to_f32(reinterpret_cast<const bf16_t*>(base)[idx]);
    return reinterpret_cast<const float*>(base)[idx];
}
template <int DT>
__device__ __forceinline__ void store1(void* base, size_t idx, float f) {
    if (DT == VGPA_DTYPE_BF16) reinterpret_cast<bf16_t*>(base)[idx] = f32_to_bf16(f);
    else reinterpret_cast<float*>(base)[idx] = f;
}
