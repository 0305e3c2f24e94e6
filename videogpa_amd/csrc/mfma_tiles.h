// MFMA tile helpers shared by the attention and LoRA kernels (gfx950, v_mfma_f32_32x32x16_bf16).
//   operand maps:  A[m = lane&31][k = 8*(lane>>5) + i],  B[k = 8*(lane>>5) + i][n = lane&31],
//                  D[m = (r&3) + 8*(r>>2) + 4*(lane>>5)][n = lane&31]   (r = accumulator register 0..15)
#pragma once
#include "common.h"

#define HD 64          // head dim
#define PITCH 72       // LDS row pitch in bf16 elements (144 B)
#define TILE 64        // rows of the streamed operand per iteration
#define WG_ROWS 128    // rows of the stationary operand per workgroup (4 waves x 32)
#define TILE_ELEMS (TILE * PITCH)

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

struct TStride {  // element strides of a [B, H, S, 64] view (last dim contiguous); every element offset < 2^31 (host-checked)
    uint32_t b, h, s;
};
#define SOFTMAX_RESCALE_THR 6.0f  // running max is only raised when a tile exceeds it by > 2^THR (keeps P <= 2^THR)

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }

// 8 consecutive fp32 accumulator registers -> bf16x8 MFMA operand
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& a, int base) {
    bf16x8_t r;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        f32x2_t f = {a[base + i], a[base + i + 1]};
        bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
        r[i] = h[0];
        r[i + 1] = h[1];
    }
    return r;
}

// ---- global -> register -> LDS staging of a [64 x 64] bf16 tile (rows clamped to S-1) ------------------------
__device__ __forceinline__ void tile_load(const bf16_t* base, uint32_t row_stride, int row0, int S, u32x4_t (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + 256 * j;
        int row = row0 + (c >> 3);
        row = row < S ? row : S - 1;
        r[j] = *reinterpret_cast<const u32x4_t*>(base + ((uint32_t)row * row_stride + (uint32_t)((c & 7) * 8)));
    }
}
// The loop form: buffer loads.  The descriptor covers rows [0, S) of one (batch, head) slice, the per-lane byte offset
// is computed once, the tile's row offset rides in an SGPR -- no per-iteration address VALU (the clamped form above
// costs ~20 VALU instructions per tile in loops that are VALU-issue bound) -- and rows at or past S are out of range,
// i.e. read as zeros, so the ragged last tile needs no special load path.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t tile_rsrc(const bf16_t* base /* wave-uniform */, uint32_t row_stride, int S) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, ((uint32_t)(S - 1) * row_stride + (uint32_t)HD) * 2u, 0x00020000);
}
__device__ __forceinline__ uint32_t tile_lane_byte_offset(uint32_t row_stride) {
    return ((uint32_t)(threadIdx.x >> 3) * row_stride + (uint32_t)((threadIdx.x & 7) * 8)) * 2u;
}
__device__ __forceinline__ void tile_load_buf(rsrc_t rs, uint32_t row_stride, int row0, uint32_t lane_off, u32x4_t (&r)[2]) {
    const uint32_t soff = (uint32_t)row0 * row_stride * 2u;   // uniform
    r[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff, 0);
    r[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff + 64u * row_stride, 0);   // rows +32
}
__device__ __forceinline__ void tile_store(bf16_t* lds, const u32x4_t (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + 256 * j;
        *reinterpret_cast<u32x4_t*>(lds + (c >> 3) * PITCH + (c & 7) * 8) = r[j];
    }
}

// the same three for 512-thread workgroups (one 16-byte chunk per thread and tile)
__device__ __forceinline__ void tile_load(const bf16_t* base, uint32_t row_stride, int row0, int S, u32x4_t (&r)[1]) {
    int row = row0 + (threadIdx.x >> 3);
    row = row < S ? row : S - 1;
    r[0] = *reinterpret_cast<const u32x4_t*>(base + ((uint32_t)row * row_stride + (uint32_t)((threadIdx.x & 7) * 8)));
}
__device__ __forceinline__ void tile_load_buf(rsrc_t rs, uint32_t row_stride, int row0, uint32_t lane_off, u32x4_t (&r)[1]) {
    r[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, (uint32_t)row0 * row_stride * 2u, 0);
}
__device__ __forceinline__ void tile_store(bf16_t* lds, const u32x4_t (&r)[1]) {
    *reinterpret_cast<u32x4_t*>(lds + (threadIdx.x >> 3) * PITCH + (threadIdx.x & 7) * 8) = r[0];
}

// A/B fragment whose contraction index runs along the row (d contiguous): lane (l&31, l>>5) reads 16 B.
__device__ __forceinline__ bf16x8_t frag_row(const bf16_t* lds, int rowbase, int ks, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(lds + (rowbase + (lane & 31)) * PITCH + ks * 16 + (lane >> 5) * 8);
}

// A fragment whose contraction index runs ACROSS rows (transposed use of a row-major tile).
// Element i of lane (m = l&31, hi = l>>5) is tile[rowbase + 4*hi + (i&3) + 8*(i>>2)][colbase + m]: exactly the row
// order in which the previous product left its accumulator rows (row = (r&3) + 8*(r>>2) + 4*hi), so packed
// accumulators can be used as the B operand unshuffled.  Two ds_read_b64_tr_b16: each 16-lane group reads a
// [4 rows x 16 cols] block and receives it column-per-lane.
__device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* lds, int rowbase, int colbase, int lane) {
    const int hi = lane >> 5;
    const bf16_t* p = lds + (rowbase + 4 * hi + ((lane & 15) >> 2)) * PITCH + colbase + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    typedef __attribute__((address_space(3))) bf16x4_t* lds_ptr_t;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p));
    bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr_t)(p + 8 * PITCH));
    bf16x8_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo[i]; r[i + 4] = hi4[i]; }
    return r;
}

// stationary-operand fragments straight from HBM: lane (row = l&31, hi) takes 8 contiguous d per k-step
__device__ __forceinline__ void load_row_frags(const bf16_t* base, uint32_t row_stride, int row, int S, int lane, bf16x8_t (&f)[4]) {
    int r = row + (lane & 31);
    r = r < S ? r : S - 1;
    const bf16_t* p = base + ((uint32_t)r * row_stride + (uint32_t)((lane >> 5) * 8));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 16);
}

// Make the compiler wait HERE for fragments loaded from HBM (a zero-instruction use).  Without it hipcc's waitcnt pass
// first meets the use inside the tile loop, keeps "a VMEM load may still be writing these registers" in the loop-header
// state, and emits s_waitcnt vmcnt(3..0) in front of the loop's MFMAs -- which also drains the loop's own K/V prefetch
// loads in every iteration.
__device__ __forceinline__ void frags_arrived(const bf16x8_t (&f)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(f[ks]));
}

// row index inside a 32-row MFMA block of accumulator register r for this lane
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
