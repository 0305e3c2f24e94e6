// Shared pieces of the "w1" attention kernels (attention_w1.hip): one wave per SIMD, the whole 512-register file per
// wave, streamed tiles brought in by LDS-DMA (buffer_load_dwordx4 ... lds) into an unpadded, chunk-swizzled ring.
//
// LDS image of one streamed [64 rows x 64 bf16] tile: 8 KiB, row pitch 128 B (no padding: the LDS-DMA destination is
// wave-uniform base + lane * 16 B, so the image must be lane-linear), the eight 16-B chunks of row r stored at chunk
// position c ^ f(r), f(r) = (r1 << 2) | (r2 << 1) | r3 (r_i = bit i of r).  The permutation is applied on the SOURCE side
// (each lane's global address) and again on every read.  With it
//   * the 16-byte row-fragment reads (ds_read_b128: 16-lane groups of 16 different rows mod 16, one logical chunk) hit
//     16 different 16-B slots of the 256-B bank row, and
//   * the transpose reads (ds_read_b64_tr_b16: 32 lanes = 4 rows x 4 chunks x 2 halves) hit all 16 slots twice 8 B,
// i.e. both read kinds are bank-conflict free, which no padded pitch achieves for the two at once.
#pragma once
#include "mfma_tiles.h"

#define W1_TILE_BYTES 8192                  // one [64][64] bf16 tile
#define W1_SLOT_BYTES (2 * W1_TILE_BYTES)   // a ring slot = the two streamed operands of one step (K|V or Q|dO)
#define W1_SLOTS 4                          // ring depth: LDS-DMA runs two tiles ahead of the first reader
#define W1_RING_BYTES (W1_SLOTS * W1_SLOT_BYTES)

typedef __attribute__((address_space(3))) bf16x4_t* w1_lds_b64_t;

__device__ __forceinline__ uint32_t w1_swz(uint32_t r) { return (((r >> 1) & 1u) << 2) | (((r >> 2) & 1u) << 1) | ((r >> 3) & 1u); }

// wave-uniform buffer descriptor over rows [0, S) of one (batch, head) slice: rows at or past S read as zeros
struct W1Rsrc { u32x4_t w; };
__device__ __forceinline__ W1Rsrc w1_rsrc(const void* base /* uniform */, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    W1Rsrc r;
    r.w[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r.w[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r.w[2] = __builtin_amdgcn_readfirstlane(bytes);
    r.w[3] = 0x00020000u;
    return r;
}

// One LDS-DMA piece: 64 lanes x 16 B from (descriptor, per-lane byte offset, uniform byte offset) to LDS bytes
// [lds_dst, lds_dst + 1024).  Invisible to hipcc's waitcnt bookkeeping on purpose (it would drain the ring at the next
// ds_read): completion is counted by hand with w1_wait_* below.  M0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void w1_dma(uint32_t lds_dst /* uniform */, const W1Rsrc& rs, uint32_t voff, uint32_t soff /* uniform */) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs.w), "s"(soff)
        : "memory");
}

// the 4-byte-per-lane form (256 B per wave-instruction)
__device__ __forceinline__ void w1_dma4(uint32_t lds_dst /* uniform */, const W1Rsrc& rs, uint32_t voff, uint32_t soff /* uniform */) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs.w), "s"(soff)
        : "memory");
}

// wait until at most N of this wave's VMEM operations are outstanding, and all of its LDS reads have returned
#define W1_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory")

// per-lane source offsets (bytes) of the PW pieces this wave moves of a [64 x 64] tile with row stride `row_stride`
// (elements): piece j = wave * PW + i covers rows 8j .. 8j+7; lane -> (row 8j + lane/8, LDS chunk lane%8)
template <int PW>
__device__ __forceinline__ void w1_dma_offsets(int wave, int lane, uint32_t row_stride, uint32_t (&voff)[PW]) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const uint32_t row = 8u * (uint32_t)(wave * PW + i) + (uint32_t)(lane >> 3);
        const uint32_t c = (uint32_t)(lane & 7) ^ w1_swz(row);
        voff[i] = (row * row_stride + c * 8u) * 2u;
    }
}

// lane-constant LDS byte offsets of the fragment reads inside a tile (add tile base + 4096 * (row block of 32) + ...):
//   row fragments (A/B operand contracted along d): rows m = lane & 31, logical chunk 2 ks + hi
//   transpose fragments (operand contracted along rows): see frag_tr in mfma_tiles.h; index [db][second 8-row read]
struct W1Lane {
    uint32_t row[4];
    uint32_t tr[2][2];
};
__device__ __forceinline__ W1Lane w1_lane_offsets(int lane) {
    W1Lane a;
    const uint32_t m = lane & 31, hi = lane >> 5;
    const uint32_t sw = w1_swz(m);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a.row[ks] = m * 128u + ((((uint32_t)(2 * ks) + hi) ^ sw) << 4);
    // transpose read: row = R0 + 4 hi + ((lane & 15) >> 2) with R0 a multiple of 16 (second read: + 8), column =
    // 32 db + 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const uint32_t r = 4u * hi + ((uint32_t)(lane & 15) >> 2);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r3 = 0; r3 < 2; ++r3) {
            const uint32_t rr = r + 8u * r3;
            const uint32_t c = 4u * db + 2u * ((uint32_t)(lane >> 4) & 1u) + (((uint32_t)lane & 3u) >> 1);
            a.tr[db][r3] = rr * 128u + ((c ^ w1_swz(rr)) << 4) + ((uint32_t)lane & 1u) * 8u;
        }
    return a;
}

__device__ __forceinline__ bf16x8_t w1_frag_row(const uint8_t* lds, uint32_t tile_off, const W1Lane& a, int rb /* 32-row block */, int ks) {
    return *reinterpret_cast<const bf16x8_t*>(lds + (tile_off + a.row[ks] + 4096u * rb));
}
// rows 32 rb + 16 cc + {4 hi + 0..3, + 8}, columns 32 db + ...
__device__ __forceinline__ bf16x8_t w1_frag_tr(const uint8_t* lds, uint32_t tile_off, const W1Lane& a, int rb, int cc, int db) {
    const uint32_t o = tile_off + 4096u * rb + 2048u * cc;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w1_lds_b64_t)(lds + (o + a.tr[db][0])));
    bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((w1_lds_b64_t)(lds + (o + a.tr[db][1])));
    bf16x8_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo[i]; r[i + 4] = hi4[i]; }
    return r;
}
