// Helpers shared by the attention translation units (attention.hip, attention_w1.hip).
#pragma once
#include "common.h"

#include <cstdlib>

#include "mfma_tiles.h"

// fp32 adds / multiplies next to MFMAs are written one element at a time and this file is built with -fno-slp-vectorize:
// a packed fp32 instruction (v_pk_add_f32, v_pk_mul_f32) does not co-issue with the matrix pipe -- tools/dot2_probe: 2
// v_pk_add_f32 per MFMA stretch a 64 ns group of four MFMAs to 107 ns, 4 scalar v_add_f32 leave it at 68 ns -- and hipcc's
// SLP vectoriser would pack adjacent scalar operations by itself.  (Not inline asm: the compiler must see these to place
// the MFMA -> VALU hazard wait states.)
__device__ __forceinline__ float nopack_add(float a, float b) { return a + b; }
__device__ __forceinline__ float nopack_mul(float a, float b) { return a * b; }

// XCD-aware remap of the linear block id: consecutive virtual ids (same head) land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}


// bf16x8 fragment <-> 8 floats
__device__ __forceinline__ void frag_to_f32(const bf16x8_t& f, float* o) {
    const u32x4_t u = __builtin_bit_cast(u32x4_t, f);
    unpack8(u, o);
}
__device__ __forceinline__ bf16x8_t f32_to_frag(const float* o) { return __builtin_bit_cast(bf16x8_t, pack8(o)); }

// The "-m" operand of the folded softmax shift: -m = a1 + a2 + a3 exactly enough (3 bf16 pieces = 24 mantissa bits),
// living in k-slots 0..2 of an extra MFMA k-step whose K-side operand is (1,1,1,0,...): the QK^T accumulators then
// come out as  c*q.k - m  and go straight into exp2.
__device__ __forceinline__ bf16x8_t shift_frag(float m, int hi) {
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (hi == 0) {
        const float t = -m;
        const float a1 = round_bf16(t), a2 = round_bf16(t - a1), a3 = round_bf16((t - a1) - a2);
        o[0] = a1; o[1] = a2; o[2] = a3;
    }
    return f32_to_frag(o);
}


// ---- host side ----------------------------------------------------------------------------------------------------
static inline bool stride_ok(const int64_t* st) { return st && st[0] >= 0 && st[1] >= 0 && st[2] >= HD && (st[0] % 8 == 0) && (st[1] % 8 == 0) && (st[2] % 8 == 0); }
// every element offset reachable inside one (batch, head) slab and across the tensor must fit 31 bits
static inline bool range_ok(const int64_t* st, int64_t B, int64_t H, int64_t S) {
    return (B - 1) * st[0] + (H - 1) * st[1] + (S - 1) * st[2] + HD < ((int64_t)1 << 31);
}
static inline TStride mk(const int64_t* st) { TStride t; t.b = (uint32_t)st[0]; t.h = (uint32_t)st[1]; t.s = (uint32_t)st[2]; return t; }
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Workgroup slots the attention kernels have on the current device (2 workgroups of 256 threads per CU): a launch whose
// task count is not a multiple of this ends in a partially filled scheduling round.  Read once per process.
static inline int wg_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
    }
    return slots;
}
// How to run `tasks` equal tasks that each sweep `nt` tiles: n_main tasks as they are, the rest split `nsplit` ways along
// the sweep so that they fill (at most) one round.  split_mode: -1 automatic, 0 never, k >= 2 force k chunks for ALL tasks
// (tests).  Splitting is only worth it when the leftover round would be mostly empty and the chunks keep a few tiles each.
static inline void split_plan(int64_t tasks, int nt, int split_mode, int max_split, int64_t* n_main, int* nsplit, int64_t slots = 0) {
    *n_main = tasks;
    *nsplit = 1;
    if (split_mode == 0 || nt < 8) return;
    if (split_mode >= 2) {
        *n_main = 0;
        *nsplit = split_mode < nt / 2 ? split_mode : nt / 2;
        if (*nsplit > max_split) *nsplit = max_split;
        return;
    }
    if (slots <= 0) slots = wg_slots();
    const int64_t rem = tasks % slots;
    if (tasks < slots || rem == 0 || rem * 2 > slots) return;      // a single round, a full last round, or one at least half full
    int64_t k = slots / rem;
    if (k > nt / 4) k = nt / 4;
    if (k > max_split) k = max_split;
    if (k < 2) return;
    *n_main = tasks - rem;
    *nsplit = (int)k;
}

// attention.hip: the online-softmax forward over the flagged 256-row strips only (redo pass of the w1 forward)
int32_t vgpa_internal_attn_fwd_redo(const void* q, const void* k, const void* v, void* o, float* lse2, TStride sq, TStride sk, TStride sv, TStride so,
                                    int S, int H, int n_qt, int64_t tasks, const int* flags, hipStream_t stream, void* o_res = nullptr, TStride sor = TStride{0, 0, 0},
                                    int res_kind = VGPA_RES_NONE);
