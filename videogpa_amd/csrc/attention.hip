// 3D full attention (all T*H*W + text tokens, non-causal, head_dim 64) for gfx950: flash-style forward and
// a split backward (dQ kernel; dK/dV kernel), hand-written around v_mfma_f32_32x32x16_bf16.
// Replaces F.scaled_dot_product_attention inside diffusers' CogVideoXAttnProcessor2_0 as reached from
// train/CogVideoX-5B/03_train.py:134-151; oracle: oracle/cogvideox.py::block_forward (softmax(QK^T/8)V).
//
// Design (MI355X-first):
//  * 256-thread workgroup = 4 waves; a wave owns 32 query rows (fwd, dQ) or 32 key rows (dK/dV) and keeps its
//    operand fragments and accumulators in registers for the whole sweep over the other sequence axis.
//  * Every product is computed TRANSPOSED (S^T = K Q^T, O^T = V^T P^T, ...) so the softmax row a lane works on
//    is the MFMA column lane&31: running max / sum / LSE / delta are lane-local scalars, and the fp32
//    accumulator registers of one product are, after a bf16 pack, directly the B operand of the next one
//    (the k-slot permutation this implies is applied to the A-side LDS reads instead of shuffling P).
//  * K/V (or Q/dO) tiles of 64 rows are staged HBM -> registers -> LDS, double-buffered, one barrier per tile;
//    row pitch 144 B makes the 16-byte fragment reads bank-conflict free.  Operands that are contracted over
//    their row index are read with the gfx950 hardware transpose read (ds_read_b64_tr_b16), so every tensor
//    stays row-major [tokens, 64] in HBM and no transposed copy is ever written.
//  * Loads clamp the row index to S-1 and the tail is masked, so S needs no padding (17 776 = 277*64 + 48).
//  * blockIdx is remapped so that the workgroups an XCD runs concurrently share (batch, head) and hit K/V in
//    that XCD's private L2.
#include "common.h"

#include <cstdlib>
#include <type_traits>

#include "attn_common.h"

#ifndef FWD_WPS
#define FWD_WPS 2   // waves per SIMD the forward is compiled for (1 = the whole 512-register file per wave)
#endif
#ifndef DQ_WPS
#define DQ_WPS 2
#endif

// =====================================================================================================
// Forward:  O = softmax(scale * Q K^T) V ;  lse2 = log2 sum_k exp2(scale*log2e * q.k)
// =====================================================================================================
// -DFWD_DIAG: s_memtime stamps at four points of tiles 128..131 plus HW_ID, written INTO THE LSE BUFFER (the results are
// then wrong by design); decoded by tools/fwd_diag.py.  -DFWD_DYN_LDS=90000 forces one workgroup per CU.
#ifdef FWD_DIAG
#define DIAG_STAMP(P)                                                                               \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) if (t == 128 + i_) diag_t[4 * i_ + (P)] = (unsigned)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#else
#define DIAG_STAMP(P)
#endif
#ifndef FWD_DYN_LDS
#define FWD_DYN_LDS 0
#endif
#define PSUM_TRIGGER 1024.0f   // a half-lane tile sum above this (or inf/NaN) means some score outgrew the running max by > ~2^5

#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_fwd_v1_kernel.inc"
#endif  // VGPA_VARIANTS

// =====================================================================================================
// Forward, software-pipelined (what vgpa_attn_fwd launches; the kernel above is kept as -DFWD_V1 for the phase-stamp
// diagnostics and as the reference point of DESIGN.md 4.1).  The tile loop body is ONE basic block in which independent work of
// neighbouring half-tiles (32 keys) can overlap inside a wave:
//     sB = QK^T(t, keys 32..63)   ||  pA = exp2(sA)            sA = scores of (t, keys 0..31), made one step earlier
//     O += V^T pA^T               ||  pB = exp2(sB)
//     sA = QK^T(t+1, keys 0..31)  ||  O += V^T pB^T
// with the same 64 score registers (exp in place; a half is overwritten right after its PV consumed it).  To make that
// legal the softmax check moves BEHIND the PV product: P is formed against the running max m, the tile is accumulated
// unconditionally, and only then is the tile's partial sum looked at.  A sum above PSUM_TRIGGER means some score
// outgrew m; O and l are still exactly consistent (everything is scaled by 2^-m), so the repair is "raise m to the true
// max, rescale O, l and the already-made sA" -- nothing is recomputed.  What cannot be repaired is an overflow inside a
// single tile (a score > m + 64 in log2 units; impossible for LayerNorm-ed q, k with sane weights): it sets a workgroup
// flag and the q-strip is redone by the plain online-softmax loop (safe_tile for every tile), which also handles the
// first tile (establishes m) and the ragged last tile.  K tiles: 3-slot LDS ring, V tiles: 2-slot ring.
// =====================================================================================================
#define PSUM_OVERFLOW 1.8446744e19f   // 2^64: a half-lane tile sum above this (or NaN) -> redo the strip in safe mode

template <int QB>
__device__ __forceinline__ void qk_half(const bf16_t* kl, int kb, int lane, const bf16x8_t& kx, const bf16x8_t (&qx)[QB],
                                        const bf16x8_t (&qf)[QB][4], f32x16_t (&s)[QB]) {
#pragma unroll
    for (int j = 0; j < QB; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[j][i] = 0.f;
        s[j] = mfma32(kx, qx[j], s[j]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t kf = frag_row(kl, kb * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < QB; ++j) s[j] = mfma32(kf, qf[j][ks], s[j]);
    }
}

template <int QB>
__device__ __forceinline__ void exp_half(f32x16_t (&s)[QB], f32x2_t (&ps2)[QB]) {
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
            s[j][r + 1] = __builtin_amdgcn_exp2f(s[j][r + 1]);
            ps2[j][0] = nopack_add(ps2[j][0], s[j][r]);
            ps2[j][1] = nopack_add(ps2[j][1], s[j][r + 1]);
        }
}

template <int QB>
__device__ __forceinline__ void pv_half(const bf16_t* vl, int kb, int lane, const f32x16_t (&s)[QB], f32x16_t (&o)[QB][2]) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        bf16x8_t pf[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) pf[j] = pack_frag(s[j], 8 * cc);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const bf16x8_t vf = frag_tr(vl, kb * 32 + 16 * cc, db * 32, lane);
#pragma unroll
            for (int j = 0; j < QB; ++j) o[j][db] = mfma32(vf, pf[j], o[j][db]);
        }
    }
}

// exact row max of tile (kl) over the valid keys, both halves; raises m and rescales O, l (and `carry`, scores that were
// already made against the old m).  Returns with qx holding the new shift and s0 / s1 the raw scores.
template <int QB, bool HAS_CARRY>
__device__ __forceinline__ void raise_max(const bf16_t* kl, int key0, int S, bool tail, int lane, int hi, const bf16x8_t (&qf)[QB][4],
                                          bf16x8_t (&qx)[QB], float (&m)[QB], float (&l)[QB], f32x16_t (&o)[QB][2],
                                          f32x16_t (&carry)[QB], f32x16_t (&s0)[QB], f32x16_t (&s1)[QB]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x16_t(&s)[QB] = kb ? s1 : s0;
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[j][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t kf = frag_row(kl, kb * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < QB; ++j) s[j] = mfma32(kf, qf[j][ks], s[j]);
        }
        if (tail) {
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + kb * 32 + acc_row(r, hi) >= S) s[j][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float mx = s0[j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[j][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[j][r]);
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m[j], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[j] - m_new);   // m = -inf on the first tile: alpha = 0, O = l = 0 stay 0
        if (HAS_CARRY) {
            const float d = m[j] - m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) carry[j][i] += d;
        }
        m[j] = m_new;
        qx[j] = shift_frag(m_new, hi);
        l[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[j][0][i] *= alpha; o[j][1][i] *= alpha; }
    }
}

// plain online-softmax step for one tile (first tile, ragged last tile, and every tile of the safe-mode redo)
template <int QB>
__device__ __forceinline__ void safe_tile(const bf16_t* kl, const bf16_t* vl, int key0, int S, bool tail, int lane, int hi,
                                          const bf16x8_t (&qf)[QB][4], bf16x8_t (&qx)[QB], float (&m)[QB], float (&l)[QB],
                                          f32x16_t (&o)[QB][2]) {
    f32x16_t s0[QB], s1[QB];
    raise_max<QB, false>(kl, key0, S, tail, lane, hi, qf, qx, m, l, o, s0, s0, s1);
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p0 = __builtin_amdgcn_exp2f(s0[j][r] - m[j]), p1 = __builtin_amdgcn_exp2f(s1[j][r] - m[j]);
            s0[j][r] = p0;
            s1[j][r] = p1;
            ps += p0 + p1;
        }
        l[j] += ps;
    }
    pv_half<QB>(vl, 0, lane, s0, o);
    pv_half<QB>(vl, 1, lane, s1, o);
}

// SPLIT = false: workgroup = task (one 256-row query strip of one head) task0 + remapped blockIdx, all key tiles.
// SPLIT = true (the leftover tasks that would otherwise run as a mostly empty last scheduling round, see vgpa_attn_fwd_ws):
// workgroup = (task, chunk) = (task0 + blockIdx / nsplit, blockIdx % nsplit) sweeps only key tiles [nt*chunk/nsplit,
// nt*(chunk+1)/nsplit) and leaves its un-normalised O, m and l in `part`; attn_fwd_merge_kernel combines the chunks.
// (blockIdx % nsplit is also the XCD the workgroup lands on for nsplit = 8: the workgroups of one XCD share one key range.)
#define FWD_PART_FLOATS (256 * (HD + 2))   // per (task, chunk): O[256][64], m[256], l[256]
template <int QB, int NW, bool SPLIT>
__global__ __launch_bounds__(64 * NW, (NW == 8) ? 4 : FWD_WPS) void attn_fwd_pipe_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                                 float* __restrict__ LSE2, TStride sq, TStride sk, TStride sv, TStride so,
                                                                 int S, int H, int n_qt, int task0, int nsplit, float* __restrict__ part,
                                                                 const int* __restrict__ only_flagged = nullptr, void* __restrict__ ORES = nullptr,
                                                                 TStride sor = TStride{0, 0, 0}, int res_kind = VGPA_RES_NONE) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[5 * TILE_ELEMS];  // K ring [3], V ring [2]
    __shared__ int redo_flag;
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    if (only_flagged && only_flagged[vid] == 0) return;   // redo pass behind the w1 forward: only the strips it flagged
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int q0 = (qt * NW + wave) * (32 * QB);

    const bf16_t* Qb = Q + ((size_t)b * sq.b + (size_t)h * sq.h);
    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    const bf16_t* Vb = V + ((size_t)b * sv.b + (size_t)h * sv.h);
    bf16_t* const kring = lds;
    bf16_t* const vring = lds + 3 * TILE_ELEMS;

    bf16x8_t qf[QB][4], qx[QB];
    f32x16_t o[QB][2];
    float m[QB], l[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        load_row_frags(Qb, sq.s, q0 + 32 * j, S, lane, qf[j]);
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
        m[j] = -INFINITY;
        l[j] = 0.f;
        qx[j] = shift_frag(0.f, hi);
    }
    bf16x8_t kx;   // K-side of the shift k-step: ones in k-slots 0..2 of the lower half-lanes
    {
        float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hi == 0) { o8[0] = 1.f; o8[1] = 1.f; o8[2] = 1.f; }
        kx = f32_to_frag(o8);
    }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's key tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;
    const bool ragged = (S & (TILE - 1)) != 0 && nt == nt_all;       // the ragged tile, if any, is the global last one
    const rsrc_t krs = tile_rsrc(Kb, sk.s, S), vrs = tile_rsrc(Vb, sv.s, S);
    const uint32_t koff = tile_lane_byte_offset(sk.s), voff = tile_lane_byte_offset(sv.s);
    u32x4_t kr[8 / NW], vr[8 / NW];
    if (threadIdx.x == 0) redo_flag = 0;
    // prologue: K(tb..tb+2), V(tb..tb+1) -> LDS (rows past S read as zeros; their scores are masked or unused)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        tile_load_buf(krs, sk.s, (tb + i) * TILE, koff, kr);
        tile_store(kring + i * TILE_ELEMS, kr);
        if (i < 2) {
            tile_load_buf(vrs, sv.s, (tb + i) * TILE, voff, vr);
            tile_store(vring + i * TILE_ELEMS, vr);
        }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) frags_arrived(qf[j]);
    __syncthreads();

    // first tile: establishes m (also the ragged tile when it is the only one)
    safe_tile<QB>(kring, vring, tb * TILE, S, ragged && nt - tb == 1, lane, hi, qf, qx, m, l, o);

    f32x16_t sA[QB], sB[QB];
    if (nt - tb > 2) qk_half<QB>(kring + TILE_ELEMS, 0, lane, kx, qx, qf, sA);
    int kslot = 1, vslot = 1;   // ring slots of tile t
    for (int t = tb + 1; t < nt - 1; ++t) {
        const bf16_t* kl = kring + kslot * TILE_ELEMS;
        const int kslot1 = kslot == 2 ? 0 : kslot + 1, kslot2 = kslot1 == 2 ? 0 : kslot1 + 1;
        const bf16_t* kl1 = kring + kslot1 * TILE_ELEMS;
        const bf16_t* vl = vring + vslot * TILE_ELEMS;
        tile_load_buf(krs, sk.s, (t + 2) * TILE, koff, kr);   // one staging register set, used for K then for V
        f32x2_t ps2[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) ps2[j] = (f32x2_t){0.f, 0.f};
        qk_half<QB>(kl, 1, lane, kx, qx, qf, sB);
        exp_half<QB>(sA, ps2);
        pv_half<QB>(vl, 0, lane, sA, o);
        tile_store(kring + kslot2 * TILE_ELEMS, kr);
        tile_load_buf(vrs, sv.s, (t + 1) * TILE, voff, kr);
#ifndef PIPE_NO_SB
        // nothing crosses: everything that reads the old sA is done before the next half-tile's scores are started, so
        // they are written into the same registers (otherwise the rotation costs 32 register copies per tile)
        __builtin_amdgcn_sched_barrier(0);
#endif
        exp_half<QB>(sB, ps2);
        qk_half<QB>(kl1, 0, lane, kx, qx, qf, sA);
        pv_half<QB>(vl, 1, lane, sB, o);
        bool grew = false, over = false;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float psum = ps2[j][0] + ps2[j][1];
            l[j] += psum;
            grew = grew || !(psum <= PSUM_TRIGGER);
            over = over || !(psum <= PSUM_OVERFLOW);
        }
        if (__any(grew)) {
            if (__any(over)) redo_flag = 1;
            raise_max<QB, true>(kl, t * TILE, S, false, lane, hi, qf, qx, m, l, o, sA, sB, sB);
        }
        tile_store(vring + (vslot ^ 1) * TILE_ELEMS, kr);
        kslot = kslot1;
        vslot ^= 1;
        __syncthreads();
    }
    if (nt - tb > 1) safe_tile<QB>(kring + kslot * TILE_ELEMS, vring + vslot * TILE_ELEMS, (nt - 1) * TILE, S, ragged, lane, hi, qf, qx, m, l, o);

    if (redo_flag) {   // workgroup-uniform (written before the loop's last barrier); essentially never taken
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
            m[j] = -INFINITY;
            l[j] = 0.f;
        }
        for (int t = tb; t < nt; ++t) {
            __syncthreads();
            tile_load_buf(krs, sk.s, t * TILE, koff, kr);
            tile_load_buf(vrs, sv.s, t * TILE, voff, vr);
            tile_store(kring, kr);
            tile_store(vring, vr);
            __syncthreads();
            safe_tile<QB>(kring, vring, t * TILE, S, ragged && t == nt - 1, lane, hi, qf, qx, m, l, o);
        }
    }

    if (SPLIT) {   // partial result of this key range: un-normalised O (scaled by 2^-m), m, l
        float* pb = part + ((size_t)(vid - task0) * nsplit + chunk) * FWD_PART_FLOATS;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const int r = wave * (32 * QB) + 32 * j + (lane & 31);
            const float lt = l[j] + other_half(l[j]);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t w = {o[j][db][4 * g], o[j][db][4 * g + 1], o[j][db][4 * g + 2], o[j][db][4 * g + 3]};
                    *reinterpret_cast<f32x4_t*>(pb + r * HD + db * 32 + 8 * g + 4 * hi) = w;
                }
            if (hi == 0) { pb[256 * HD + r] = m[j]; pb[256 * HD + 256 + r] = lt; }
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const float lt = l[j] + other_half(l[j]);
        const float inv = 1.f / lt;
        const int q = q0 + 32 * j + (lane & 31);
        if (q < S) {
            bf16_t* op = O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s);
            const size_t ro = (size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s;   // the output's residual for the backward's delta (attention_w1.hip)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[i] = o[j][db][4 * g + i] * inv;
                    u32x2_t w;
                    w[0] = pack_bf16x2(x[0], x[1]);
                    w[1] = pack_bf16x2(x[2], x[3]);
                    *reinterpret_cast<u32x2_t*>(op + db * 32 + 8 * g + 4 * hi) = w;
                    if (res_kind == VGPA_RES_8) {
                        *reinterpret_cast<uint32_t*>((uint8_t*)ORES + ro + db * 32 + 8 * g + 4 * hi) = res8_pack4(x, w);
                    } else if (res_kind == VGPA_RES_BF16) {
                        u32x2_t r;
                        r[0] = pack_bf16x2(x[0] - __uint_as_float(w[0] << 16), x[1] - __uint_as_float(w[0] & 0xffff0000u));
                        r[1] = pack_bf16x2(x[2] - __uint_as_float(w[1] << 16), x[3] - __uint_as_float(w[1] & 0xffff0000u));
                        *reinterpret_cast<u32x2_t*>((bf16_t*)ORES + ro + db * 32 + 8 * g + 4 * hi) = r;
                    }
                }
            if (hi == 0) LSE2[(int64_t)bh * S + q] = m[j] + __builtin_amdgcn_logf(lt);  // v_log_f32 is log2
        }
    }
}

// combine the key-range chunks of the split tasks: one wave per query row, lane = d
__global__ __launch_bounds__(256) void attn_fwd_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_qt, bf16_t* __restrict__ O,
                                                               TStride so, float* __restrict__ LSE2, int S, int H) {
    const int lane = threadIdx.x & 63, r = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6), tl = blockIdx.x >> 6;
    const int vid = task0 + tl, bh = vid / n_qt, qt = vid % n_qt;
    const int q = qt * 256 + r;
    if (q >= S) return;
    const float* pb = part + (size_t)tl * nsplit * FWD_PART_FLOATS;
    float M = -INFINITY;
    for (int c = 0; c < nsplit; ++c) M = fmaxf(M, pb[(size_t)c * FWD_PART_FLOATS + 256 * HD + r]);
    float acc = 0.f, L = 0.f;
    for (int c = 0; c < nsplit; ++c) {
        const float* pc = pb + (size_t)c * FWD_PART_FLOATS;
        const float w = __builtin_amdgcn_exp2f(pc[256 * HD + r] - M);
        acc += w * pc[r * HD + lane];
        L += w * pc[256 * HD + 256 + r];
    }
    const int b = bh / H, h = bh % H;
    O[(size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s + lane] = f32_to_bf16(acc / L);
    if (lane == 0) LSE2[(int64_t)bh * S + q] = M + __builtin_amdgcn_logf(L);
}

#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_fwd_pp_kernel.inc"
#endif  // VGPA_VARIANTS

// =====================================================================================================
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]
// =====================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, TStride sdo,
                                                           TStride so, int S, int H, int64_t total /* B*H*S */, float* __restrict__ delta,
                                                           const void* __restrict__ ORES = nullptr, TStride sor = TStride{0, 0, 0},
                                                           int res_kind = VGPA_RES_NONE) {
    // 8 lanes per (b,h,q) row, 16 B each
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gid >> 3;
    const int c8 = (int)(gid & 7);
    float acc = 0.f;
    if (row < total) {
        const int q = (int)(row % S);
        const int64_t bh = row / S;
        const int h = (int)(bh % H), b = (int)(bh / H);
        float a[8], o[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h + (size_t)q * sdo.s + c8 * 8)), a);
        const u32x4_t ob = *reinterpret_cast<const u32x4_t*>(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s + c8 * 8));
        const size_t ro = (size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s + c8 * 8;
        if (res_kind == VGPA_RES_8) {   // the forward's 8 further mantissa bits (common.h res8)
            unpack8_res8(ob, *reinterpret_cast<const u32x2_t*>((const uint8_t*)ORES + ro), o);
        } else {
            unpack8(ob, o);
            if (res_kind == VGPA_RES_BF16) {   // the forward's rounding residual (attention_w1.hip, w1_residual4)
                float r[8];
                unpack8(*reinterpret_cast<const u32x4_t*>((const bf16_t*)ORES + ro), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j] * o[j];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (row < total && c8 == 0) delta[row] = acc;
}

// =====================================================================================================
// Backward, dQ:  dQ = scale * sum_k dS[q,k] K[k],  dS = P o (dP - delta),  P = exp2(c*s - lse2),  dP = dO V^T
// =====================================================================================================
// SPLIT: as in the forward -- workgroup (task0 + blockIdx / nsplit, chunk blockIdx % nsplit) sweeps key tiles
// [nt*chunk/nsplit, nt*(chunk+1)/nsplit) and leaves its unscaled fp32 dQ [128*QB][64] in `part`; attn_dq_merge_kernel adds them.
template <int QB, bool SPLIT>
__global__ __launch_bounds__(256, DQ_WPS) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ LSE2, const float* __restrict__ DELTA,
                                                               bf16_t* __restrict__ dQ, TStride sq, TStride sk, TStride sv, TStride sdo,
                                                               TStride sdq, int S, int H, int n_qt, float scale, int task0, int nsplit,
                                                               float* __restrict__ part) {
    // a wave owns QB blocks of 32 query rows: every K / V fragment read from LDS feeds QB MFMAs
    __shared__ __attribute__((aligned(16))) bf16_t lds[4 * TILE_ELEMS];  // K[2], V[2]
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int q0 = (qt * 4 + wave) * (32 * QB);

    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    const bf16_t* Vb = V + ((size_t)b * sv.b + (size_t)h * sv.h);
    bf16x8_t qf[QB][4], dof[QB][4], qx[QB], dx[QB];
    f32x16_t dq[QB][2];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        load_row_frags(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0 + 32 * j, S, lane, qf[j]);
        load_row_frags(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, q0 + 32 * j, S, lane, dof[j]);
        int qc = q0 + 32 * j + (lane & 31);
        qc = qc < S ? qc : S - 1;
        qx[j] = shift_frag(LSE2[(int64_t)bh * S + qc], hi);   // -lse folded into the QK^T chain (see forward)
        dx[j] = shift_frag(DELTA[(int64_t)bh * S + qc], hi);  // -delta folded into the dP chain
#pragma unroll
        for (int i = 0; i < 16; ++i) { dq[j][0][i] = 0.f; dq[j][1][i] = 0.f; }
    }
    bf16x8_t kx;
    {
        float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hi == 0) { o8[0] = 1.f; o8[1] = 1.f; o8[2] = 1.f; }
        kx = f32_to_frag(o8);
    }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's key tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;
    const rsrc_t krs = tile_rsrc(Kb, sk.s, S), vrs = tile_rsrc(Vb, sv.s, S);
    const uint32_t koff = tile_lane_byte_offset(sk.s), voff = tile_lane_byte_offset(sv.s);
    u32x4_t kr[2];   // one staging register set: K of the next tile during the first key half, V during the second
    tile_load_buf(krs, sk.s, tb * TILE, koff, kr);
    tile_store(lds + (tb & 1) * TILE_ELEMS, kr);
    tile_load_buf(vrs, sv.s, tb * TILE, voff, kr);
    tile_store(lds + (2 + (tb & 1)) * TILE_ELEMS, kr);
#pragma unroll
    for (int j = 0; j < QB; ++j) { frags_arrived(qf[j]); frags_arrived(dof[j]); }
    __syncthreads();

    for (int t = tb; t < nt; ++t) {
        const bf16_t* kl = lds + (t & 1) * TILE_ELEMS;
        const bf16_t* vl = lds + (2 + (t & 1)) * TILE_ELEMS;
        tile_load_buf(krs, sk.s, (t + 1) * TILE, koff, kr);   // past the last tile: zeros into a buffer nobody reads
        const bool tail = (t == nt_all - 1) && (S & (TILE - 1));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t s[QB], dp[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[j][i] = 0.f; dp[j][i] = 0.f; }
                s[j] = mfma32(kx, qx[j], s[j]);                                                          // - lse2[q]
                dp[j] = mfma32(kx, dx[j], dp[j]);                                                        // - delta[q]
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t kf = frag_row(kl, kb * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < QB; ++j) s[j] = mfma32(kf, qf[j][ks], s[j]);                         // S^T[key,q] - lse2
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t vf = frag_row(vl, kb * 32, ks, lane);
#pragma unroll
                for (int j = 0; j < QB; ++j) dp[j] = mfma32(vf, dof[j][ks], dp[j]);                      // dP^T[key,q] - delta
            }
            if (tail) {   // keys past the end contribute nothing: exp2(-inf) = 0 (masking kept out of the exp loop)
#pragma unroll
                for (int j = 0; j < QB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * TILE + kb * 32 + acc_row(r, hi) >= S) s[j][r] = -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[j][r] = nopack_mul(dp[j][r], __builtin_amdgcn_exp2f(s[j][r]));   // dS^T = P * (dP - delta)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                bf16x8_t dsf[QB];
#pragma unroll
                for (int j = 0; j < QB; ++j) dsf[j] = pack_frag(s[j], 8 * cc);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8_t ktf = frag_tr(kl, kb * 32 + 16 * cc, db * 32, lane);
#pragma unroll
                    for (int j = 0; j < QB; ++j) dq[j][db] = mfma32(ktf, dsf[j], dq[j][db]);             // dQ^T[d,q]
                }
            }
            if (kb == 0) {
                tile_store(lds + ((t + 1) & 1) * TILE_ELEMS, kr);
                tile_load_buf(vrs, sv.s, (t + 1) * TILE, voff, kr);
            }
        }
        tile_store(lds + (2 + ((t + 1) & 1)) * TILE_ELEMS, kr);
        __syncthreads();
    }
    if (SPLIT) {
        float* pb = part + ((size_t)(vid - task0) * nsplit + chunk) * (128 * QB * HD);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const int r = wave * (32 * QB) + 32 * j + (lane & 31);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t w = {dq[j][db][4 * g], dq[j][db][4 * g + 1], dq[j][db][4 * g + 2], dq[j][db][4 * g + 3]};
                    *reinterpret_cast<f32x4_t*>(pb + r * HD + db * 32 + 8 * g + 4 * hi) = w;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int q = q0 + 32 * j + (lane & 31);
        if (q < S) {
            bf16_t* op = dQ + ((size_t)b * sdq.b + (size_t)h * sdq.h + (size_t)q * sdq.s);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2_t w;
                    w[0] = pack_bf16x2(dq[j][db][4 * g] * scale, dq[j][db][4 * g + 1] * scale);
                    w[1] = pack_bf16x2(dq[j][db][4 * g + 2] * scale, dq[j][db][4 * g + 3] * scale);
                    *reinterpret_cast<u32x2_t*>(op + db * 32 + 8 * g + 4 * hi) = w;
                }
        }
    }
}

// sum the key-range chunks of the split dQ tasks: one wave per query row, lane = d
__global__ __launch_bounds__(256) void attn_dq_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_qt, int rows_per_task,
                                                              bf16_t* __restrict__ dQ, TStride sdq, int S, int H, float scale) {
    const int lane = threadIdx.x & 63, rows4 = rows_per_task >> 2;
    const int r = ((int)blockIdx.x % rows4) * 4 + (threadIdx.x >> 6), tl = (int)blockIdx.x / rows4;
    const int vid = task0 + tl, bh = vid / n_qt, qt = vid % n_qt;
    const int q = qt * rows_per_task + r;
    if (q >= S) return;
    const float* pb = part + (size_t)tl * nsplit * rows_per_task * HD + r * HD + lane;
    float acc = 0.f;
    for (int c = 0; c < nsplit; ++c) acc += pb[(size_t)c * rows_per_task * HD];
    const int b = bh / H, h = bh % H;
    dQ[(size_t)b * sdq.b + (size_t)h * sdq.h + (size_t)q * sdq.s + lane] = f32_to_bf16(acc * scale);
}

// =====================================================================================================
// Backward, dK / dV:  dV = P^T dO ,  dK = scale * dS^T Q       (workgroup owns 128 keys, streams 64-query tiles)
// =====================================================================================================
#ifndef DKV_WAVES
#define DKV_WAVES 2
#endif
// SPLIT: workgroup (task0 + blockIdx / nsplit, chunk blockIdx % nsplit) sweeps query tiles [nt*chunk/nsplit, nt*(chunk+1)/nsplit)
// and leaves unscaled fp32 dK [128][64], dV [128][64] in `part`; attn_dkv_merge_kernel adds the chunks.
template <bool SPLIT>
__global__ __launch_bounds__(256, DKV_WAVES) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                             const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                             const float* __restrict__ LSE2, const float* __restrict__ DELTA,
                                                             bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, TStride sq, TStride sk,
                                                             TStride sv, TStride sdo, TStride sdk, TStride sdv, int S, int H, int n_kt,
                                                             float kscale /* scale / (scale*log2e) = ln 2: Q is pre-scaled */, int task0,
                                                             int nsplit, float* __restrict__ part) {
    // Q[3], dO[3] tile rings.  The 8 padding columns (64..71) of every row carry the row's softmax statistics as three
    // bf16 pieces (-lse in the Q tile, -delta in the dO tile); one extra MFMA k-step against a (1,1,1,0,...) operand folds
    // them into the S and dP accumulators, so P = exp2(acc) and dS = P * acc with no per-score subtract.
    // Software pipeline (as in the forward): the scores of the NEXT tile's first q-block are made at the end of this
    // tile, between the two dV/dK products, so every stretch of the loop body has both MFMA and VALU work in it.
    __shared__ __attribute__((aligned(16))) bf16_t lds[6 * TILE_ELEMS + 16];
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_kt, kt = vid % n_kt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int k0 = kt * WG_ROWS + wave * 32;

    const bf16_t* Qb = Q + ((size_t)b * sq.b + (size_t)h * sq.h);
    const bf16_t* dOb = dO + ((size_t)b * sdo.b + (size_t)h * sdo.h);
    const float* Lb = LSE2 + (int64_t)bh * S;
    const float* Db = DELTA + (int64_t)bh * S;
    bf16x8_t kf[4], vf[4];
    load_row_frags(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, k0, S, lane, kf);
    load_row_frags(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, k0, S, lane, vf);
    bf16x8_t ones;   // B operand of the statistics k-step (the upper half-lanes read the next row's data: times 0)
    {
        float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hi == 0) { o8[0] = 1.f; o8[1] = 1.f; o8[2] = 1.f; }
        ones = f32_to_frag(o8);
    }
    // everything the statistics k-step can touch must be finite: clear the whole LDS array once
    for (int i = threadIdx.x; i < (6 * TILE_ELEMS + 16) / 8; i += 256) {
        u32x4_t z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4_t*>(lds + i * 8) = z;
    }
    __syncthreads();

    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's query tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;
    bf16_t* const qring = lds;
    bf16_t* const doring = lds + 3 * TILE_ELEMS;
    u32x4_t qr[2], dor[2];
    float st = 0.f;
    auto stat_load = [&](int t) {
        if (threadIdx.x < 2 * TILE) {
            int q = t * TILE + (threadIdx.x & (TILE - 1));
            q = q < S ? q : S - 1;
            st = (threadIdx.x < TILE) ? Lb[q] : Db[q];
        }
    };
    auto stat_store = [&](int slot) {   // threads 0..63: -lse pieces into the Q tile, 64..127: -delta pieces into the dO tile
        if (threadIdx.x < 2 * TILE) {
            const float t0 = -st;
            const float a1 = round_bf16(t0), a2 = round_bf16(t0 - a1), a3 = round_bf16((t0 - a1) - a2);
            u32x4_t w = {pack_bf16x2(a1, a2), pack_bf16x2(a3, 0.f), 0u, 0u};
            bf16_t* tile = (threadIdx.x < TILE ? qring : doring) + slot * TILE_ELEMS;
            *reinterpret_cast<u32x4_t*>(tile + (threadIdx.x & (TILE - 1)) * PITCH + 64) = w;
        }
    };
    const rsrc_t qrs = tile_rsrc(Qb, sq.s, S), dors = tile_rsrc(dOb, sdo.s, S);
    const uint32_t qoff = tile_lane_byte_offset(sq.s), dooff = tile_lane_byte_offset(sdo.s);
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // tiles tb and tb + 1 (a tile past the end reads zeros)
        tile_load_buf(qrs, sq.s, (tb + i) * TILE, qoff, qr);
        tile_load_buf(dors, sdo.s, (tb + i) * TILE, dooff, dor);
        stat_load(tb + i);
        tile_store(qring + i * TILE_ELEMS, qr);
        tile_store(doring + i * TILE_ELEMS, dor);
        stat_store(i);
    }
    frags_arrived(kf);
    frags_arrived(vf);
    __syncthreads();

    // S[q,key] - lse2 and dP[q,key] - delta of one 32-row q-block
    auto scores = [&](const bf16_t* ql, const bf16_t* dol, int qb, f32x16_t& s, f32x16_t& dp) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        bf16x8_t qa[5], da[5];   // all ten row fragments are read up front so the two MFMA chains issue back to back
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) { qa[ks] = frag_row(ql, qb * 32, ks, lane); da[ks] = frag_row(dol, qb * 32, ks, lane); }
        s = mfma32(qa[4], ones, s);
        dp = mfma32(da[4], ones, dp);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = mfma32(qa[ks], kf[ks], s);
            dp = mfma32(da[ks], vf[ks], dp);
        }
    };
    // P = exp2(s) (in place), dS = P * dp (into dp)
    auto softmax_grad = [&](f32x16_t& s, f32x16_t& dp, int row0, bool mask) {
        if (mask) {   // query rows past the end: exp2(-inf) = 0
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (row0 + acc_row(r, hi) >= S) s[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r]);
            dp[r] = nopack_mul(dp[r], s[r]);
        }
    };
    // dV^T[d,key] += dO^T P,  dK^T[d,key] += Q^T dS   for one q-block
    auto accumulate = [&](const bf16_t* ql, const bf16_t* dol, int qb, const f32x16_t& p, const f32x16_t& ds) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const bf16x8_t pf = pack_frag(p, 8 * cc);
            const bf16x8_t dsf = pack_frag(ds, 8 * cc);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = mfma32(frag_tr(dol, qb * 32 + 16 * cc, db * 32, lane), pf, dv[db]);
                dk[db] = mfma32(frag_tr(ql, qb * 32 + 16 * cc, db * 32, lane), dsf, dk[db]);
            }
        }
    };

    f32x16_t s0, dp0;   // carried: scores of (tile t, q-block 0)
    scores(qring, doring, 0, s0, dp0);
    int slot = 0;       // ring slot of tile t
    auto tile_body = [&](int t, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const bf16_t* ql = qring + slot * TILE_ELEMS;
        const bf16_t* dol = doring + slot * TILE_ELEMS;
        tile_load_buf(qrs, sq.s, (t + 2) * TILE, qoff, qr);     // loads past the last tile read zeros into a slot nobody uses
        tile_load_buf(dors, sdo.s, (t + 2) * TILE, dooff, dor);
        stat_load(t + 2);
        f32x16_t s1, dp1;
        scores(ql, dol, 1, s1, dp1);
        softmax_grad(s0, dp0, t * TILE, TAIL);
        accumulate(ql, dol, 0, s0, dp0);
        softmax_grad(s1, dp1, t * TILE + 32, TAIL);
        scores(qring + slot1 * TILE_ELEMS, doring + slot1 * TILE_ELEMS, 0, s0, dp0);
        accumulate(ql, dol, 1, s1, dp1);
        tile_store(qring + slot2 * TILE_ELEMS, qr);
        tile_store(doring + slot2 * TILE_ELEMS, dor);
        stat_store(slot2);
        slot = slot1;
        __syncthreads();
    };
    const bool ragged = (S & (TILE - 1)) != 0 && nt == nt_all;   // the ragged tile, if any, is the global last one
    const int nfull = ragged ? nt - 1 : nt;
    for (int t = tb; t < nfull; ++t) tile_body(t, std::false_type{});
    if (ragged) tile_body(nt - 1, std::true_type{});
    if (SPLIT) {
        float* pb = part + ((size_t)(vid - task0) * nsplit + chunk) * (2 * WG_ROWS * HD);
        const int r = wave * 32 + (lane & 31);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t wk = {dk[db][4 * g], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]};
                const f32x4_t wv = {dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]};
                *reinterpret_cast<f32x4_t*>(pb + r * HD + db * 32 + 8 * g + 4 * hi) = wk;
                *reinterpret_cast<f32x4_t*>(pb + WG_ROWS * HD + r * HD + db * 32 + 8 * g + 4 * hi) = wv;
            }
        return;
    }
    const int k = k0 + (lane & 31);
    if (k < S) {
        bf16_t* kp = dK + ((size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)k * sdk.s);
        bf16_t* vp = dV + ((size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)k * sdv.s);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t w;
                w[0] = pack_bf16x2(dk[db][4 * g] * kscale, dk[db][4 * g + 1] * kscale);
                w[1] = pack_bf16x2(dk[db][4 * g + 2] * kscale, dk[db][4 * g + 3] * kscale);
                *reinterpret_cast<u32x2_t*>(kp + db * 32 + 8 * g + 4 * hi) = w;
                w[0] = pack_bf16x2(dv[db][4 * g], dv[db][4 * g + 1]);
                w[1] = pack_bf16x2(dv[db][4 * g + 2], dv[db][4 * g + 3]);
                *reinterpret_cast<u32x2_t*>(vp + db * 32 + 8 * g + 4 * hi) = w;
            }
    }
}

// sum the query-range chunks of the split dK/dV tasks: one wave per key row, lane = d
__global__ __launch_bounds__(256) void attn_dkv_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_kt, bf16_t* __restrict__ dK,
                                                               bf16_t* __restrict__ dV, TStride sdk, TStride sdv, int S, int H, float kscale) {
    const int lane = threadIdx.x & 63;
    const int r = ((int)blockIdx.x % (WG_ROWS / 4)) * 4 + (threadIdx.x >> 6), tl = (int)blockIdx.x / (WG_ROWS / 4);
    const int vid = task0 + tl, bh = vid / n_kt, kt = vid % n_kt;
    const int key = kt * WG_ROWS + r;
    if (key >= S) return;
    const float* pb = part + (size_t)tl * nsplit * (2 * WG_ROWS * HD) + r * HD + lane;
    float ak = 0.f, av = 0.f;
    for (int c = 0; c < nsplit; ++c) {
        ak += pb[(size_t)c * (2 * WG_ROWS * HD)];
        av += pb[(size_t)c * (2 * WG_ROWS * HD) + WG_ROWS * HD];
    }
    const int b = bh / H, h = bh % H;
    dK[(size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)key * sdk.s + lane] = f32_to_bf16(ak * kscale);
    dV[(size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)key * sdv.s + lane] = f32_to_bf16(av);
}

#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_bwd_fused_kernel.inc"
#endif  // VGPA_VARIANTS

#ifndef DQ_QB
#define DQ_QB 2    // query blocks (of 32 rows) per wave in the dQ kernel
#endif
#ifndef FWD_NW
#define FWD_NW 4   // waves per workgroup in the forward kernel
#endif
#ifndef FWD_QB
#define FWD_QB 2   // query blocks (of 32 rows) per wave in the forward kernel
#endif
#define SOK(st) (stride_ok(st) && range_ok(st, B, H, S))

// Redo pass behind the w1 forward (attention_w1.hip): the online-softmax kernel over every 256-row strip whose flag is set.
int32_t vgpa_internal_attn_fwd_redo(const void* q, const void* k, const void* v, void* o, float* lse2, TStride sq, TStride sk, TStride sv, TStride so,
                                    int S, int H, int n_qt, int64_t tasks, const int* flags, hipStream_t stream, void* o_res, TStride sor, int res_kind) {
    VGPA_LAUNCH((attn_fwd_pipe_kernel<2, 4, false>), dim3((unsigned)tasks), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                (bf16_t*)o, lse2, sq, sk, sv, so, S, H, n_qt, 0, 1, (float*)nullptr, flags, o_res, sor, o_res ? res_kind : VGPA_RES_NONE);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

extern "C" {

// All tensors are bf16 views [B, H, S, 64] given by element strides {batch, head, token} (last dim contiguous,
// strides multiples of 8, base pointers 16-byte aligned).  lse2 / delta are fp32 [B, H, S] contiguous.
// CONTRACT: q holds the queries PRE-MULTIPLIED by scale*log2(e) (vgpa_qknorm_rope_fwd writes them that way through
// q_out_scale), in all four entry points; `scale` is still the softmax scale (used for the dQ / dK multipliers).
// dq is the gradient w.r.t. the UNscaled query.
#define FWD_MAX_SPLIT 16

size_t vgpa_attn_fwd_workspace_bytes(int64_t B, int64_t H, int64_t S) {
    // worst case of split_plan: (leftover tasks) x (chunks) <= slots in automatic mode; forced mode (tests) splits every task
    const int64_t n_qt = (S + 255) / 256, tasks = n_qt * B * H;
    int64_t parts = wg_slots();
    if (tasks * FWD_MAX_SPLIT < parts) parts = tasks * FWD_MAX_SPLIT;
    return (size_t)parts * FWD_PART_FLOATS * sizeof(float);
}

static int32_t attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                             const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H, int64_t S,
                             int64_t head_dim, int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !o || !lse2 || head_dim != HD || B <= 0 || H <= 0 || S <= 0 || S > (1 << 24)) return VGPA_ERR_INVALID;
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(o_strides)) return VGPA_ERR_INVALID;
    if (!al16(q) || !al16(k) || !al16(v) || !al16(o)) return VGPA_ERR_INVALID;
#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_fwd_pp_dispatch.inc"
#endif  // VGPA_VARIANTS
    const int n_qt = (int)((S + FWD_QB * FWD_NW * 32 - 1) / (FWD_QB * FWD_NW * 32));
    const int64_t nblk = (int64_t)n_qt * B * H;
    if (nblk > 0x7fffffff) return VGPA_ERR_INVALID;
#ifndef FWD_V1   // product path: the software-pipelined kernel; -DFWD_V1 builds the three-block kernel (diagnostic hooks live there)
    if (FWD_NW * FWD_QB == 8) {   // 256 query rows per task either way
        int64_t n_main = nblk;
        int nsplit = 1;
        if (workspace) split_plan(nblk, (int)((S + TILE - 1) / TILE), split_mode, FWD_MAX_SPLIT, &n_main, &nsplit);
        const int64_t n_tail = nblk - n_main;
        if (n_tail > 0 && ws_bytes < (size_t)n_tail * nsplit * FWD_PART_FLOATS * sizeof(float)) {
            if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
            n_main = nblk;   // automatic mode: fall back to the single launch
        }
        if (n_main > 0) {
            VGPA_LAUNCH((attn_fwd_pipe_kernel<FWD_QB, FWD_NW, false>), dim3((unsigned)n_main), dim3(64 * FWD_NW), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                        (const bf16_t*)v, (bf16_t*)o, lse2, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S, (int)H, n_qt, 0, 1,
                        (float*)nullptr);
            VGPA_CHECK_LAUNCH();
        }
        if (n_main < nblk) {
            VGPA_LAUNCH((attn_fwd_pipe_kernel<FWD_QB, FWD_NW, true>), dim3((unsigned)(n_tail * nsplit)), dim3(64 * FWD_NW), 0, stream, (const bf16_t*)q,
                        (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S,
                        (int)H, n_qt, (int)n_main, nsplit, (float*)workspace);
            VGPA_CHECK_LAUNCH();
            VGPA_LAUNCH(attn_fwd_merge_kernel, dim3((unsigned)(n_tail * 64)), dim3(256), 0, stream, (const float*)workspace, nsplit, (int)n_main, n_qt,
                        (bf16_t*)o, mk(o_strides), lse2, (int)S, (int)H);
            VGPA_CHECK_LAUNCH();
        }
        return VGPA_OK;
    }
#endif
#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_fwd_v1_dispatch.inc"
#else
    return VGPA_ERR_INVALID;   // other blockings exist in variant builds only
#endif
}

int32_t vgpa_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                      const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H, int64_t S,
                      int64_t head_dim, float scale, hipStream_t stream) {
    (void)scale;
    return attn_fwd_impl(q, k, v, o, lse2, q_strides, k_strides, v_strides, o_strides, B, H, S, head_dim, 0, nullptr, 0, stream);
}

// Same, with a workspace (vgpa_attn_fwd_workspace_bytes) that lets the launcher cut the leftover tasks of a partially filled
// last scheduling round into key-range chunks (a second small launch + a merge).  split_mode: -1 automatic, 0 never, k >= 2
// force k chunks for every task.
int32_t vgpa_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                         const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H, int64_t S,
                         int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    (void)scale;
    if (workspace && !al16(workspace)) return VGPA_ERR_INVALID;
    return attn_fwd_impl(q, k, v, o, lse2, q_strides, k_strides, v_strides, o_strides, B, H, S, head_dim, split_mode, workspace, ws_bytes, stream);
}

// workspace: fp32 delta [B,H,S]  (vgpa_attn_bwd_workspace_bytes)
size_t vgpa_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t S) { return (size_t)B * H * S * sizeof(float); }

static inline bool bwd_common_ok(int64_t B, int64_t H, int64_t S, int64_t head_dim) {
    return head_dim == HD && B > 0 && H > 0 && S > 0 && S <= (1 << 24) && (int64_t)((S + WG_ROWS - 1) / WG_ROWS) * B * H <= 0x7fffffff;
}

// step 1 of the backward: delta[b,h,q] = sum_d dO * O  (_res: of the output as the forward's residual tensor completes it -- vgpa_attn_fwd_w1_res,
// res_kind VGPA_RES_BF16 / VGPA_RES_8; o_res may be NULL)
int32_t vgpa_attn_bwd_delta_res(const void* o, const void* o_res, int32_t res_kind, const void* d_o, const int64_t* o_strides, const int64_t* ores_strides,
                                const int64_t* do_strides, float* delta, int64_t B, int64_t H, int64_t S, int64_t head_dim, hipStream_t stream) {
    if (!o || !d_o || !delta || !bwd_common_ok(B, H, S, head_dim) || !SOK(o_strides) || !SOK(do_strides) || !al16(o) || !al16(d_o))
        return VGPA_ERR_INVALID;
    if (o_res && (!SOK(ores_strides) || !al16(o_res) || (res_kind != VGPA_RES_BF16 && res_kind != VGPA_RES_8))) return VGPA_ERR_INVALID;
    const int64_t total = B * H * S;
    VGPA_LAUNCH(attn_delta_kernel, dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)d_o, (const bf16_t*)o,
                       mk(do_strides), mk(o_strides), (int)S, (int)H, total, delta, o_res, o_res ? mk(ores_strides) : mk(o_strides),
                       o_res ? (int)res_kind : VGPA_RES_NONE);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
int32_t vgpa_attn_bwd_delta(const void* o, const void* d_o, const int64_t* o_strides, const int64_t* do_strides, float* delta, int64_t B,
                            int64_t H, int64_t S, int64_t head_dim, hipStream_t stream) {
    return vgpa_attn_bwd_delta_res(o, nullptr, VGPA_RES_NONE, d_o, o_strides, nullptr, do_strides, delta, B, H, S, head_dim, stream);
}

// step 2: dK, dV (workgroup per 128 keys).  With a workspace the leftover tasks of a mostly empty last scheduling round are
// cut into query-range chunks (split_plan); split_mode as in vgpa_attn_fwd_ws.
#define BWD_MAX_SPLIT 16
#define DKV_PART_FLOATS (2 * WG_ROWS * HD)
#define DQ_PART_FLOATS (DQ_QB * WG_ROWS * HD)
static int32_t attn_bwd_dkv_impl(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dk,
                                 void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                 const int64_t* do_strides, const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H, int64_t S,
                                 int64_t head_dim, int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !d_o || !lse2 || !delta || !dk || !dv || !bwd_common_ok(B, H, S, head_dim)) return VGPA_ERR_INVALID;
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(do_strides) || !SOK(dk_strides) ||
        !SOK(dv_strides) || !al16(q) || !al16(k) || !al16(v) || !al16(d_o) || !al16(dk) || !al16(dv) || (workspace && !al16(workspace)))
        return VGPA_ERR_INVALID;
    const int n_t = (int)((S + WG_ROWS - 1) / WG_ROWS);
    const int64_t tasks = (int64_t)n_t * B * H;
    const float kscale = 0.6931471805599453f;
    int64_t n_main = tasks;
    int nsplit = 1;
    if (workspace) split_plan(tasks, (int)((S + TILE - 1) / TILE), split_mode, BWD_MAX_SPLIT, &n_main, &nsplit);
    const int64_t n_tail = tasks - n_main;
    if (n_tail > 0 && ws_bytes < (size_t)n_tail * nsplit * DKV_PART_FLOATS * sizeof(float)) {
        if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
        n_main = tasks;
    }
    if (n_main > 0) {
        VGPA_LAUNCH(attn_bwd_dkv_kernel<false>, dim3((unsigned)n_main), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dk_strides), mk(dv_strides), (int)S, (int)H, n_t, kscale, 0, 1, (float*)nullptr);
        VGPA_CHECK_LAUNCH();
    }
    if (n_main < tasks) {
        VGPA_LAUNCH(attn_bwd_dkv_kernel<true>, dim3((unsigned)(n_tail * nsplit)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, mk(q_strides), mk(k_strides), mk(v_strides),
                    mk(do_strides), mk(dk_strides), mk(dv_strides), (int)S, (int)H, n_t, kscale, (int)n_main, nsplit, (float*)workspace);
        VGPA_CHECK_LAUNCH();
        VGPA_LAUNCH(attn_dkv_merge_kernel, dim3((unsigned)(n_tail * (WG_ROWS / 4))), dim3(256), 0, stream, (const float*)workspace, nsplit, (int)n_main,
                    n_t, (bf16_t*)dk, (bf16_t*)dv, mk(dk_strides), mk(dv_strides), (int)S, (int)H, kscale);
        VGPA_CHECK_LAUNCH();
    }
    return VGPA_OK;
}

// step 3: dQ (workgroup per 128 * DQ_QB queries); split as above, along the key tiles
static int32_t attn_bwd_dq_impl(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dq,
                                const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                                const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode,
                                void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !d_o || !lse2 || !delta || !dq || !bwd_common_ok(B, H, S, head_dim)) return VGPA_ERR_INVALID;
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(do_strides) || !SOK(dq_strides) ||
        !al16(q) || !al16(k) || !al16(v) || !al16(d_o) || !al16(dq) || (workspace && !al16(workspace)))
        return VGPA_ERR_INVALID;
    const int n_t = (int)((S + DQ_QB * WG_ROWS - 1) / (DQ_QB * WG_ROWS));
    const int64_t tasks = (int64_t)n_t * B * H;
    int64_t n_main = tasks;
    int nsplit = 1;
    if (workspace) split_plan(tasks, (int)((S + TILE - 1) / TILE), split_mode, BWD_MAX_SPLIT, &n_main, &nsplit);
    const int64_t n_tail = tasks - n_main;
    if (n_tail > 0 && ws_bytes < (size_t)n_tail * nsplit * DQ_PART_FLOATS * sizeof(float)) {
        if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
        n_main = tasks;
    }
    if (n_main > 0) {
        VGPA_LAUNCH((attn_bwd_dq_kernel<DQ_QB, false>), dim3((unsigned)n_main), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dq, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dq_strides), (int)S, (int)H, n_t, scale, 0, 1, (float*)nullptr);
        VGPA_CHECK_LAUNCH();
    }
    if (n_main < tasks) {
        VGPA_LAUNCH((attn_bwd_dq_kernel<DQ_QB, true>), dim3((unsigned)(n_tail * nsplit)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dq, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dq_strides), (int)S, (int)H, n_t, scale, (int)n_main, nsplit, (float*)workspace);
        VGPA_CHECK_LAUNCH();
        VGPA_LAUNCH(attn_dq_merge_kernel, dim3((unsigned)(n_tail * (DQ_QB * WG_ROWS / 4))), dim3(256), 0, stream, (const float*)workspace, nsplit,
                    (int)n_main, n_t, DQ_QB * WG_ROWS, (bf16_t*)dq, mk(dq_strides), (int)S, (int)H, scale);
        VGPA_CHECK_LAUNCH();
    }
    return VGPA_OK;
}

int32_t vgpa_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dk,
                          void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                          const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale,
                          hipStream_t stream) {
    (void)scale;
    return attn_bwd_dkv_impl(q, k, v, d_o, lse2, delta, dk, dv, q_strides, k_strides, v_strides, do_strides, dk_strides, dv_strides, B, H, S,
                             head_dim, 0, nullptr, 0, stream);
}
int32_t vgpa_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dq,
                         const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                         const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, hipStream_t stream) {
    return attn_bwd_dq_impl(q, k, v, d_o, lse2, delta, dq, q_strides, k_strides, v_strides, do_strides, dq_strides, B, H, S, head_dim, scale, 0,
                            nullptr, 0, stream);
}

// Workspace for the split forms below (shared by the two; they run one after the other on a stream).
size_t vgpa_attn_bwd_split_workspace_bytes(int64_t B, int64_t H, int64_t S) {
    const int64_t t_dkv = (S + WG_ROWS - 1) / WG_ROWS * B * H, t_dq = (S + DQ_QB * WG_ROWS - 1) / (DQ_QB * WG_ROWS) * B * H;
    int64_t p_dkv = wg_slots(), p_dq = wg_slots();
    if (t_dkv * BWD_MAX_SPLIT < p_dkv) p_dkv = t_dkv * BWD_MAX_SPLIT;
    if (t_dq * BWD_MAX_SPLIT < p_dq) p_dq = t_dq * BWD_MAX_SPLIT;
    const size_t a = (size_t)p_dkv * DKV_PART_FLOATS * sizeof(float), b = (size_t)p_dq * DQ_PART_FLOATS * sizeof(float);
    return a > b ? a : b;
}
int32_t vgpa_attn_bwd_dkv_ws(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dk,
                             void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                             const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale,
                             int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    (void)scale;
    return attn_bwd_dkv_impl(q, k, v, d_o, lse2, delta, dk, dv, q_strides, k_strides, v_strides, do_strides, dk_strides, dv_strides, B, H, S,
                             head_dim, split_mode, workspace, ws_bytes, stream);
}
int32_t vgpa_attn_bwd_dq_ws(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dq,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                            const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode,
                            void* workspace, size_t ws_bytes, hipStream_t stream) {
    return attn_bwd_dq_impl(q, k, v, d_o, lse2, delta, dq, q_strides, k_strides, v_strides, do_strides, dq_strides, B, H, S, head_dim, scale,
                            split_mode, workspace, ws_bytes, stream);
}

#ifdef VGPA_VARIANTS   // measured-slower experiments / diagnostics live in tools/variants/ (variant builds only)
#include "attn_bwd_fused_entry.inc"
#endif  // VGPA_VARIANTS

// the whole backward (delta -> dK/dV -> dQ) with a caller-provided workspace for delta
int32_t vgpa_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                      void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides,
                      const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides, int64_t B,
                      int64_t H, int64_t S, int64_t head_dim, float scale, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!workspace) return VGPA_ERR_INVALID;
    if (!bwd_common_ok(B, H, S, head_dim)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_attn_bwd_workspace_bytes(B, H, S)) return VGPA_ERR_WORKSPACE;
    float* delta = (float*)workspace;
    int32_t rc = vgpa_attn_bwd_delta(o, d_o, o_strides, do_strides, delta, B, H, S, head_dim, stream);
    if (rc) return rc;
    rc = vgpa_attn_bwd_dkv(q, k, v, d_o, lse2, delta, dk, dv, q_strides, k_strides, v_strides, do_strides, dk_strides, dv_strides, B, H, S,
                           head_dim, scale, stream);
    if (rc) return rc;
    return vgpa_attn_bwd_dq(q, k, v, d_o, lse2, delta, dq, q_strides, k_strides, v_strides, do_strides, dq_strides, B, H, S, head_dim, scale,
                            stream);
}

}  // extern "C"
