// "w1" attention kernels for gfx950: one wave per SIMD (a 4-wave workgroup owns a CU), the whole 512-entry register file per
// wave, streamed tiles by LDS-DMA.  Same mathematics, operand conventions and C ABI as attention.hip (see the header there);
// what changes is the blocking:
//   * __launch_bounds__(256, 1): accumulators and the stationary operand fragments (MFMA-only values) may live in the
//     accumulator half of the register file, which leaves the 256 architectural VGPRs to a software pipeline that is one
//     32-row half-tile deep in every stage:   scores(g+1)  ||  exp / multiply / pack (g)  ||  accumulate (g-1),
//     the stage the 2-waves-per-SIMD kernels could not hold in 256 registers (the round-1/2 analysis: profiles/HISTORY.md).
//   * K/V (Q/dO) tiles never pass through VGPRs: each wave issues `buffer_load_dwordx4 ... lds` pieces two tiles ahead
//     into a 4-slot ring (attn_w1.h), completion counted by hand (s_waitcnt vmcnt(N)), ONE s_barrier per 64-row tile.
//   * the LDS image is unpadded and chunk-swizzled: row-fragment reads and transpose reads are both conflict-free.
#include "attn_common.h"

#include <type_traits>

#include "attn_w1.h"

// =====================================================================================================
// Backward, dQ:  dQ = scale * sum_k dS[q,k] K[k],  dS = P o (dP - delta),  P = exp2(c*s - lse2),  dP = dO V^T
// (S^T and dP^T are made per 32-key half-tile with q on the MFMA columns; -lse2 and -delta ride an extra k-step.)
// The main loop is tools/gen_w1_asm.py::DqLoop (w1_dq_loop.inc): read its docstring for the pipeline and the register map.
// =====================================================================================================
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16_t;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;

__device__ __forceinline__ u32x16_t pack4(const bf16x8_t& a, const bf16x8_t& b, const bf16x8_t& c, const bf16x8_t& d) {
    const u32x4_t w[4] = {__builtin_bit_cast(u32x4_t, a), __builtin_bit_cast(u32x4_t, b), __builtin_bit_cast(u32x4_t, c), __builtin_bit_cast(u32x4_t, d)};
    u32x16_t r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = w[i >> 2][i & 3];
    return r;
}

// SPLIT: workgroup (task0 + blockIdx / nsplit, chunk blockIdx % nsplit) sweeps key tiles [nt*chunk/nsplit, nt*(chunk+1)/nsplit)
// and leaves its unscaled fp32 dQ [256][64] in `part` (attn_dq_merge_kernel of attention.hip adds the chunks).
template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                  const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                  const float* __restrict__ LSE2, const float* __restrict__ DELTA,
                                                                  bf16_t* __restrict__ dQ, TStride sq, TStride sk, TStride sv, TStride sdo,
                                                                  TStride sdq, int S, int H, int n_qt, float scale, int task0, int nsplit,
                                                                  float* __restrict__ part) {
    constexpr int QB = 2;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1_RING_BYTES];   // slot = [K tile | V tile]
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * (32 * QB);

    // stationary operands first, and waited for, so that no compiler-counted VMEM operation is in flight next to the LDS-DMA
    bf16x8_t qf[QB][4], dof[QB][4];
    f32x16_t cs[QB], cd[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        load_row_frags(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0 + 32 * j, S, lane, qf[j]);
        load_row_frags(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h), sdo.s, q0 + 32 * j, S, lane, dof[j]);
        int qc = q0 + 32 * j + (lane & 31);
        qc = qc < S ? qc : S - 1;
        // -lse2[q] and -delta[q] enter the score chains as the srcC of their first k-step (every accumulator row of column q)
        const float nl = -LSE2[(int64_t)bh * S + qc], nd = -DELTA[(int64_t)bh * S + qc];
#pragma unroll
        for (int i = 0; i < 16; ++i) { cs[j][i] = nl; cd[j][i] = nd; }
    }
#pragma unroll
    for (int j = 0; j < QB; ++j) { frags_arrived(qf[j]); frags_arrived(dof[j]); }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's key tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;

    // the pipeline's first transposed-K reads hit the slot "before" tile tb (ring slot 3): make it finite
    {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4_t*>(lds + 3 * W1_SLOT_BYTES + threadIdx.x * 16) = z;
        *reinterpret_cast<u32x4_t*>(lds + 3 * W1_SLOT_BYTES + 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    const bf16_t* Vb = V + ((size_t)b * sv.b + (size_t)h * sv.h);
    const W1Rsrc krs = w1_rsrc(Kb, ((uint32_t)(S - 1) * sk.s + (uint32_t)HD) * 2u);
    const W1Rsrc vrs = w1_rsrc(Vb, ((uint32_t)(S - 1) * sv.s + (uint32_t)HD) * 2u);
    uint32_t kvo[2], vvo[2];
    w1_dma_offsets<2>(wave, lane, sk.s, kvo);
    w1_dma_offsets<2>(wave, lane, sv.s, vvo);
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * sk.s * 2u), vstep = __builtin_amdgcn_readfirstlane(64u * sv.s * 2u);   // bytes per tile
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 2048u);
    // The tile offset rides in the per-lane offset (not in the scalar offset): the descriptor's range check must see it, so
    // that rows at or past S -- and whole tiles past the end -- arrive as zeros.
    u32x4_t voff = {kvo[0] + (uint32_t)tb * kstep, kvo[1] + (uint32_t)tb * kstep, vvo[0] + (uint32_t)tb * vstep, vvo[1] + (uint32_t)tb * vstep};
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // tiles tb, tb + 1 -> ring slots 0, 1
        const uint32_t dst = wbase + (uint32_t)i * W1_SLOT_BYTES;
        w1_dma(dst, krs, voff[0], 0u);
        w1_dma(dst + 1024u, krs, voff[1], 0u);
        w1_dma(dst + W1_TILE_BYTES, vrs, voff[2], 0u);
        w1_dma(dst + W1_TILE_BYTES + 1024u, vrs, voff[3], 0u);
        voff[0] += kstep; voff[1] += kstep; voff[2] += vstep; voff[3] += vstep;
    }

    const W1Lane la = w1_lane_offsets(lane);
    const u32x8_t la8 = {la.row[0], la.row[1], la.row[2], la.row[3], la.tr[0][0], la.tr[0][1], la.tr[1][0], la.tr[1][1]};
    const u32x16_t qf0 = pack4(qf[0][0], qf[0][1], qf[0][2], qf[0][3]), qf1 = pack4(qf[1][0], qf[1][1], qf[1][2], qf[1][3]);
    const u32x16_t do0 = pack4(dof[0][0], dof[0][1], dof[0][2], dof[0][3]), do1 = pack4(dof[1][0], dof[1][1], dof[1][2], dof[1][3]);
    const uint32_t niter = (uint32_t)(nt - tb + 1);   // one extra tile step drains the pipeline
    f32x16_t dq[QB][2];
    uint32_t t0, t1, t2;
    asm volatile(
#include "w1_dq_loop.inc"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "={a[0:15]}"(dq[0][0]), "={a[16:31]}"(dq[0][1]), "={a[32:47]}"(dq[1][0]), "={a[48:63]}"(dq[1][1]),
          "+{v[232:235]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), "{a[64:79]}"(qf0),
          "{a[80:95]}"(qf1), "{a[96:111]}"(do0), "{a[112:127]}"(do1), "{v[160:175]}"(cs[0]), "{v[176:191]}"(cs[1]), "{v[192:207]}"(cd[0]), "{v[208:223]}"(cd[1]), "{v[224:231]}"(la8)
        : "memory", "scc",
#include "w1_dq_clobbers.inc"
    );
    // bring the accumulators over to the architectural registers as whole tuples (element reads straight off the asm's
    // physical AGPR outputs make hipcc 7.2 emit illegal V_MOVs)
#pragma unroll
    for (int j = 0; j < QB; ++j) { asm volatile("" : "+v"(dq[j][0])); asm volatile("" : "+v"(dq[j][1])); }

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
                for (int gp = 0; gp < 2; ++gp) {          // 16-byte stores: common.h pair_rows8
                    u32x2_t w[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * gp + e;
                        w[e][0] = pack_bf16x2(dq[j][db][4 * g] * scale, dq[j][db][4 * g + 1] * scale);
                        w[e][1] = pack_bf16x2(dq[j][db][4 * g + 2] * scale, dq[j][db][4 * g + 3] * scale);
                    }
                    *reinterpret_cast<u32x4_t*>(op + db * 32 + 8 * (2 * gp + hi)) = pair_rows8(w[0], w[1]);
                }
        }
    }
}

// =====================================================================================================
// Forward:  O = softmax(scale * Q K^T) V ;  lse2 = log2 sum_k exp2(scale*log2e * q.k)        (q arrives pre-scaled by scale*log2e)
// The main loop is tools/gen_w1_asm.py::FwdLoop (w1_fwd_loop.inc).  Scores are shifted by M[q] = |q| * max_k |k| (>= every score of
// the row), so the loop needs no running maximum.  A row whose true maximum lies more than ~100 (log2) below that bound would
// underflow, and a bound above W1_M_MAX would cost precision (the fp32 accumulators start at -M): such strips are flagged
// (flags[task] = 1) and redone by the online-softmax kernel of attention.hip
// (vgpa_internal_attn_fwd_redo), so the result never depends on the bound being tight.
// =====================================================================================================
#include "w1_fwd_knobs.inc"                   // W1_FWD_MFSUM: what the generated loop expects around it (tools/gen_w1_asm.py W1_KNOBS; 0 in the product)
#define W1_FWD_PART_FLOATS (256 * (HD + 2))   // per (task, chunk): O[256][64] (un-normalised), M[256], l[256] -- layout of attention.hip's split forward
#define W1_L_MIN 7.8886e-31f                  // 2^-100: below this the row's sum is too close to underflow -> redo
// ... and above 2^118 too close to overflow: the O accumulators carry sum_j p_j v_j <= l max|v| (a row whose true maximum lies 112-128 above the shift has a FINITE
// l next to O = +-inf, and 1 / l flushes to zero from 2^126 on).  Found by `bench.py --weights trained_like` (one row of block 38, true maximum 127.7 above M':
// l = 2^127.7, O = inf, strip not flagged -> NaN loss; tools/attn_fault_repro.py).  2^118 leaves |v| < 2^10 before an accumulator overflows, and the epilogue checks
// the accumulators themselves (oabs) for whatever |v| the caller brings; a first cut at 2^100 moved the cliff of tools/attn_robust.py in by 18 log2 units of row
// maximum for nothing (gain 4: 11 % -> 46 % of the strips redone).
#define W1_L_MAX 3.3230699e35f                // 2^118
// With the shift at the row BOUND the largest weight of a row is exp2(s_max - M), not 1, so it carries a bf16 rounding error in the
// numerator (2^-9 relative) that the fp32 denominator does not share; over a few dozen keys these errors average out, over one or
// two they do not (S = 1: O off by up to 0.4 %).  Rows that short are not a performance case: below this length every strip goes to
// the online-softmax kernel.
#define W1_FWD_MIN_S 128
// The shift M' only has to put exp2(s - M') inside fp32's range for every score that matters, it does not have to be an upper bound: the weights go to the matrix
// pipe as bf16 (fp32's exponent range) and l, O accumulate in fp32.  M'[q] = min(b[q], m_s[q] + W1_SAMPLE_UP) with b = |q| max|k| (Cauchy-Schwarz, >= every score) and
// m_s = the row's maximum over W1_SAMPLE_KEYS keys spread evenly over the sequence (16 MFMAs per wave in the prologue: 0.2 % of the sweep), a LOWER bound of the true
// maximum m*.  Then  M' - m* <= M' - m_s <= 64  always (nothing that matters underflows: terms below 2^-62 of the row's largest are dropped), and  m* - M' <= 112
// (no overflow: l <= S 2^112 < 2^127) whenever b - m_s <= 176 or, beyond that, whenever the true maximum is not more than 176 log2 units above the sampled one.  A row
// outside (an extreme outlier key the sample missed) makes l = inf: the strip is flagged and redone by the online-softmax kernel, as before.
// Round 5 shifted by b itself (p <= 1) and flagged every strip whose maximum lay > 100 below b or whose b exceeded 160: a QK-norm gain of 2.5 with a few outlier
// channels (row entropy < 1 bit) sent the whole launch to the redo kernel, 2.5-3 x the time (tools/attn_robust.py, profiles/r06*_attn_trained_like.*).
#define W1_SAMPLE_KEYS 64
#define W1_SAMPLE_UP 64.0f
// the scores are accumulated on top of -M' in fp32: at |M'| = 1024 the accumulator's ulp is 2^-13 log2 units = 8e-5 relative in a weight, a fiftieth of the bf16
// rounding the weight gets anyway (round 5 flagged every strip above 160: a QK-norm gain of 3.8 already sent the whole launch to the online-softmax kernel)
#define W1_M_MAX 1024.0f

// The rounding residual of four outputs, x - bf16(x), itself as bf16 (relative error 2^-9 of a quantity that is 2^-9 of x: the pair (o, o_res)
// carries O to ~2^-17).  Consumer: the backward's delta = rowsum(dO o (O + O_res)) (w1_bwd_prep_kernel / attn_delta_kernel).  Why: delta stands for
// rowsum(P o dP), an identity that holds for the UNROUNDED O = P V only; formed from the bf16 O alone every row's dS stops summing to zero and dQ_i
// picks up -d(delta_i) sum_j P_ij K_j, a coherent error that swamps q / k gradients which are small by cancellation (measured at BASELINE configs[0]
// width: 37-87 % of the last block's to_q / to_k adapter gradients, in this path and in plain torch bf16 alike; tools/cfg1_round_diag.py).
__device__ __forceinline__ u32x2_t w1_residual4(const float* x, u32x2_t packed) {
    u32x2_t r;
    r[0] = pack_bf16x2(x[0] - __uint_as_float(packed[0] << 16), x[1] - __uint_as_float(packed[0] & 0xffff0000u));
    r[1] = pack_bf16x2(x[2] - __uint_as_float(packed[1] << 16), x[3] - __uint_as_float(packed[1] & 0xffff0000u));
    return r;
}

// max_k |k| per (batch, head): kmax2[bh] = max over keys of sum_d k^2 (fp32 bits compared as integers: non-negative floats)
__global__ __launch_bounds__(256) void w1_kmax_kernel(const bf16_t* __restrict__ K, TStride sk, int S, int H, unsigned* __restrict__ kmax2) {
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)S * 8; i += (int64_t)gridDim.x * 256) {   // 8 lanes per row
        const int row = (int)(i >> 3), c8 = (int)(i & 7);
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(Kb + ((size_t)row * sk.s + c8 * 8)), f);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += f[j] * f[j];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax(kmax2 + bh, __float_as_uint(mx));
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void attn_fwd_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               bf16_t* __restrict__ O, float* __restrict__ LSE2, const unsigned* __restrict__ KMAX2,
                                                               int* __restrict__ flags, TStride sq, TStride sk, TStride sv, TStride so, int S, int H,
                                                               int n_qt, int task0, int nsplit, float* __restrict__ part, void* __restrict__ ORES,
                                                               TStride sor, int res_kind) {
    constexpr int QB = 2;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1_RING_BYTES];   // slot = [K tile | V tile]
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_qt, qt = vid % n_qt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int q0 = (qt * 4 + wave) * (32 * QB);

    bf16x8_t qf[QB][4];
    float nm[QB];   // -M[q]
    const float kmax = sqrtf(__uint_as_float(KMAX2[bh]));
#pragma unroll
    for (int j = 0; j < QB; ++j) load_row_frags(Q + ((size_t)b * sq.b + (size_t)h * sq.h), sq.s, q0 + 32 * j, S, lane, qf[j]);
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        frags_arrived(qf[j]);
        float a = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float f[8];
            frag_to_f32(qf[j][ks], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) a += f[i] * f[i];
        }
        a += other_half(a);                              // the row's other 32 columns live in lane ^ 32
        nm[j] = sqrtf(a) * kmax * 1.0009765625f;         // b[q], a hair above |q| |k|max: rounding of the bound itself can never let a score exceed it
    }
    {   // m_s[q]: the row's maximum over W1_SAMPLE_KEYS keys spread evenly over the sequence -- a LOWER bound of the true maximum (see W1_SAMPLE_UP)
        const bf16_t* Ks = K + ((size_t)b * sk.b + (size_t)h * sk.h);
        const uint32_t step = (uint32_t)S / W1_SAMPLE_KEYS;          // S >= W1_FWD_MIN_S = 128 here: step >= 2, the last sampled row is 63 step < S
        float ms[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) ms[j] = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < W1_SAMPLE_KEYS / 32; ++kb) {
            bf16x8_t kf[4];
            load_row_frags(Ks, sk.s * step, 32 * kb, W1_SAMPLE_KEYS, lane, kf);
            frags_arrived(kf);
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc = mfma32(kf[ks], qf[j][ks], acc);      // S^T[key][q]: this lane holds 16 keys of column q = lane & 31
                float m = acc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[i]);
                ms[j] = fmaxf(ms[j], fmaxf(m, other_half(m)));
            }
        }
#pragma unroll
        for (int j = 0; j < QB; ++j) nm[j] = -fminf(nm[j], ms[j] + W1_SAMPLE_UP);        // -M'[q]
    }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's key tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;

    {   // the pipeline's first transposed reads hit the V tile of the slot "before" tile tb (ring slot 3): make it finite
        const u32x4_t z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4_t*>(lds + 3 * W1_SLOT_BYTES + W1_TILE_BYTES + threadIdx.x * 16) = z;
        *reinterpret_cast<u32x4_t*>(lds + 3 * W1_SLOT_BYTES + W1_TILE_BYTES + 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const bf16_t* Kb = K + ((size_t)b * sk.b + (size_t)h * sk.h);
    const bf16_t* Vb = V + ((size_t)b * sv.b + (size_t)h * sv.h);
    const W1Rsrc krs = w1_rsrc(Kb, ((uint32_t)(S - 1) * sk.s + (uint32_t)HD) * 2u);
    const W1Rsrc vrs = w1_rsrc(Vb, ((uint32_t)(S - 1) * sv.s + (uint32_t)HD) * 2u);
    uint32_t kvo[2], vvo[2];
    w1_dma_offsets<2>(wave, lane, sk.s, kvo);
    w1_dma_offsets<2>(wave, lane, sv.s, vvo);
    const uint32_t kstep = __builtin_amdgcn_readfirstlane(64u * sk.s * 2u), vstep = __builtin_amdgcn_readfirstlane(64u * sv.s * 2u);
    const uint32_t wbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (uint32_t)wave * 2048u);
    u32x4_t voff = {kvo[0] + (uint32_t)tb * kstep, kvo[1] + (uint32_t)tb * kstep, vvo[0] + (uint32_t)tb * vstep, vvo[1] + (uint32_t)tb * vstep};
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // tiles tb, tb + 1 -> ring slots 0, 1
        const uint32_t dst = wbase + (uint32_t)i * W1_SLOT_BYTES;
        w1_dma(dst, krs, voff[0], 0u);
        w1_dma(dst + 1024u, krs, voff[1], 0u);
        w1_dma(dst + W1_TILE_BYTES, vrs, voff[2], 0u);
        w1_dma(dst + W1_TILE_BYTES + 1024u, vrs, voff[3], 0u);
        voff[0] += kstep; voff[1] += kstep; voff[2] += vstep; voff[3] += vstep;
    }

    const W1Lane la = w1_lane_offsets(lane);
    const u32x8_t la8 = {la.row[0], la.row[1], la.row[2], la.row[3], la.tr[0][0], la.tr[0][1], la.tr[1][0], la.tr[1][1]};
    const u32x16_t qf0 = pack4(qf[0][0], qf[0][1], qf[0][2], qf[0][3]), qf1 = pack4(qf[1][0], qf[1][1], qf[1][2], qf[1][3]);
    const uint32_t niter = (uint32_t)(nt - tb + 1);                       // one extra tile step drains the pipeline
    const int kend = nt * TILE < S ? nt * TILE : S;
    const uint32_t krem = (uint32_t)(kend - tb * TILE);                   // valid keys from tile tb on (of this chunk)
    const uint32_t hi4 = 4u * (uint32_t)hi;
#if W1_FWD_MFSUM   // row sums on the matrix pipe: the selector operand of the 16x16x32 products (1.0 pairs in lanes 0, 32 -> row 0 and 17, 49 -> row 1)
    const uint32_t sel = (lane == 0 || lane == 32 || lane == 17 || lane == 49) ? 0x3f803f80u : 0u;
#define W1_FWD_SEL_IN , "{v158}"(sel)
#else
#define W1_FWD_SEL_IN
#endif
    f32x16_t o[QB][2];
    u32x8_t lv;   // l[j][0..3]: four partial row sums per q-block (W1_FWD_MFSUM: the 16x16 accumulator of the selector product)
    uint32_t t0, t1, t2, t3;
    uint64_t c0, c1;   // s_memtime at the loop's start and end (read by tools/w1_clock.py through -DW1_CLOCKS builds)
    asm volatile(
#include "w1_fwd_loop.inc"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), [c0] "=&s"(c0), [c1] "=&s"(c1), "={a[0:15]}"(o[0][0]), "={a[16:31]}"(o[0][1]), "={a[32:47]}"(o[1][0]), "={a[48:63]}"(o[1][1]),
          "={v[128:135]}"(lv), "+{v[152:155]}"(voff)
        : [rk] "s"(krs.w), [rv] "s"(vrs.w), [kstep] "s"(kstep), [vstep] "s"(vstep), [wbase] "s"(wbase), [niter] "s"(niter), [krem] "s"(krem),
          "{a[64:79]}"(qf0), "{a[80:95]}"(qf1), "{v136}"(nm[0]), "{v137}"(nm[1]), "{v[144:151]}"(la8), "{v156}"(hi4) W1_FWD_SEL_IN
        : "memory", "scc", "vcc",
#include "w1_fwd_clobbers.inc"
    );
#pragma unroll
    for (int j = 0; j < QB; ++j) { asm volatile("" : "+v"(o[j][0])); asm volatile("" : "+v"(o[j][1])); }

    float l[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
#if W1_FWD_MFSUM   // D'[0][n'] = rowsum(q = n'), D'[1][n'] = rowsum(q = n' + 16): lanes 0-15 of the accumulator's first two registers
        const float s0 = __shfl(__uint_as_float(lv[4 * j]), lane & 15, 64), s1 = __shfl(__uint_as_float(lv[4 * j + 1]), lane & 15, 64);
        l[j] = (lane & 16) ? s1 : s0;
#else
        const float a = (__uint_as_float(lv[4 * j]) + __uint_as_float(lv[4 * j + 1])) + (__uint_as_float(lv[4 * j + 2]) + __uint_as_float(lv[4 * j + 3]));
        l[j] = a + other_half(a);   // the other 16 key rows of every 32-key block live in lane ^ 32
#endif
    }
    if (SPLIT) {   // partial result of this key range: un-normalised O (scaled by 2^-M), M, l
        float* pb = part + ((size_t)(vid - task0) * nsplit + chunk) * W1_FWD_PART_FLOATS;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const int r = wave * (32 * QB) + 32 * j + (lane & 31);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t w = {o[j][db][4 * g], o[j][db][4 * g + 1], o[j][db][4 * g + 2], o[j][db][4 * g + 3]};
                    *reinterpret_cast<f32x4_t*>(pb + r * HD + db * 32 + 8 * g + 4 * hi) = w;
                }
            if (hi == 0) { pb[256 * HD + r] = -nm[j]; pb[256 * HD + 256 + r] = l[j]; }
        }
        return;
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int q = q0 + 32 * j + (lane & 31);
        if (q < S) {
            float oabs = 0.f;                                 // inf / NaN in any accumulator of the row survives the sum (fmaxf would drop a NaN)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) oabs += fabsf(o[j][db][i]);
            bad = bad || !(l[j] >= W1_L_MIN && l[j] < W1_L_MAX) || !(-nm[j] <= W1_M_MAX) || !(oabs < INFINITY);
            const float inv = 1.f / l[j];
            bf16_t* op = O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s);
            const size_t ro = (size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s;      // residual row (elements of either kind)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {          // 16-byte stores of the output (8-byte ones of its res8 bytes): common.h pair_rows8
                    u32x2_t w[2], rw[2];
                    uint32_t rb[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * gp + e;
                        float x[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = o[j][db][4 * g + i] * inv;
                        w[e][0] = pack_bf16x2(x[0], x[1]);
                        w[e][1] = pack_bf16x2(x[2], x[3]);
                        if (res_kind == VGPA_RES_8) rb[e] = res8_pack4(x, w[e]);
                        else if (res_kind == VGPA_RES_BF16) rw[e] = w1_residual4(x, w[e]);
                    }
                    const int d0 = db * 32 + 8 * (2 * gp + hi);
                    *reinterpret_cast<u32x4_t*>(op + d0) = pair_rows8(w[0], w[1]);
                    if (res_kind == VGPA_RES_8) *reinterpret_cast<u32x2_t*>((uint8_t*)ORES + ro + d0) = pair_rows8_dword(rb[0], rb[1]);
                    else if (res_kind == VGPA_RES_BF16) *reinterpret_cast<u32x4_t*>((bf16_t*)ORES + ro + d0) = pair_rows8(rw[0], rw[1]);
                }
            if (hi == 0) LSE2[(int64_t)bh * S + q] = -nm[j] + __builtin_amdgcn_logf(l[j]);  // v_log_f32 is log2
        }
    }
#ifdef W1_CLOCKS   // diagnostic build: the redo flags carry the loop's cycle count instead (every strip is then redone -- results stay right)
    if (threadIdx.x == 0) flags[vid] = (int)(c1 - c0);
    (void)bad;
#else
    if (__any(bad) && lane == 0) flags[vid] = 1;
#endif
}

// combine the key-range chunks of the split forward tasks (all chunks share M): one wave per query row, lane = d
__global__ __launch_bounds__(256) void w1_fwd_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_qt, bf16_t* __restrict__ O, TStride so,
                                                             float* __restrict__ LSE2, int* __restrict__ flags, int S, int H, void* __restrict__ ORES,
                                                             TStride sor, int res_kind) {
    const int lane = threadIdx.x & 63, r = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6), tl = blockIdx.x >> 6;
    const int vid = task0 + tl, bh = vid / n_qt, qt = vid % n_qt;
    const int q = qt * 256 + r;
    if (q >= S) return;
    const float* pb = part + (size_t)tl * nsplit * W1_FWD_PART_FLOATS;
    float acc = 0.f, L = 0.f;
    for (int c = 0; c < nsplit; ++c) {
        const float* pc = pb + (size_t)c * W1_FWD_PART_FLOATS;
        acc += pc[r * HD + lane];
        L += pc[256 * HD + 256 + r];
    }
    const int b = bh / H, h = bh % H;
    const float x = acc / L;
    const bf16_t xb = f32_to_bf16(x);
    O[(size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s + lane] = xb;
    const size_t ro = (size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s + lane;
    if (res_kind == VGPA_RES_8) ((uint8_t*)ORES)[ro] = (uint8_t)res8_byte(x, xb);
    else if (res_kind == VGPA_RES_BF16) ((bf16_t*)ORES)[ro] = f32_to_bf16(x - bf16_to_f32(xb));
    const bool obad = __any(!(fabsf(acc) < INFINITY));      // the row's 64 un-normalised outputs live one per lane
    if (lane == 0) {
        LSE2[(int64_t)bh * S + q] = pb[256 * HD + r] + __builtin_amdgcn_logf(L);
        if (!(L >= W1_L_MIN && L < W1_L_MAX) || !(pb[256 * HD + r] <= W1_M_MAX) || obad) flags[vid] = 1;
    }
}

// =====================================================================================================
// Backward, dK / dV:  dV = P^T dO,  dK = scale * dS^T Q      (workgroup = 256 keys = 4 waves x 2 key blocks; streams Q | dO tiles)
// The main loop is tools/gen_w1_asm.py::DkvLoop (w1_dkv_loop.inc).  `stats` = fp32 [B, H, 2, S]: plane 0 = -lse2, plane 1 = -delta
// (w1_bwd_prep_kernel); the 64 rows' statistics of a tile travel next to it by LDS-DMA and enter the score chains as srcC.
// =====================================================================================================
#define W1_STAT_BYTES 1024   // per ring slot: 4 waves x (16 x -lse2 | 16 x -delta | 128 B unused)
template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_w1_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                   const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                   const float* __restrict__ STATS, bf16_t* __restrict__ dK, bf16_t* __restrict__ dV,
                                                                   TStride sq, TStride sk, TStride sv, TStride sdo, TStride sdk, TStride sdv, int S,
                                                                   int H, int n_kt, float kscale, int task0, int nsplit, float* __restrict__ part) {
    constexpr int KB = 2;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[W1_RING_BYTES + W1_SLOTS * W1_STAT_BYTES];   // slot = [Q tile | dO tile]; statistics behind the ring
    const int vid = task0 + (SPLIT ? (int)blockIdx.x / nsplit : xcd_remap(blockIdx.x, gridDim.x));
    const int chunk = SPLIT ? (int)blockIdx.x % nsplit : 0;
    const int bh = vid / n_kt, kt = vid % n_kt;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int k0 = (kt * 4 + wave) * (32 * KB);

    bf16x8_t kf[KB][4], vf[KB][4];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        load_row_frags(K + ((size_t)b * sk.b + (size_t)h * sk.h), sk.s, k0 + 32 * j, S, lane, kf[j]);
        load_row_frags(V + ((size_t)b * sv.b + (size_t)h * sv.h), sv.s, k0 + 32 * j, S, lane, vf[j]);
    }
#pragma unroll
    for (int j = 0; j < KB; ++j) { frags_arrived(kf[j]); frags_arrived(vf[j]); }

    const int nt_all = (S + TILE - 1) / TILE;
    const int tb = SPLIT ? nt_all * chunk / nsplit : 0;              // this workgroup's query tiles: [tb, nt)
    const int nt = SPLIT ? nt_all * (chunk + 1) / nsplit : nt_all;

    // the pipeline's first transposed reads hit the slot "before" tile tb (ring slot 3, both tiles): make it finite
    {
        const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(lds + 3 * W1_SLOT_BYTES + i * 4096 + threadIdx.x * 16) = z;
    }
    __syncthreads();

    const bf16_t* Qb = Q + ((size_t)b * sq.b + (size_t)h * sq.h);
    const bf16_t* dOb = dO + ((size_t)b * sdo.b + (size_t)h * sdo.h);
    const W1Rsrc qrs = w1_rsrc(Qb, ((uint32_t)(S - 1) * sq.s + (uint32_t)HD) * 2u);
    const W1Rsrc dors = w1_rsrc(dOb, ((uint32_t)(S - 1) * sdo.s + (uint32_t)HD) * 2u);
    const W1Rsrc strs = w1_rsrc(STATS + (int64_t)bh * 2 * S, (uint32_t)(2 * S) * 4u);
    uint32_t qvo[2], dvo[2];
    w1_dma_offsets<2>(wave, lane, sq.s, qvo);
    w1_dma_offsets<2>(wave, lane, sdo.s, dvo);
    const uint32_t qstep = __builtin_amdgcn_readfirstlane(64u * sq.s * 2u), dstep = __builtin_amdgcn_readfirstlane(64u * sdo.s * 2u);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 2048u);
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)W1_RING_BYTES + (uint32_t)wave * 256u);
    // statistics piece of this wave: lanes 0..15 fetch -lse2 of rows 16 wave + lane, lanes 16..31 -delta of the same rows (plane 1);
    // the upper half-wave repeats the lower one (its 128 bytes of the LDS piece are never read)
    const uint32_t srow = (uint32_t)(16 * wave + (lane & 15)), splane = (uint32_t)((lane >> 4) & 1);
    uint32_t svo = (splane * (uint32_t)S + srow + (uint32_t)tb * 64u) * 4u;
    u32x4_t voff = {qvo[0] + (uint32_t)tb * qstep, qvo[1] + (uint32_t)tb * qstep, dvo[0] + (uint32_t)tb * dstep, dvo[1] + (uint32_t)tb * dstep};
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // tiles tb, tb + 1 -> ring slots 0, 1
        const uint32_t dst = wbase + (uint32_t)i * W1_SLOT_BYTES;
        w1_dma(dst, qrs, voff[0], 0u);
        w1_dma(dst + 1024u, qrs, voff[1], 0u);
        w1_dma(dst + W1_TILE_BYTES, dors, voff[2], 0u);
        w1_dma(dst + W1_TILE_BYTES + 1024u, dors, voff[3], 0u);
        w1_dma4(sbase + (uint32_t)i * W1_STAT_BYTES, strs, svo, 0u);
        voff[0] += qstep; voff[1] += qstep; voff[2] += dstep; voff[3] += dstep;
        svo += 256u;
    }

    const W1Lane la = w1_lane_offsets(lane);
    const u32x8_t la8 = {la.row[0], la.row[1], la.row[2], la.row[3], la.tr[0][0], la.tr[0][1], la.tr[1][0], la.tr[1][1]};
    const uint32_t sread = lds0 + (uint32_t)W1_RING_BYTES + 16u * (uint32_t)hi;
    const u32x16_t kf0 = pack4(kf[0][0], kf[0][1], kf[0][2], kf[0][3]), kf1 = pack4(kf[1][0], kf[1][1], kf[1][2], kf[1][3]);
    const u32x16_t vf0 = pack4(vf[0][0], vf[0][1], vf[0][2], vf[0][3]), vf1 = pack4(vf[1][0], vf[1][1], vf[1][2], vf[1][3]);
    const uint32_t niter = (uint32_t)(nt - tb + 1);   // one extra tile step drains the pipeline
    f32x16_t dk[KB][2], dv[KB][2];
    uint32_t t0, t1;
    asm volatile(
#include "w1_dkv_loop.inc"
        : "=&s"(t0), "=&s"(t1), "={a[0:15]}"(dk[0][0]), "={a[16:31]}"(dk[0][1]), "={a[32:47]}"(dk[1][0]), "={a[48:63]}"(dk[1][1]),
          "={a[64:79]}"(dv[0][0]), "={a[80:95]}"(dv[0][1]), "={a[96:111]}"(dv[1][0]), "={a[112:127]}"(dv[1][1]), "+{v[232:235]}"(voff), "+{v236}"(svo)
        : [rq] "s"(qrs.w), [rdo] "s"(dors.w), [rst] "s"(strs.w), [qstep] "s"(qstep), [dstep] "s"(dstep), [wbase] "s"(wbase), [sbase] "s"(sbase),
          [niter] "s"(niter), "{a[128:143]}"(kf0), "{a[144:159]}"(kf1), "{a[160:175]}"(vf0), "{a[176:191]}"(vf1), "{v[224:231]}"(la8), "{v237}"(sread)
        : "memory", "scc",
#include "w1_dkv_clobbers.inc"
    );
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int db = 0; db < 2; ++db) { asm volatile("" : "+v"(dk[j][db])); asm volatile("" : "+v"(dv[j][db])); }

    if (SPLIT) {   // unscaled fp32 partials: [dK 256 x 64 | dV 256 x 64] per (task, chunk)
        float* pb = part + ((size_t)(vid - task0) * nsplit + chunk) * (2 * 256 * HD);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int r = wave * (32 * KB) + 32 * j + (lane & 31);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t wk = {dk[j][db][4 * g], dk[j][db][4 * g + 1], dk[j][db][4 * g + 2], dk[j][db][4 * g + 3]};
                    const f32x4_t wv = {dv[j][db][4 * g], dv[j][db][4 * g + 1], dv[j][db][4 * g + 2], dv[j][db][4 * g + 3]};
                    *reinterpret_cast<f32x4_t*>(pb + r * HD + db * 32 + 8 * g + 4 * hi) = wk;
                    *reinterpret_cast<f32x4_t*>(pb + 256 * HD + r * HD + db * 32 + 8 * g + 4 * hi) = wv;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const int k = k0 + 32 * j + (lane & 31);
        if (k < S) {
            bf16_t* kp = dK + ((size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)k * sdk.s);
            bf16_t* vp = dV + ((size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)k * sdv.s);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {          // 16-byte stores: common.h pair_rows8
                    u32x2_t wk[2], wv[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * gp + e;
                        wk[e][0] = pack_bf16x2(dk[j][db][4 * g] * kscale, dk[j][db][4 * g + 1] * kscale);
                        wk[e][1] = pack_bf16x2(dk[j][db][4 * g + 2] * kscale, dk[j][db][4 * g + 3] * kscale);
                        wv[e][0] = pack_bf16x2(dv[j][db][4 * g], dv[j][db][4 * g + 1]);
                        wv[e][1] = pack_bf16x2(dv[j][db][4 * g + 2], dv[j][db][4 * g + 3]);
                    }
                    *reinterpret_cast<u32x4_t*>(kp + db * 32 + 8 * (2 * gp + hi)) = pair_rows8(wk[0], wk[1]);
                    *reinterpret_cast<u32x4_t*>(vp + db * 32 + 8 * (2 * gp + hi)) = pair_rows8(wv[0], wv[1]);
                }
        }
    }
}

// sum the query-range chunks of the split dK/dV tasks: one wave per key row, lane = d
__global__ __launch_bounds__(256) void w1_dkv_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_kt, bf16_t* __restrict__ dK,
                                                             bf16_t* __restrict__ dV, TStride sdk, TStride sdv, int S, int H, float kscale) {
    const int lane = threadIdx.x & 63;
    const int r = ((int)blockIdx.x % 64) * 4 + (threadIdx.x >> 6), tl = (int)blockIdx.x / 64;
    const int vid = task0 + tl, bh = vid / n_kt, kt = vid % n_kt;
    const int key = kt * 256 + r;
    if (key >= S) return;
    const float* pb = part + (size_t)tl * nsplit * (2 * 256 * HD) + r * HD + lane;
    float ak = 0.f, av = 0.f;
    for (int c = 0; c < nsplit; ++c) {
        ak += pb[(size_t)c * (2 * 256 * HD)];
        av += pb[(size_t)c * (2 * 256 * HD) + 256 * HD];
    }
    const int b = bh / H, h = bh % H;
    dK[(size_t)b * sdk.b + (size_t)h * sdk.h + (size_t)key * sdk.s + lane] = f32_to_bf16(ak * kscale);
    dV[(size_t)b * sdv.b + (size_t)h * sdv.h + (size_t)key * sdv.s + lane] = f32_to_bf16(av);
}

// step 1 of the w1 backward: delta[b,h,q] = sum_d dO * O (as vgpa_attn_bwd_delta) and the statistics planes the dK/dV kernel
// streams: stats[b,h,0,q] = -lse2, stats[b,h,1,q] = -delta
__global__ __launch_bounds__(256) void w1_bwd_prep_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O, const float* __restrict__ LSE2,
                                                            TStride sdo, TStride so, int S, int H, int64_t total /* B*H*S */, float* __restrict__ delta,
                                                            float* __restrict__ stats, const void* __restrict__ ORES, TStride sor, int res_kind) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 8 lanes per (b,h,q) row, 16 B each
    const int64_t row = gid >> 3;
    const int c8 = (int)(gid & 7);
    float acc = 0.f;
    int64_t bh = 0;
    int q = 0;
    if (row < total) {
        q = (int)(row % S);
        bh = row / S;
        const int h = (int)(bh % H), b = (int)(bh / H);
        float a[8], o[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(dO + ((size_t)b * sdo.b + (size_t)h * sdo.h + (size_t)q * sdo.s + c8 * 8)), a);
        const u32x4_t ob = *reinterpret_cast<const u32x4_t*>(O + ((size_t)b * so.b + (size_t)h * so.h + (size_t)q * so.s + c8 * 8));
        const size_t ro = (size_t)b * sor.b + (size_t)h * sor.h + (size_t)q * sor.s + c8 * 8;
        if (res_kind == VGPA_RES_8) {          // the forward's 8 further mantissa bits (common.h res8): delta from the output to 2^-17
            unpack8_res8(ob, *reinterpret_cast<const u32x2_t*>((const uint8_t*)ORES + ro), o);
        } else {
            unpack8(ob, o);
            if (res_kind == VGPA_RES_BF16) {   // the forward's rounding residual: delta from O + O_res (see w1_residual4)
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
    if (row < total && c8 == 0) {
        delta[row] = acc;
        stats[bh * 2 * S + q] = -LSE2[row];
        stats[bh * 2 * S + S + q] = -acc;
    }
}

// sum the key-range chunks of the split dQ tasks: one wave per query row, lane = d
__global__ __launch_bounds__(256) void w1_dq_merge_kernel(const float* __restrict__ part, int nsplit, int task0, int n_qt, int rows_per_task,
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

#define W1_MAX_SPLIT 16
static inline int64_t w1_slots() { return wg_slots() / 2; }   // one 256-thread workgroup per CU

extern "C" {

// dQ on the w1 structure; arguments as vgpa_attn_bwd_dq_ws (include/videogpa_hip.h): with a workspace
// (>= vgpa_attn_bwd_split_workspace_bytes) the tasks of a mostly empty last scheduling round are cut into key-range chunks.
int32_t vgpa_attn_bwd_dq_w1(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta, void* dq,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                            const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode,
                            void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!q || !k || !v || !d_o || !lse2 || !delta || !dq || head_dim != HD || B <= 0 || H <= 0 || S <= 0 || S > (1 << 24)) return VGPA_ERR_INVALID;
#define SOK(st) (stride_ok(st) && range_ok(st, B, H, S))
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(do_strides) || !SOK(dq_strides) || !al16(q) || !al16(k) || !al16(v) ||
        !al16(d_o) || !al16(dq) || (workspace && !al16(workspace)))
        return VGPA_ERR_INVALID;
#undef SOK
    const int rows = 256;
    const int n_t = (int)((S + rows - 1) / rows);
    const int64_t tasks = (int64_t)n_t * B * H;
    if (tasks > 0x7fffffff) return VGPA_ERR_INVALID;
    int64_t n_main = tasks;
    int nsplit = 1;
    if (workspace) split_plan(tasks, (int)((S + TILE - 1) / TILE), split_mode, W1_MAX_SPLIT, &n_main, &nsplit, w1_slots());
    const int64_t n_tail = tasks - n_main;
    if (n_tail > 0 && ws_bytes < (size_t)n_tail * nsplit * rows * HD * sizeof(float)) {
        if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
        n_main = tasks;
    }
    if (n_main > 0) {
        VGPA_LAUNCH((attn_bwd_dq_w1_kernel<false>), dim3((unsigned)n_main), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dq, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dq_strides), (int)S, (int)H, n_t, scale, 0, 1, (float*)nullptr);
        VGPA_CHECK_LAUNCH();
    }
    if (n_main < tasks) {
        VGPA_LAUNCH((attn_bwd_dq_w1_kernel<true>), dim3((unsigned)(n_tail * nsplit)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dq, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dq_strides), (int)S, (int)H, n_t, scale, (int)n_main, nsplit, (float*)workspace);
        VGPA_CHECK_LAUNCH();
        VGPA_LAUNCH(w1_dq_merge_kernel, dim3((unsigned)(n_tail * (rows / 4))), dim3(256), 0, stream, (const float*)workspace, nsplit, (int)n_main, n_t,
                    rows, (bf16_t*)dq, mk(dq_strides), (int)S, (int)H, scale);
        VGPA_CHECK_LAUNCH();
    }
    return VGPA_OK;
}


// step 1 for the w1 dK/dV kernel: delta (fp32 [B,H,S]) and stats (fp32 [B,H,2,S] = {-lse2, -delta})
// o_res (optional, with its own strides, res_kind as vgpa_attn_fwd_w1_res wrote it): delta is then rowsum(dO o O) of the output as o_res completes it
int32_t vgpa_attn_bwd_prep_w1_res(const void* o, const void* o_res, int32_t res_kind, const void* d_o, const float* lse2, const int64_t* o_strides,
                                  const int64_t* ores_strides, const int64_t* do_strides, float* delta, float* stats, int64_t B, int64_t H, int64_t S,
                                  int64_t head_dim, hipStream_t stream) {
    if (!o || !d_o || !lse2 || !delta || !stats || head_dim != HD || B <= 0 || H <= 0 || S <= 0 || S > (1 << 24)) return VGPA_ERR_INVALID;
#define SOK(st) (stride_ok(st) && range_ok(st, B, H, S))
    if (!SOK(o_strides) || !SOK(do_strides) || !al16(o) || !al16(d_o)) return VGPA_ERR_INVALID;
    if (o_res && (!SOK(ores_strides) || !al16(o_res) || (res_kind != VGPA_RES_BF16 && res_kind != VGPA_RES_8))) return VGPA_ERR_INVALID;
    const int64_t total = B * H * S;
    VGPA_LAUNCH(w1_bwd_prep_kernel, dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)d_o, (const bf16_t*)o, lse2,
                mk(do_strides), mk(o_strides), (int)S, (int)H, total, delta, stats, o_res, o_res ? mk(ores_strides) : mk(o_strides),
                o_res ? (int)res_kind : VGPA_RES_NONE);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
int32_t vgpa_attn_bwd_prep_w1(const void* o, const void* d_o, const float* lse2, const int64_t* o_strides, const int64_t* do_strides, float* delta,
                              float* stats, int64_t B, int64_t H, int64_t S, int64_t head_dim, hipStream_t stream) {
    return vgpa_attn_bwd_prep_w1_res(o, nullptr, VGPA_RES_NONE, d_o, lse2, o_strides, nullptr, do_strides, delta, stats, B, H, S, head_dim, stream);
}

// dK, dV on the w1 structure: arguments as vgpa_attn_bwd_dkv_ws, with `stats` (vgpa_attn_bwd_prep_w1) in the place of lse2 / delta
int32_t vgpa_attn_bwd_dkv_w1(const void* q, const void* k, const void* v, const void* d_o, const float* stats, void* dk, void* dv,
                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                             const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale,
                             int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    (void)scale;
    if (!q || !k || !v || !d_o || !stats || !dk || !dv || head_dim != HD || B <= 0 || H <= 0 || S <= 0 || S > (1 << 24)) return VGPA_ERR_INVALID;
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(do_strides) || !SOK(dk_strides) || !SOK(dv_strides) || !al16(q) || !al16(k) ||
        !al16(v) || !al16(d_o) || !al16(dk) || !al16(dv) || (workspace && !al16(workspace)))
        return VGPA_ERR_INVALID;
#undef SOK
    const int n_t = (int)((S + 255) / 256);
    const int64_t tasks = (int64_t)n_t * B * H;
    if (tasks > 0x7fffffff) return VGPA_ERR_INVALID;
    const float kscale = 0.6931471805599453f;   // q arrives pre-scaled by scale * log2(e): dK = ln 2 * (dS^T Q)
    int64_t n_main = tasks;
    int nsplit = 1;
    if (workspace) split_plan(tasks, (int)((S + TILE - 1) / TILE), split_mode, W1_MAX_SPLIT, &n_main, &nsplit, w1_slots());
    const int64_t n_tail = tasks - n_main;
    if (n_tail > 0 && ws_bytes < (size_t)n_tail * nsplit * 2 * 256 * HD * sizeof(float)) {
        if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
        n_main = tasks;
    }
    if (n_main > 0) {
        VGPA_LAUNCH((attn_bwd_dkv_w1_kernel<false>), dim3((unsigned)n_main), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (const bf16_t*)d_o, stats, (bf16_t*)dk, (bf16_t*)dv, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides), mk(dk_strides),
                    mk(dv_strides), (int)S, (int)H, n_t, kscale, 0, 1, (float*)nullptr);
        VGPA_CHECK_LAUNCH();
    }
    if (n_main < tasks) {
        VGPA_LAUNCH((attn_bwd_dkv_w1_kernel<true>), dim3((unsigned)(n_tail * nsplit)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (const bf16_t*)d_o, stats, (bf16_t*)dk, (bf16_t*)dv, mk(q_strides), mk(k_strides), mk(v_strides), mk(do_strides),
                    mk(dk_strides), mk(dv_strides), (int)S, (int)H, n_t, kscale, (int)n_main, nsplit, (float*)workspace);
        VGPA_CHECK_LAUNCH();
        VGPA_LAUNCH(w1_dkv_merge_kernel, dim3((unsigned)(n_tail * 64)), dim3(256), 0, stream, (const float*)workspace, nsplit, (int)n_main, n_t,
                    (bf16_t*)dk, (bf16_t*)dv, mk(dk_strides), mk(dv_strides), (int)S, (int)H, kscale);
        VGPA_CHECK_LAUNCH();
    }
    return VGPA_OK;
}


// Forward on the w1 structure: arguments and results as vgpa_attn_fwd_ws; the workspace (>= vgpa_attn_fwd_w1_workspace_bytes) is REQUIRED:
// it holds max_k |k|^2 per (batch, head), one redo flag per 256-row strip and the tail-split partials.
size_t vgpa_attn_fwd_w1_workspace_bytes(int64_t B, int64_t H, int64_t S) {
    const int64_t n_qt = (S + 255) / 256, tasks = n_qt * B * H;
    int64_t parts = w1_slots();
    if (tasks * W1_MAX_SPLIT < parts) parts = tasks * W1_MAX_SPLIT;
    const size_t head = (((size_t)(B * H) + (size_t)tasks) * 4 + 255) / 256 * 256;
    return head + (size_t)parts * W1_FWD_PART_FLOATS * sizeof(float);
}
// vgpa_attn_fwd_w1 that also leaves what the bf16 rounding of the output dropped, for the backward's delta (vgpa_attn_bwd_prep_w1_res /
// vgpa_attn_bwd_delta_res): o_res = a [B,H,S,64] view with its own element strides (NULL: not written) of
//   res_kind VGPA_RES_BF16 (1): bf16, O_fp32 - bf16(O);   VGPA_RES_8 (2): uint8, eight further mantissa bits (common.h res8) -- half the bytes, O to 2^-17 either way
static int32_t fwd_w1_impl(const void* q, const void* k, const void* v, void* o, void* o_res, int32_t res_kind, float* lse2, const int64_t* q_strides,
                           const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* ores_strides, int64_t B,
                           int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes,
                           hipStream_t stream, bool force_online) {
    (void)scale;
    if (!q || !k || !v || !o || !lse2 || !workspace || head_dim != HD || B <= 0 || H <= 0 || S <= 0 || S > (1 << 24)) return VGPA_ERR_INVALID;
#define SOK(st) (stride_ok(st) && range_ok(st, B, H, S))
    if (!SOK(q_strides) || !SOK(k_strides) || !SOK(v_strides) || !SOK(o_strides) || !al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(workspace))
        return VGPA_ERR_INVALID;
    if (o_res && (!SOK(ores_strides) || !al16(o_res) || (res_kind != VGPA_RES_BF16 && res_kind != VGPA_RES_8))) return VGPA_ERR_INVALID;
#undef SOK
    void* ores = o_res;
    const int rk = o_res ? (int)res_kind : VGPA_RES_NONE;
    const TStride sor = o_res ? mk(ores_strides) : mk(o_strides);
    const int n_qt = (int)((S + 255) / 256);
    const int64_t tasks = (int64_t)n_qt * B * H;
    if (tasks > 0x7fffffff || B * H > 65535) return VGPA_ERR_INVALID;
    const size_t head = (((size_t)(B * H) + (size_t)tasks) * 4 + 255) / 256 * 256;
    if (ws_bytes < head) return VGPA_ERR_WORKSPACE;
    unsigned* kmax2 = (unsigned*)workspace;
    int* flags = (int*)workspace + B * H;
    float* part = (float*)((char*)workspace + head);
    if (S < W1_FWD_MIN_S || force_online) {   // a handful of keys per row: the online-softmax kernel (its top weight is exactly 1; see W1_FWD_MIN_S)
        if (hipMemsetAsync(workspace, 0xff, head, stream) != hipSuccess) return VGPA_ERR_LAUNCH;   // every strip flagged
        return vgpa_internal_attn_fwd_redo(q, k, v, o, lse2, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S, (int)H, n_qt, tasks,
                                           flags, stream, ores, sor, rk);
    }
    if (hipMemsetAsync(workspace, 0, head, stream) != hipSuccess) return VGPA_ERR_LAUNCH;
    VGPA_LAUNCH(w1_kmax_kernel, dim3(16, (unsigned)(B * H)), dim3(256), 0, stream, (const bf16_t*)k, mk(k_strides), (int)S, (int)H, kmax2);
    VGPA_CHECK_LAUNCH();
    int64_t n_main = tasks;
    int nsplit = 1;
    split_plan(tasks, (int)((S + TILE - 1) / TILE), split_mode, W1_MAX_SPLIT, &n_main, &nsplit, w1_slots());
    const int64_t n_tail = tasks - n_main;
    if (n_tail > 0 && ws_bytes < head + (size_t)n_tail * nsplit * W1_FWD_PART_FLOATS * sizeof(float)) {
        if (split_mode >= 2) return VGPA_ERR_WORKSPACE;
        n_main = tasks;
    }
    if (n_main > 0) {
        VGPA_LAUNCH((attn_fwd_w1_kernel<false>), dim3((unsigned)n_main), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                    (bf16_t*)o, lse2, (const unsigned*)kmax2, flags, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S, (int)H, n_qt, 0, 1,
                    (float*)nullptr, ores, sor, rk);
        VGPA_CHECK_LAUNCH();
    }
    if (n_main < tasks) {
        VGPA_LAUNCH((attn_fwd_w1_kernel<true>), dim3((unsigned)(n_tail * nsplit)), dim3(256), 0, stream, (const bf16_t*)q, (const bf16_t*)k,
                    (const bf16_t*)v, (bf16_t*)o, lse2, (const unsigned*)kmax2, flags, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S,
                    (int)H, n_qt, (int)n_main, nsplit, part, ores, sor, rk);
        VGPA_CHECK_LAUNCH();
        VGPA_LAUNCH(w1_fwd_merge_kernel, dim3((unsigned)(n_tail * 64)), dim3(256), 0, stream, (const float*)part, nsplit, (int)n_main, n_qt, (bf16_t*)o,
                    mk(o_strides), lse2, flags, (int)S, (int)H, ores, sor, rk);
        VGPA_CHECK_LAUNCH();
    }
    return vgpa_internal_attn_fwd_redo(q, k, v, o, lse2, mk(q_strides), mk(k_strides), mk(v_strides), mk(o_strides), (int)S, (int)H, n_qt, tasks, flags, stream,
                                       ores, sor, rk);
}
int32_t vgpa_attn_fwd_w1_res(const void* q, const void* k, const void* v, void* o, void* o_res, int32_t res_kind, float* lse2, const int64_t* q_strides,
                             const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* ores_strides, int64_t B,
                             int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes,
                             hipStream_t stream) {
    return fwd_w1_impl(q, k, v, o, o_res, res_kind, lse2, q_strides, k_strides, v_strides, o_strides, ores_strides, B, H, S, head_dim, scale, split_mode, workspace,
                       ws_bytes, stream, false);
}
// Same arguments, same results, same workspace: EVERY strip on the online-softmax (running-maximum) kernel.  For inputs whose row maxima lie further below
// |q| max|k| than the bound-shifted kernel represents (most strips of vgpa_attn_fwd_w1_res flagged: its redo count is the first int32 block of the workspace
// behind the B*H kmax words) this is the faster call: one sweep instead of two (videogpa_amd.transformer.AttentionCore switches on the measured count).
int32_t vgpa_attn_fwd_online_res(const void* q, const void* k, const void* v, void* o, void* o_res, int32_t res_kind, float* lse2, const int64_t* q_strides,
                                 const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* ores_strides, int64_t B,
                                 int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes,
                                 hipStream_t stream) {
    return fwd_w1_impl(q, k, v, o, o_res, res_kind, lse2, q_strides, k_strides, v_strides, o_strides, ores_strides, B, H, S, head_dim, scale, split_mode, workspace,
                       ws_bytes, stream, true);
}
int32_t vgpa_attn_fwd_w1(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides, const int64_t* k_strides,
                         const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale,
                         int32_t split_mode, void* workspace, size_t ws_bytes, hipStream_t stream) {
    return vgpa_attn_fwd_w1_res(q, k, v, o, nullptr, VGPA_RES_NONE, lse2, q_strides, k_strides, v_strides, o_strides, nullptr, B, H, S, head_dim, scale, split_mode,
                                workspace, ws_bytes, stream);
}

}  // extern "C"
