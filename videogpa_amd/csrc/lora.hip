// LoRA A.B contractions on MFMA (SURVEY K5/K9 adapter part).  Replaces peft.tuners.lora.layer.Linear's
// `lora_B(lora_A(x)) * scaling` and its autograd backward for the adapters the reference puts on
// to_q / to_k / to_v / to_out.0 (train/CogVideoX-5B/03_train.py:102-106); oracle: oracle/cogvideox.py::_lora_linear.
//
// All three kernels are HBM-bound (rank r = 64 against 3072-wide activations: 2 r FLOP per activation byte), so
// they are built to touch the big [M, 3072] operand exactly once with 16-byte row-contiguous accesses and to keep
// the small operands in L2 / LDS:
//   down   : T[M,R]  = X[M,K] A[R,K]^T                 (R = all adapters sharing the input, e.g. 3*64 for q,k,v)
//   up_add : Y[M,N] (+)= s * T[M,r] Bw[N,r]^T           (in place on a column slice of the fused projection output)
//   grad   : G[P,Q] += s * U[M,P]^T V[M,Q]   (fp32)    (dA = dT^T X, dB = s dY^T T; contraction over tokens, split
//                                                       over M across workgroups, fp32 atomics into the small result)
// Every product is taken transposed (D[n][m]) where that makes each lane's accumulator run along the contiguous
// output dimension, so results leave as 8-byte row-contiguous pieces.
#include "attn_w1.h"

// measurement knobs of variant builds (tools/build_variant.sh ... -DLORA_DOWN_NS=3); the product library is built with the defaults and reads no environment
#ifndef LORA_DOWN_DMA
#define LORA_DOWN_DMA 1
#endif
#ifndef LORA_DOWN_DMA_NCB
#define LORA_DOWN_DMA_NCB 2
#endif
#ifndef LORA_DOWN_NS
#define LORA_DOWN_NS 4
#endif
#ifndef LORA_GRAD_WGS
#define LORA_GRAD_WGS 768
#endif

#include <cstdlib>

// ---- staging helpers with bounds (zero fill) -------------------------------------------------------------------
// rows [row0, row0+64) x cols [col0, col0+64) of a row-major bf16 matrix; rows >= nrows or cols >= ncols read as 0
__device__ __forceinline__ void tile_load_zfill(const bf16_t* base, int64_t ld, int64_t row0, int64_t nrows, int col0, int ncols,
                                                u32x4_t (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + 256 * j;
        const int64_t row = row0 + (c >> 3);
        const int col = col0 + (c & 7) * 8;
        u32x4_t z = {0u, 0u, 0u, 0u};
        r[j] = (row < nrows && col < ncols) ? *reinterpret_cast<const u32x4_t*>(base + row * ld + col) : z;
    }
}

// =====================================================================================================
// down:  T = X A^T.   Workgroup = 64 rows of X, all RP = 32*NCB output columns; K streamed in chunks of 64.
// =====================================================================================================
template <int NCB>
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ A,
                                                          bf16_t* __restrict__ T, int64_t ldt, int64_t M, int K, int R) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    constexpr int RP = 32 * NCB;
    constexpr int NACC = (NCB + 1) / 2;
    constexpr int BUF = (64 + RP) * PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int rb = wave & 1, cg = wave >> 1;
    const int64_t row0 = (int64_t)blockIdx.x * 64;

    f32x16_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    u32x4_t xr[2], ar[NCB];
    auto load = [&](int k0) {
        tile_load_zfill(X, ldx, row0, M, k0, K, xr);
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int c = threadIdx.x + 256 * j;
            const int row = c >> 3;
            u32x4_t z = {0u, 0u, 0u, 0u};
            ar[j] = (row < R) ? *reinterpret_cast<const u32x4_t*>(A + (int64_t)row * K + k0 + (c & 7) * 8) : z;
        }
    };
    auto store = [&](int buf) {
        bf16_t* xl = smem + buf * BUF;
        bf16_t* al = xl + 64 * PITCH;
        tile_store(xl, xr);
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int c = threadIdx.x + 256 * j;
            *reinterpret_cast<u32x4_t*>(al + (c >> 3) * PITCH + (c & 7) * 8) = ar[j];
        }
    };
    const int nk = K / 64;
    load(0);
    store(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const bf16_t* xl = smem + (t & 1) * BUF;
        const bf16_t* al = xl + 64 * PITCH;
        if (t + 1 < nk) load((t + 1) * 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t xf = frag_row(xl, rb * 32, ks, lane);            // B operand: column = X row
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int cb = cg + 2 * a;
                if (cb < NCB) acc[a] = mfma32(frag_row(al, cb * 32, ks, lane), xf, acc[a]);   // D[n = adapter col][m = X row]
            }
        }
        if (t + 1 < nk) store((t + 1) & 1);
        __syncthreads();
    }
    const int64_t m = row0 + rb * 32 + (lane & 31);
    if (m < M) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int cb = cg + 2 * a;
            if (cb < NCB) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = cb * 32 + 8 * g + 4 * hi;
                    if (n + 3 < R || n < R) {
                        if (n + 3 < R) {
                            u32x2_t w;
                            w[0] = pack_bf16x2(acc[a][4 * g], acc[a][4 * g + 1]);
                            w[1] = pack_bf16x2(acc[a][4 * g + 2], acc[a][4 * g + 3]);
                            *reinterpret_cast<u32x2_t*>(T + m * ldt + n) = w;
                        } else {
                            for (int i = 0; i < 4; ++i)
                                if (n + i < R) T[m * ldt + n + i] = f32_to_bf16(acc[a][4 * g + i]);
                        }
                    }
                }
            }
        }
    }
}

// =====================================================================================================
// down, LDS-DMA form: the same product with the [64 x 64] X chunks and the [RP x 64] A chunks streamed by `buffer_load ... lds` into a
// ring of NS stages (chunk-swizzled image of attn_w1.h, conflict-free 16-byte fragment reads), NS - 1 chunks in flight per workgroup.
// The register-staged kernel above has ONE chunk in flight while it computes eight MFMAs: with about one workgroup per CU (M = 18 480
// rows -> 289 workgroups) every K step waits a full HBM round trip (56 us for 113 MB = 2.0 TB/s); with the ring the stream is
// bandwidth-bound.  Chunks past K are requested too (they land in slots nobody reads): the hand-counted s_waitcnt stays uniform.
// =====================================================================================================
template <int N>
__device__ __forceinline__ void lora_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int NCB, int NS>
__global__ __launch_bounds__(256) void lora_down_dma_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ A, bf16_t* __restrict__ T,
                                                              int64_t ldt, int64_t M, int K, int R) {
    constexpr int RP = 32 * NCB;
    constexpr int NACC = (NCB + 1) / 2;
    constexpr int STAGE = 8192 + RP * 128;           // X chunk | A chunk
    constexpr int PW = 2 + NCB;                      // LDS-DMA pieces per wave and stage
    __shared__ __attribute__((aligned(1024))) uint8_t lds[NS * STAGE];
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rb = wave & 1, cg = wave >> 1;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t rows = (M - row0) < 64 ? (M - row0) : 64;

    const W1Rsrc xrs = w1_rsrc(X + row0 * ldx, (uint32_t)(((rows - 1) * ldx + K) * 2));
    const W1Rsrc ars = w1_rsrc(A, (uint32_t)((int64_t)R * K * 2));
    // piece p of a chunk = rows 8p .. 8p+7; lane -> (row 8p + lane / 8, LDS chunk position lane % 8)
    uint32_t xo[2], ao[NCB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t row = 8u * (uint32_t)(wave * 2 + i) + (uint32_t)(lane >> 3);
        xo[i] = (uint32_t)((row * (uint64_t)ldx + (((uint32_t)(lane & 7) ^ w1_swz(row)) * 8u)) * 2u);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
        const uint32_t row = 8u * (uint32_t)(wave * NCB + i) + (uint32_t)(lane >> 3);
        ao[i] = (row * (uint32_t)K + (((uint32_t)(lane & 7) ^ w1_swz(row)) * 8u)) * 2u;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t xdst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 2048u);
    const uint32_t adst = __builtin_amdgcn_readfirstlane(lds0 + 8192u + (uint32_t)wave * (uint32_t)NCB * 1024u);
    auto issue = [&](int slot) {
        const uint32_t so = (uint32_t)slot * (uint32_t)STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) { w1_dma(xdst + so + 1024u * i, xrs, xo[i], 0u); xo[i] += 128u; }
#pragma unroll
        for (int i = 0; i < NCB; ++i) { w1_dma(adst + so + 1024u * i, ars, ao[i], 0u); ao[i] += 128u; }
    };
    f32x16_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const W1Lane la = w1_lane_offsets(lane);
    const int nk = K / 64;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);
    for (int t = 0; t < nk; ++t) {
        lora_wait_vm<PW * (NS - 2)>();          // chunk t has landed (this wave's pieces); chunks t+1 .. t+NS-2 may still fly
        __syncthreads();                        // ... for every wave, and everybody is done reading chunk t-1
        issue((t + NS - 1) % NS);               // refill chunk t-1's slot
        const uint32_t xb = (uint32_t)(t % NS) * (uint32_t)STAGE, ab = xb + 8192u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t xf = w1_frag_row(lds, xb, la, rb, ks);
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int cb = cg + 2 * a;
                if (cb < NCB) acc[a] = mfma32(w1_frag_row(lds, ab, la, cb, ks), xf, acc[a]);
            }
        }
    }
    lora_wait_vm<0>();                          // nothing may still be writing this workgroup's LDS when it retires
    const int64_t m = row0 + rb * 32 + (lane & 31);
    if (m < M) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int cb = cg + 2 * a;
            if (cb < NCB) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = cb * 32 + 8 * g + 4 * hi;
                    if (n + 3 < R) {
                        u32x2_t w;
                        w[0] = pack_bf16x2(acc[a][4 * g], acc[a][4 * g + 1]);
                        w[1] = pack_bf16x2(acc[a][4 * g + 2], acc[a][4 * g + 3]);
                        *reinterpret_cast<u32x2_t*>(T + m * ldt + n) = w;
                    } else {
                        for (int i = 0; i < 4; ++i)
                            if (n + i < R) T[m * ldt + n + i] = f32_to_bf16(acc[a][4 * g + i]);
                    }
                }
            }
        }
    }
}

// =====================================================================================================
// up_add:  Y (+)= s * T Bw^T.   T / Bw fragments come straight from L2 (both are tiny).  A workgroup owns a
// 64-row x 128-column block of Y: the four waves put their s*T*Bw^T sub-blocks into an fp32 LDS tile, then all 256
// threads read-modify-write Y in 16-byte row-contiguous pieces (the 436 MB RMW of Y is the whole cost of this op).
// =====================================================================================================
#ifndef UP_COLS
#define UP_COLS 128
#endif
#define UP_PITCH (UP_COLS + 4)   // fp32 words per LDS row (+4: conflict-free 16-byte writes down a column of rows)
template <int KS>
__global__ __launch_bounds__(256) void lora_up_add_kernel(bf16_t* __restrict__ Y, int64_t ldy, const bf16_t* __restrict__ T, int64_t ldt,
                                                            const bf16_t* __restrict__ Bw, int64_t ldb, float s, int64_t M, int N,
                                                            int accumulate) {
    __shared__ __attribute__((aligned(16))) float tile[64 * UP_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * UP_COLS;
    const int rb = wave & 1, cg = wave >> 1;                       // wave: 32 rows x 64 cols of the block
    int64_t mr = m0 + rb * 32 + (lane & 31);
    mr = mr < M ? mr : M - 1;
    bf16x8_t tf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) tf[ks] = *reinterpret_cast<const bf16x8_t*>(T + mr * ldt + ks * 16 + hi * 8);   // B operand: col = row m
#pragma unroll
    for (int cbi = 0; cbi < UP_COLS / 64; ++cbi) {
        const int nl = cg * (UP_COLS / 2) + cbi * 32;               // local column of this 32x32 block
        int nrow = n0 + nl + (lane & 31);
        nrow = nrow < N ? nrow : N - 1;
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const bf16_t* bp = Bw + (int64_t)nrow * ldb + hi * 8;                                                         // A operand: row = out col n
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = mfma32(*reinterpret_cast<const bf16x8_t*>(bp + ks * 16), tf[ks], acc);  // D[n][m]
        float* tp = tile + (rb * 32 + (lane & 31)) * UP_PITCH + nl + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4_t v = {s * acc[4 * g], s * acc[4 * g + 1], s * acc[4 * g + 2], s * acc[4 * g + 3]};
            *reinterpret_cast<f32x4_t*>(tp + 8 * g) = v;
        }
    }
    __syncthreads();
    // 64 rows x UP_COLS/8 chunks of 8 columns, UP_COLS/32 per thread; UP_COLS/8 consecutive lanes cover one row's contiguous bytes
#pragma unroll
    for (int j = 0; j < UP_COLS / 32; ++j) {
        const int c = threadIdx.x + 256 * j;
        const int row = c / (UP_COLS / 8), col = (c % (UP_COLS / 8)) * 8;
        const int64_t m = m0 + row;
        if (m < M && n0 + col < N) {
            const float* tp = tile + row * UP_PITCH + col;
            const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(tp), a1 = *reinterpret_cast<const f32x4_t*>(tp + 4);
            float o[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            bf16_t* yp = Y + m * ldy + n0 + col;
            if (accumulate) {
                float y[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(yp), y);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += y[i];
            }
            *reinterpret_cast<u32x4_t*>(yp) = pack8(o);
        }
    }
}

// =====================================================================================================
// grad:  G[P,Q] += s * U^T V  over the token rows [blockIdx.y * rows_per_split, ...).  64x64 output tile per workgroup,
// both operands are contracted over their ROW index -> hardware transpose reads of two row-major LDS tiles.
// =====================================================================================================
__global__ __launch_bounds__(256) void lora_grad_kernel(const bf16_t* __restrict__ U, int64_t ldu, const bf16_t* __restrict__ V, int64_t ldv,
                                                          float* __restrict__ G, int64_t ldg, float s, int64_t M, int P, int Q,
                                                          int64_t rows_per_split, int n_qt, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[4 * TILE_ELEMS];   // U[2], V[2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int pb = wave & 1, qb = wave >> 1;
    const int p0 = (blockIdx.x / n_qt) * 64, q0 = (blockIdx.x % n_qt) * 64;
    const int64_t mbeg = (int64_t)blockIdx.y * rows_per_split;
    int64_t mend = mbeg + rows_per_split;
    mend = mend < M ? mend : M;
    if (mbeg >= mend) return;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    u32x4_t ur[2], vr[2];
    const int nt = (int)((mend - mbeg + 63) / 64);
    tile_load_zfill(U, ldu, mbeg, mend, p0, P, ur);
    tile_load_zfill(V, ldv, mbeg, mend, q0, Q, vr);
    tile_store(lds, ur);
    tile_store(lds + 2 * TILE_ELEMS, vr);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const bf16_t* ul = lds + (t & 1) * TILE_ELEMS;
        const bf16_t* vl = lds + (2 + (t & 1)) * TILE_ELEMS;
        if (t + 1 < nt) {
            tile_load_zfill(U, ldu, mbeg + (int64_t)(t + 1) * 64, mend, p0, P, ur);
            tile_load_zfill(V, ldv, mbeg + (int64_t)(t + 1) * 64, mend, q0, Q, vr);
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) acc = mfma32(frag_tr(ul, 16 * kc, pb * 32, lane), frag_tr(vl, 16 * kc, qb * 32, lane), acc);
        if (t + 1 < nt) {
            tile_store(lds + ((t + 1) & 1) * TILE_ELEMS, ur);
            tile_store(lds + (2 + ((t + 1) & 1)) * TILE_ELEMS, vr);
        }
        __syncthreads();
    }
    const int q = q0 + qb * 32 + (lane & 31);
    if (q < Q) {
        // deterministic form: this row range's partial goes to its own [P, Q] slab, lora_grad_merge_kernel adds the slabs in order
        float* slab = part ? part + (int64_t)blockIdx.y * P * Q : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + pb * 32 + acc_row(r, hi);
            if (p < P) {
                if (slab) slab[(int64_t)p * Q + q] = acc[r];
                else atomicAdd(&G[(int64_t)p * ldg + q], s * acc[r]);
            }
        }
    }
}

// G[p, q] = s * sum over the row-range slabs, in slab order (fixed summation order: bit-reproducible)
__global__ __launch_bounds__(256) void lora_grad_merge_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ G, int64_t ldg, float s,
                                                                int P, int Q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)P * Q) return;
    float a = 0.f;
    for (int c = 0; c < nsplit; ++c) a += part[(int64_t)c * P * Q + i];
    G[(i / Q) * ldg + (i % Q)] = s * a;
}

// =====================================================================================================
// adapter refresh of the K-extended projection operands (ops.LoraExt): after an optimizer step the fp32 adapters A [r, K], B [Dn, r] of ONE
// adapter are written, rounded exactly as PEFT's forward rounds them (a = bf16(A); sB = bf16(float(bf16(B)) * s)), to the four places the
// extended GEMMs read:  A_cat rows, the [K, N + R] operand's tail columns (A^T), the [N, K + R] operand's tail columns (sB), sBt [rp, Dn].
// One launch instead of eight strided torch copies per adapter (168 / 240 adapters per step at cfg2 / cfg5).
// =====================================================================================================
__global__ __launch_bounds__(256) void lora_ext_refresh_kernel(const float* __restrict__ A, const float* __restrict__ B, float s, int r, int K, int Dn,
                                                                 bf16_t* __restrict__ a_cat, bf16_t* __restrict__ wt_tail, int64_t ld_wt,
                                                                 bf16_t* __restrict__ w_tail, int64_t ld_w, bf16_t* __restrict__ sbt) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t na = (int64_t)r * K;
    if (idx < na) {
        const int rr = (int)(idx / K), kk = (int)(idx % K);
        const bf16_t a = f32_to_bf16(A[idx]);
        a_cat[idx] = a;
        wt_tail[(int64_t)kk * ld_wt + rr] = a;
    } else if (idx < na + (int64_t)Dn * r) {
        const int64_t e = idx - na;
        const int n = (int)(e / r), rr = (int)(e % r);
        const bf16_t v = f32_to_bf16(bf16_to_f32(f32_to_bf16(B[e])) * s);
        w_tail[(int64_t)n * ld_w + rr] = v;
        sbt[(int64_t)rr * Dn + n] = v;
    }
}

static inline bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" {

// T[M,R] = X[M,K] A[R,K]^T  (bf16; X row stride ldx, T row stride ldt; K % 64 == 0; R <= 256; ldx, ldt, K multiples of 8)
int32_t vgpa_lora_down(const void* X, int64_t ldx, const void* A, void* T, int64_t ldt, int64_t M, int64_t K, int64_t R, hipStream_t stream) {
    if (!X || !A || !T || M <= 0 || K <= 0 || K % 64 != 0 || R <= 0 || R > 256 || ldx % 8 || ldt % 8 || ldt < R || !a16(X) || !a16(A)) return VGPA_ERR_INVALID;
    if (M > ((int64_t)1 << 31) * 32) return VGPA_ERR_INVALID;
    const int ncb = (int)((R + 31) / 32);
    const dim3 grid((unsigned)((M + 63) / 64));
    // LDS-DMA form (descriptor offsets are 32 bit: one workgroup's 64 rows of X and the whole of A must be addressable, incl. the
    // chunks requested past K); -DLORA_DOWN_DMA=0 (a variant build: tools/build_variant.sh) selects the register-staged kernel.  The library reads no
    // environment: every measurement knob below is a compile-time constant of a variant build.
    constexpr bool dma_off = LORA_DOWN_DMA == 0;
    // measured (tools/lora_bench.py, round 5: stand-alone, every call on the next of six operands so that the 256 MB infinity cache cannot serve the
    // 218 MB read -- a loop over ONE operand reports 5.0-5.4 TB/s, the step sees what follows): R = 64 with the 4-stage ring 59.7 / 57.6 us at M = 35 552 /
    // 36 960 = 3.66 / 3.95 TB/s, with 3 stages (three workgroups per CU resident) 60.6 / 58.3 us; two 64-row blocks per A chunk (half the L2 -> LDS stream
    // of A) 68.7 us -- fewer, larger workgroups lose more than the halved A traffic gains.  R = 192 (the shared q/k/v down-projection, ONE launch per layer
    // and policy pass) is 116-128 us on either kernel, with 2-4 stages and with one or two row blocks alike (-DLORA_DOWN_DMA_NCB / -DLORA_DOWN_NS
    // variant builds): the ring stays with the narrow adapters
    constexpr int dma_max_ncb = LORA_DOWN_DMA_NCB;
    if (!dma_off && ncb <= dma_max_ncb && (uint64_t)64 * (uint64_t)ldx * 2 + (uint64_t)K * 2 + 1024 < (1ull << 32) && ((uint64_t)R + 8) * (uint64_t)K * 2 + 1024 < (1ull << 32)) {
#define LDD(N, S) VGPA_LAUNCH((lora_down_dma_kernel<N, S>), grid, dim3(256), 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)A, (bf16_t*)T, ldt, M, (int)K, (int)R)
        constexpr int ns = LORA_DOWN_NS;
        switch (ncb * 10 + ns) {
            case 14: LDD(1, 4); break; case 24: LDD(2, 4); break;
            case 13: LDD(1, 3); break; case 23: LDD(2, 3); break;
            case 63: LDD(6, 3); break; case 64: LDD(6, 4); break;
            default: return VGPA_ERR_INVALID;
        }
#undef LDD
        VGPA_CHECK_LAUNCH();
        return VGPA_OK;
    }
    const size_t shmem = (size_t)2 * (64 + 32 * ncb) * PITCH * sizeof(bf16_t);
#define LD(N)                                                                                                                      \
    if (hipFuncSetAttribute((const void*)lora_down_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) \
        return VGPA_ERR_LAUNCH;                                                                                                    \
    VGPA_LAUNCH((lora_down_kernel<N>), grid, dim3(256), shmem, stream, (const bf16_t*)X, ldx, (const bf16_t*)A, (bf16_t*)T, ldt, M, (int)K, (int)R)
    switch (ncb) {
        case 1: LD(1); break; case 2: LD(2); break; case 3: LD(3); break; case 4: LD(4); break;
        case 5: LD(5); break; case 6: LD(6); break; case 7: LD(7); break; case 8: LD(8); break;
        default: return VGPA_ERR_INVALID;
    }
#undef LD
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// Y[M,N] = (accumulate ? Y : 0) + s * T[M,rp] Bw[N,rp]^T   (bf16; rp in {16,32,48,64,96,128,192}; N % 8 == 0; 16-byte aligned rows of Y)
int32_t vgpa_lora_up_add(void* Y, int64_t ldy, const void* T, int64_t ldt, const void* Bw, int64_t ldb, float s, int64_t M, int64_t N,
                         int64_t rp, int32_t accumulate, hipStream_t stream) {
    if (!Y || !T || !Bw || M <= 0 || N <= 0 || N % 8 != 0 || ldy % 8 || ldt % 8 || ldb % 8 || !a16(T) || !a16(Bw) || !a16(Y))
        return VGPA_ERR_INVALID;
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + UP_COLS - 1) / UP_COLS));
#define UP(KS) VGPA_LAUNCH((lora_up_add_kernel<KS>), grid, dim3(256), 0, stream, (bf16_t*)Y, ldy, (const bf16_t*)T, ldt, (const bf16_t*)Bw, ldb, s, M, (int)N, (int)accumulate)
    switch (rp) {
        case 16: UP(1); break; case 32: UP(2); break; case 48: UP(3); break; case 64: UP(4); break;
        case 96: UP(6); break; case 128: UP(8); break; case 192: UP(12); break;
        default: return VGPA_ERR_INVALID;
    }
#undef UP
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

static void lora_grad_plan(int64_t M, int64_t P, int64_t Q, int* n_pt, int* n_qt, int64_t* rows, int64_t* splits) {
    *n_pt = (int)((P + 63) / 64);
    *n_qt = (int)((Q + 63) / 64);
    // workgroups aimed at: 768 = three per CU (37 KiB of LDS each).  Measured at M = 35 552 on cache-cold operands (tools/lora_bench.py): 512 -> 62.4 us,
    // 768 -> 57.0, 1024 -> 62.0 (3.84 TB/s at 768)
    constexpr int64_t target = LORA_GRAD_WGS;
    int64_t sp = target / ((int64_t)*n_pt * *n_qt);
    if (sp < 1) sp = 1;
    const int64_t max_splits = (M + 255) / 256;
    if (sp > max_splits) sp = max_splits;
    int64_t r = (M + sp - 1) / sp;
    r = (r + 63) / 64 * 64;
    *rows = r;
    *splits = (M + r - 1) / r;
}

// G[P,Q] (fp32, row stride ldg, caller-zeroed) += s * U[M,P]^T V[M,Q]   (bf16 operands, P, Q multiples of 8); fp32 atomics: the
// result is reproducible only to rounding.  vgpa_lora_grad_ws is the deterministic form.
int32_t vgpa_lora_grad(const void* U, int64_t ldu, const void* V, int64_t ldv, float* G, int64_t ldg, float s, int64_t M, int64_t P, int64_t Q,
                       hipStream_t stream) {
    if (!U || !V || !G || M <= 0 || P <= 0 || Q <= 0 || P % 8 || Q % 8 || ldu % 8 || ldv % 8 || !a16(U) || !a16(V)) return VGPA_ERR_INVALID;
    int n_pt, n_qt;
    int64_t rows, splits;
    lora_grad_plan(M, P, Q, &n_pt, &n_qt, &rows, &splits);
    VGPA_LAUNCH(lora_grad_kernel, dim3((unsigned)(n_pt * n_qt), (unsigned)splits), dim3(256), 0, stream, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv,
                G, ldg, s, M, (int)P, (int)Q, rows, n_qt, (float*)nullptr);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// The same product, bit-reproducible: every row range writes its partial [P, Q] into the caller's workspace
// (>= vgpa_lora_grad_workspace_bytes) and a merge kernel adds the partials in a fixed order.  G is overwritten (no zeroing needed).
size_t vgpa_lora_grad_workspace_bytes(int64_t M, int64_t P, int64_t Q) {
    if (M <= 0 || P <= 0 || Q <= 0) return 0;
    int n_pt, n_qt;
    int64_t rows, splits;
    lora_grad_plan(M, P, Q, &n_pt, &n_qt, &rows, &splits);
    return (size_t)splits * (size_t)P * (size_t)Q * sizeof(float);
}
int32_t vgpa_lora_grad_ws(const void* U, int64_t ldu, const void* V, int64_t ldv, float* G, int64_t ldg, float s, int64_t M, int64_t P,
                          int64_t Q, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!U || !V || !G || !workspace || M <= 0 || P <= 0 || Q <= 0 || P % 8 || Q % 8 || ldu % 8 || ldv % 8 || !a16(U) || !a16(V) || !a16(workspace))
        return VGPA_ERR_INVALID;
    int n_pt, n_qt;
    int64_t rows, splits;
    lora_grad_plan(M, P, Q, &n_pt, &n_qt, &rows, &splits);
    if (ws_bytes < (size_t)splits * (size_t)P * (size_t)Q * sizeof(float)) return VGPA_ERR_WORKSPACE;
    VGPA_LAUNCH(lora_grad_kernel, dim3((unsigned)(n_pt * n_qt), (unsigned)splits), dim3(256), 0, stream, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv,
                G, ldg, s, M, (int)P, (int)Q, rows, n_qt, (float*)workspace);
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(lora_grad_merge_kernel, dim3((unsigned)((P * Q + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, (int)splits, G, ldg, s,
                (int)P, (int)Q);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// a_cat [r, K] rows of this adapter (row stride K); wt_tail = &Wt_ext[0][N + j rp] (row stride ld_wt); w_tail = &W_ext[i Dn][K + j rp] (row stride ld_w);
// sbt [>= r, Dn] (row stride Dn).  All bf16, A / B fp32 contiguous.
int32_t vgpa_lora_ext_refresh(const float* A, const float* B, float s, int64_t r, int64_t K, int64_t Dn, void* a_cat, void* wt_tail, int64_t ld_wt, void* w_tail,
                              int64_t ld_w, void* sbt, hipStream_t stream) {
    if (!A || !B || !a_cat || !wt_tail || !w_tail || !sbt || r <= 0 || K <= 0 || Dn <= 0 || ld_wt < r || ld_w < r) return VGPA_ERR_INVALID;
    const int64_t total = r * K + Dn * r;
    if (total >= ((int64_t)1 << 31)) return VGPA_ERR_INVALID;
    VGPA_LAUNCH(lora_ext_refresh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A, B, s, (int)r, (int)K, (int)Dn, (bf16_t*)a_cat, (bf16_t*)wt_tail,
                ld_wt, (bf16_t*)w_tail, ld_w, (bf16_t*)sbt);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
