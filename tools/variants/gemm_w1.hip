// bf16 GEMM with fused epilogues for the feed-forward pair of a CogVideoX block (reference call sites: diffusers
// attention.py FeedForward / GELU(approximate="tanh"), reached from cogvideox_transformer_3d.py CogVideoXBlock.forward):
//     C[M, N] = epilogue(X[M, K] W[N, K]^T + bias[N])
//       epilogue 0: identity          1: gelu_tanh (and, when `aux` is given, the pre-activation is ALSO stored there)
//                2: C = acc * gelu_tanh'(aux[M, N])      (the backward of 1: X = dY_ff2, W = W_ff2^T cached)
// Built on the w1 structure of attention_w1.hip: one wave per SIMD, all 128 accumulator tuples of a 128 x 64 wave tile in the
// accumulator half of the register file, operands streamed by LDS-DMA into a 3-stage ring of chunk-swizzled panels, the K loop
// from tools/gen_w1_asm.py::GemmLoop (w1_gemm_loop.inc: read its docstring for the LDS and register maps).
// Shapes: N % 128 == 0, K % 64 == 0, rows 16-byte aligned; M is free (rows past M read as zeros and are not stored).
// Measured slower than hipBLASLt + the separate GELU pass (profiles/r03_gemm_probe.txt), so it is compiled in variant builds only
// (tools/build_variant.sh, -DVGPA_VARIANTS) and the product feed-forward stays on hipBLASLt.
#ifdef VGPA_VARIANTS
#include "attn_common.h"

#include "attn_w1.h"

#define GW_BM 256
#define GW_BN 128
#define GW_BK 64
#define GW_STAGE_BYTES 49152
#define GW_LDS_BYTES (3 * GW_STAGE_BYTES)

#define GELU_K1 (-2.302208198f)      // -2 * 0.7978845608028654 * log2(e)      (same constants as norm.hip)
#define GELU_K2 (-0.1029432396f)     // 0.044715 * K1
__device__ __forceinline__ float gw_gelu_sig(float x, float x2) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (GELU_K1 + GELU_K2 * x2))); }

template <int EPI, bool PRE>
__global__ __launch_bounds__(256, 1) void gemm_w1_kernel(const bf16_t* __restrict__ X, int64_t ldx, const bf16_t* __restrict__ W, int64_t ldw,
                                                           const bf16_t* __restrict__ bias, bf16_t* __restrict__ C, int64_t ldc,
                                                           bf16_t* __restrict__ aux, int64_t ldaux, int M, int N, int K, int tiles_n, int group_m) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[GW_LDS_BYTES];
    if (K < 0) lds[threadIdx.x] = 0;   // never taken: keeps the allocation (at LDS address 0) that the loop addresses by number
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // tile order: consecutive ids of one XCD walk a (group_m x tiles_n) band column by column, so the workgroups resident together
    // share few X panels and few W panels in that XCD's L2
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int band = id / (group_m * tiles_n), in_band = id % (group_m * tiles_n);
    const int tiles_m = (M + GW_BM - 1) / GW_BM;
    const int band_rows = min(group_m, tiles_m - band * group_m);
    const int tm = band * group_m + in_band % band_rows, tn = in_band / band_rows;
    const int m0 = tm * GW_BM, n0 = tn * GW_BN;
    const int rows = min(GW_BM, M - m0);

    const W1Rsrc xrs = w1_rsrc(X + (size_t)m0 * ldx, (uint32_t)(((size_t)(rows - 1) * ldx + K) * 2));
    const W1Rsrc wrs = w1_rsrc(W + (size_t)n0 * ldw, (uint32_t)(((size_t)(GW_BN - 1) * ldw + K) * 2));
    u32x4_t vo[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // X sub-tile `wave`, piece j: rows 8j .. 8j+7
        const uint32_t rin = 8u * j + (uint32_t)(lane >> 3);
        vo[j >> 2][j & 3] = (uint32_t)(((size_t)(64u * wave + rin) * ldx + (((uint32_t)(lane & 7) ^ w1_swz(rin)) * 8u)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // W sub-tile wave / 2, piece 4 (wave & 1) + j
        const uint32_t rin = 8u * (4u * (wave & 1) + j) + (uint32_t)(lane >> 3);
        vo[2][j] = (uint32_t)(((size_t)(64u * (wave >> 1) + rin) * ldw + (((uint32_t)(lane & 7) ^ w1_swz(rin)) * 8u)) * 2);
    }
    const uint32_t wba = __builtin_amdgcn_readfirstlane(wave * 8192), wbw = __builtin_amdgcn_readfirstlane((wave >> 1) * 8192 + (wave & 1) * 4096);
    // stages 0 and 1
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w1_dma(st * GW_STAGE_BYTES + wba + j * 1024, xrs, vo[j >> 2][j & 3], 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) w1_dma(st * GW_STAGE_BYTES + 32768 + wbw + j * 1024, wrs, vo[2][j], 0);
#pragma unroll
        for (int j = 0; j < 12; ++j) vo[j >> 2][j & 3] += 2 * GW_BK;
    }
    const W1Lane a = w1_lane_offsets(lane);
    u32x4_t la[6];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            la[2 * st][ks] = st * GW_STAGE_BYTES + wm * 16384 + a.row[ks];
            la[2 * st + 1][ks] = st * GW_STAGE_BYTES + 32768 + wn * 8192 + a.row[ks];
        }
    const uint32_t niter = (uint32_t)(K / GW_BK);
    f32x16_t acc[2][4];
    uint32_t t0, t1;
    asm volatile(
#include "w1_gemm_loop.inc"
        : "=&s"(t0), "=&s"(t1), "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]),
          "={a[64:79]}"(acc[1][0]), "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]),
          "+{v[0:3]}"(vo[0]), "+{v[4:7]}"(vo[1]), "+{v[8:11]}"(vo[2])
        : [ra] "s"(xrs.w), [rw] "s"(wrs.w), [wba] "s"(wba), [wbw] "s"(wbw), [niter] "s"(niter), "{v[12:15]}"(la[0]), "{v[16:19]}"(la[1]),
          "{v[20:23]}"(la[2]), "{v[24:27]}"(la[3]), "{v[28:31]}"(la[4]), "{v[32:35]}"(la[5])
        : "memory", "scc",
#include "w1_gemm_clobbers.inc"
    );
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) asm volatile("" : "+v"(acc[ni][mi]));

    // epilogue: lane (m = lane & 31, hi) holds, per (ni, mi), rows n = 8 g + 4 hi + 0..3 (g = 0..3) of column m
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 128 + mi * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hi;
                float v[4] = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                if (bias) {
                    const u32x2_t bw = *reinterpret_cast<const u32x2_t*>(bias + n);
                    v[0] += __uint_as_float(bw[0] << 16); v[1] += __uint_as_float(bw[0] & 0xffff0000u);
                    v[2] += __uint_as_float(bw[1] << 16); v[3] += __uint_as_float(bw[1] & 0xffff0000u);
                }
                if (EPI == 1) {
                    if (PRE) {
                        u32x2_t p;
                        p[0] = pack_bf16x2(v[0], v[1]); p[1] = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<u32x2_t*>(aux + (size_t)m * ldaux + n) = p;
                        // the activation is taken of the ROUNDED pre-activation, as the two-kernel path (GEMM, then GELU of its bf16 output) does
                        v[0] = __uint_as_float(p[0] << 16); v[1] = __uint_as_float(p[0] & 0xffff0000u);
                        v[2] = __uint_as_float(p[1] << 16); v[3] = __uint_as_float(p[1] & 0xffff0000u);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] * gw_gelu_sig(v[i], v[i] * v[i]);
                } else if (EPI == 2) {
                    const u32x2_t uw = *reinterpret_cast<const u32x2_t*>(aux + (size_t)m * ldaux + n);
                    const float u[4] = {__uint_as_float(uw[0] << 16), __uint_as_float(uw[0] & 0xffff0000u), __uint_as_float(uw[1] << 16), __uint_as_float(uw[1] & 0xffff0000u)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // d/dx [x sig(2z)] = sig + x sig (1 - sig) (2z)'      (same arithmetic as norm.hip gelu_tanh_bwd_kernel)
                        const float x = u[i], x2 = x * x, sg = gw_gelu_sig(x, x2);
                        const float dz2 = 1.5957691216057308f + 0.2140644488f * x2;
                        v[i] *= sg + x * (sg - sg * sg) * dz2;
                    }
                }
                u32x2_t o;
                o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2_t*>(C + (size_t)m * ldc + n) = o;
            }
    }
}

extern "C" int32_t vgpa_gemm_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc, void* aux,
                                  int64_t ldaux, int32_t M, int32_t N, int32_t K, int32_t epilogue, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % GW_BN || K % GW_BK) return VGPA_ERR_INVALID;
    if (ldx % 8 || ldw % 8 || ldc % 4 || (aux && ldaux % 4)) return VGPA_ERR_INVALID;
    if (epilogue < 0 || epilogue > 2 || (epilogue == 2 && !aux)) return VGPA_ERR_INVALID;
    // the LDS-DMA descriptors address at most 4 GiB - 1 of one panel: 256 rows
    if ((uint64_t)255 * ldx * 2 + (uint64_t)K * 2 >= (1ull << 32) || (uint64_t)127 * ldw * 2 + (uint64_t)K * 2 >= (1ull << 32)) return VGPA_ERR_INVALID;
    const int tiles_m = (M + GW_BM - 1) / GW_BM, tiles_n = N / GW_BN;
    const int group_m = 8;
    const dim3 grid(tiles_m * tiles_n), block(256);
#define GW_LAUNCH(E, P) VGPA_LAUNCH((gemm_w1_kernel<E, P>), grid, block, 0, stream, (const bf16_t*)X, ldx, (const bf16_t*)W, ldw, (const bf16_t*)bias, \
                                    (bf16_t*)C, ldc, (bf16_t*)aux, ldaux, M, N, K, tiles_n, group_m)
    if (epilogue == 0) GW_LAUNCH(0, false);
    else if (epilogue == 1 && aux) GW_LAUNCH(1, true);
    else if (epilogue == 1) GW_LAUNCH(1, false);
    else GW_LAUNCH(2, false);
#undef GW_LAUNCH
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}
#endif   // VGPA_VARIANTS
