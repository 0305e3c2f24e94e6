// Micro-benchmark: how much plain-VALU / transcendental work hides beside v_mfma_f32_32x32x16_bf16 on one gfx950 SIMD,
// (a) inside one wave's instruction stream, (b) between two waves that share a SIMD.  Sizes the attention softmax
// (DESIGN.md 4.1).   hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_ITERS 4096

template <int E4, int C4, int A4, bool do_mfma, bool do_fill, int NACC>
__device__ __forceinline__ void body(f32x16 (&acc)[4], float (&e)[8], float (&a)[8], unsigned (&c)[4], const bf16x8& fa, const bf16x8& fb) {
    const float one = 1.0f;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (do_mfma) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q % NACC]) : "v"(fa), "v"(fb));
            if (do_fill) {
#pragma unroll
                for (int i = q; i < E4; i += 4) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i & 7]));
#pragma unroll
                for (int i = q; i < C4; i += 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c[i & 3]) : "v"(e[i & 7]), "v"(one));
#pragma unroll
                for (int i = q; i < A4; i += 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(one));
            }
        }
    }
}

// One "group" = 4 MFMAs (independent accumulators) with E4 v_exp, C4 v_cvt_pk, A4 v_add spread evenly behind them.
// ROLE 0: everything in every wave.  ROLE 1: waves 0-3 issue only the MFMAs, waves 4-7 only the fillers.
template <int E4, int C4, int A4, int ROLE, bool MFMA, int NACC = 4>
__global__ __launch_bounds__(512) void mix(float* out) {
    f32x16 acc[4];
    float e[8], a[8];
    unsigned c[4];
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = threadIdx.x * 1e-4f; a[i] = i; fa[i] = (__bf16)0.001f; fb[i] = (__bf16)0.002f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = 0;
    if (ROLE == 0) {
        if (MFMA) body<E4, C4, A4, true, true, NACC>(acc, e, a, c, fa, fb); else body<E4, C4, A4, false, true, NACC>(acc, e, a, c, fa, fb);
    } else {
        if (threadIdx.x < 256) body<E4, C4, A4, true, false, NACC>(acc, e, a, c, fa, fb); else body<E4, C4, A4, false, true, NACC>(acc, e, a, c, fa, fb);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i] + a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += __uint_as_float(c[i]);
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int E4, int C4, int A4, int ROLE, bool MFMA, int NACC = 4>
void run(const char* what, int threads, float ghz) {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mix<E4, C4, A4, ROLE, MFMA, NACC><<<256, threads>>>(d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<E4, C4, A4, ROLE, MFMA, NACC><<<256, threads>>>(d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_group = ms * 1e6 / N_ITERS;
    printf("%-44s E4=%2d C4=%2d A4=%2d  %7.3f ms  %7.1f ns/group = %6.1f cyc @%.2f GHz  (%.1f cyc per MFMA-slot)\n", what, E4, C4, A4, ms, ns_group,
           ns_group * ghz, ghz, ns_group * ghz / 4);
    hipFree(d);
}

int main(int argc, char** argv) {
    const float ghz = argc > 1 ? atof(argv[1]) : 2.4f;
#define ROW(E, C, A)                                                                     \
    run<E, C, A, 0, true>("1 wave/SIMD, MFMA + fillers in one stream", 256, ghz);        \
    run<E, C, A, 0, false>("1 wave/SIMD, fillers only", 256, ghz);                       \
    run<E, C, A, 0, true>("2 waves/SIMD, both MFMA + fillers", 512, ghz);                \
    run<E, C, A, 1, true>("2 waves/SIMD, one MFMA-only, one fillers-only", 512, ghz);
    run<0, 0, 0, 0, true>("1 wave/SIMD, MFMA only", 256, ghz);
    run<0, 0, 0, 0, true>("2 waves/SIMD, MFMA only", 512, ghz);
    ROW(0, 0, 8)
    ROW(0, 0, 16)
    ROW(0, 0, 24)
    ROW(0, 0, 32)
    ROW(4, 0, 0)
    ROW(8, 0, 0)
    ROW(12, 0, 0)
    ROW(0, 8, 0)
    ROW(4, 2, 4)
    ROW(7, 4, 7)    // the forward attention kernel's mix per 4 MFMAs at head_dim 64
    ROW(8, 4, 16)   // the dK/dV backward's mix
    printf("---- dependent accumulator chains (NACC = accumulators cycled) ----\n");
#define CH(E, C, A)                                                                      \
    run<E, C, A, 0, true, 1>("1 wave/SIMD, 1 accumulator chain", 256, ghz);             \
    run<E, C, A, 0, true, 2>("1 wave/SIMD, 2 accumulator chains", 256, ghz);            \
    run<E, C, A, 0, true, 4>("1 wave/SIMD, 4 accumulator chains", 256, ghz);            \
    run<E, C, A, 0, true, 2>("2 waves/SIMD, 2 accumulator chains", 512, ghz);
    CH(0, 0, 0)
    CH(0, 0, 4)
    CH(4, 0, 0)
    CH(7, 4, 7)
    CH(12, 0, 6)
    return 0;
}
