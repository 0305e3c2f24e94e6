// Micro-benchmarks of per-instruction issue cost on gfx950 (used to size the attention softmax).  hipcc tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define N_ITERS 4096
#define U 16

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float v[U];
    f32x2 w[U / 2];
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < U / 2; ++i) w[i] = (f32x2){v[2 * i], v[2 * i + 1]};
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (OP == 0) v[i] = __builtin_fmaf(v[i], a, b);
            if (OP == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.f + v[i];           // exp + fma
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            if (OP == 6) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[i]), "v"(a)); v[i] = __uint_as_float(r); }
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < U / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"((f32x2){a, a}), "v"((f32x2){b, b}));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < U / 2; ++i) s += w[i][0] + w[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int waves_per_simd, int insts_per_iter) {
    float* d;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves -> 1 wave per SIMD per block
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)waves_per_simd * N_ITERS * insts_per_iter;
    const double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-28s waves/SIMD %d: %8.3f ms  %6.2f ns/wave-inst/SIMD (= %5.2f cyc @2.4GHz, %5.2f @1.9GHz)\n", name, waves_per_simd, ms, ns_per_inst,
           ns_per_inst * 2.4, ns_per_inst * 1.9);
    hipFree(d);
    return ns_per_inst;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<3>("v_fma_f32", w, U);
        run<2>("v_exp_f32", w, U);
        run<4>("v_pk_fma_f32", w, U / 2);
        run<5>("v_max3_f32", w, U);
        run<6>("v_cvt_pk_bf16_f32", w, U);
    }
    return 0;
}
