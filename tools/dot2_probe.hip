// Which VALU instructions co-issue with v_mfma_f32_32x32x16_bf16 on gfx950?  For each candidate: time of a group of four MFMAs
// with N of them behind every MFMA, against the MFMA-only group (64 ns at ~2 GHz) and against the candidate alone.
//   hipcc --offload-arch=gfx950 -O3 tools/dot2_probe.hip -o tools/dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_ITERS 4096
enum { ADD2, PK_ADD, DOT2, PK_MUL, MOV64, MOV32X2, CVT_PK, EXP, FMA, PK_FMA, MAX3, NOPS };
static const char* kName[NOPS] = {"2x v_add_f32", "v_pk_add_f32", "v_dot2_f32_bf16", "v_pk_mul_f32", "v_mov_b64", "2x v_mov_b32", "v_cvt_pk_bf16_f32",
                                  "v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max3_f32"};
template <int OP, int PER_MFMA, bool MFMA>
__global__ __launch_bounds__(256) void k(float* out) {
    f32x16 acc[4];
    float a[8];
    unsigned pk[8];
    f32x2 pa[8];
    bf16x8 fa, fb;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) { a[i] = i * 0.01f; pk[i] = 0x3f803f80u; pa[i] = (f32x2){1.f, 2.f}; fa[i] = (__bf16)0.001f; fb[i] = (__bf16)0.002f; }
    const unsigned ones = 0x3f803f80u;
    const float one = 1.f;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int i = 0; i < PER_MFMA; ++i) {
                const int r = (q * PER_MFMA + i) & 7, r2 = (r + 4) & 7;
                if (OP == ADD2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(one)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r2]) : "v"(one)); }
                if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[r]) : "v"(pa[r2]));
                if (OP == DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[r]) : "v"(pk[r]), "v"(ones));
                if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[r]) : "v"(pa[r2]));
                if (OP == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(pa[r]) : "v"(pa[r2]));
                if (OP == MOV32X2) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[r]) : "v"(a[r2])); asm volatile("v_mov_b32 %0, %1" : "=v"(pk[r]) : "v"(pk[r2])); }
                if (OP == CVT_PK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[r]) : "v"(a[r]), "v"(a[r2]));
                if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[r]));
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[r]) : "v"(one));
                if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[r]) : "v"(pa[r2]));
                if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(a[r2]), "v"(one));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i][0] + pa[i][1] + __uint_as_float(pk[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP, int P, bool M>
float run() {
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP, P, M><<<256, 256>>>(d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP, P, M><<<256, 256>>>(d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms * 1e6f / N_ITERS;
}
template <int OP>
void row() {
    printf("%-20s alone x8: %6.1f ns | with 4 MFMAs: x4 %6.1f  x8 %6.1f  x16 %6.1f ns per group\n", kName[OP], run<OP, 2, false>(), run<OP, 1, true>(), run<OP, 2, true>(),
           run<OP, 4, true>());
}
int main() {
    printf("4 MFMAs alone: %.1f ns per group\n", run<ADD2, 0, true>());
    row<ADD2>(); row<PK_ADD>(); row<PK_MUL>(); row<PK_FMA>(); row<DOT2>(); row<MOV64>(); row<MOV32X2>(); row<CVT_PK>(); row<EXP>(); row<FMA>(); row<MAX3>();
    return 0;
}
