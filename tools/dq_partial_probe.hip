// What a 5-product attention backward at head_dim 128 would have to move.
// With dK/dV stationary over Kw keys per workgroup, every workgroup emits one fp32 [64 q x 128] dQ tile per 64 queries and key block;
// the tiles are reduced over the S / Kw key blocks by a merge pass.  This probe runs ONLY that traffic -- the emitting stores in the
// dK/dV kernel's order (no MFMA work at all: the floor of what the fused kernel would add) and the merge (read S/Kw partials, write bf16) --
// at the cfg5 self-attention shape, so the cost can be set against the two matrix products (2 x 2 S^2 d BH FLOPs) the fusion removes.
//   hipcc --offload-arch=gfx950 -O3 tools/dq_partial_probe.hip -o /tmp/dq_partial_probe && /tmp/dq_partial_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid (key blocks, BH); 256 threads; tile of 64 x 128 fp32 = 32 KiB = 8 x (256 lanes x 16 B)
__global__ __launch_bounds__(256) void emit(float* __restrict__ part, int nq64, int nkb) {
    const int kb = blockIdx.x, bh = blockIdx.y;
    float* base = part + ((size_t)bh * nkb + kb) * (size_t)nq64 * 8192;
    const f32x4 v = {1.f, 2.f, 3.f, (float)kb};
    for (int t = 0; t < nq64; ++t) {
        float* p = base + (size_t)t * 8192 + threadIdx.x * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + i * 1024));
    }
}

// dq[bh][row][128] (bf16) = sum over key blocks; one thread = 8 consecutive features of one row
__global__ __launch_bounds__(256) void merge(const float* __restrict__ part, unsigned short* __restrict__ dq, int nq64, int nkb, size_t rows_total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // over BH * rows * 16
    if (i >= rows_total * 16) return;
    const size_t row = i >> 4;
    const int c = (int)(i & 15) * 8;
    const size_t rows_per_bh = (size_t)nq64 * 64;
    const size_t bh = row / rows_per_bh, r = row % rows_per_bh;
    const float* p = part + (bh * nkb) * rows_per_bh * 128 + r * 128 + c;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    for (int kb = 0; kb < nkb; ++kb) {
        const f32x4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)), x1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
        a0 += x0; a1 += x1;
        p += rows_per_bh * 128;
    }
    unsigned short o[8];
    const float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (unsigned short)(__float_as_uint(f[j]) >> 16);
    *reinterpret_cast<uint4*>(dq + row * 128 + c) = *reinterpret_cast<const uint4*>(o);
}

int main() {
    const int S = 18480, BH = 48, nq64 = (S + 63) / 64;
    const double saved_flops = 2.0 * 2.0 * (double)S * S * 128 * BH;          // the S and dP recomputes of the dQ kernel
    printf("cfg5 self-attention backward, S = %d, B*H = %d, head_dim 128: the two removed products = %.2f TFLOP = %.2f ms at the kernels' 1.31 PFLOP/s executed\n", S, BH,
           saved_flops * 1e-12, saved_flops / 1.31e15 * 1e3);
    unsigned short* dq;
    CHECK(hipMalloc(&dq, (size_t)BH * nq64 * 64 * 128 * 2));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int kws[] = {128, 256, 512, 1024, 2048};
    for (int kw : kws) {
        const int nkb = (S + kw - 1) / kw;
        const size_t bytes = (size_t)BH * nkb * nq64 * 8192 * 4;
        float* part;
        if (hipMalloc(&part, bytes) != hipSuccess) { printf("Kw %4d: %.1f GB of partials do not fit\n", kw, bytes * 1e-9); continue; }
        float best_e = 1e9f, best_m = 1e9f;
        for (int it = 0; it < 3; ++it) {
            CHECK(hipEventRecord(e0));
            emit<<<dim3(nkb, BH), 256>>>(part, nq64, nkb);
            CHECK(hipEventRecord(e1));
            const size_t rows_total = (size_t)BH * nq64 * 64;
            merge<<<(unsigned)((rows_total * 16 + 255) / 256), 256>>>(part, dq, nq64, nkb, rows_total);
            CHECK(hipEventRecord(e2));
            CHECK(hipEventSynchronize(e2));
            float a, b;
            CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
            best_e = a < best_e ? a : best_e; best_m = b < best_m ? b : best_m;
        }
        printf("Kw %4d keys per workgroup: %6.1f GB of fp32 dQ partials; stores alone %7.2f ms (%5.2f TB/s), merge %7.2f ms (%5.2f TB/s), together %7.2f ms\n", kw,
               bytes * 1e-9, best_e, bytes / best_e * 1e-9, best_m, bytes / best_m * 1e-9, best_e + best_m);
        CHECK(hipFree(part));
    }
    return 0;
}
