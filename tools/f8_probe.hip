// Probe of the gfx950 fp8 matrix path the cfg5 attention forward is built on (run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/f8_probe.hip -o /tmp/f8_probe && /tmp/f8_probe):
//   1. operand / result layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, against a host product;
//   2. what the E8M0 scale operands do (per lane: row i = lane % 32 of A resp. column j of B, k-block lane / 32);
//   3. v_cvt_scalef32_pk_fp8_f32 / v_cvt_pk_fp8_f32: scaling direction, rounding, saturation, byte placement;
//   4. issue rate of the instruction (cycles per MFMA on one wave, 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v2s __attribute__((ext_vector_type(2)));

static float e4m3_to_f(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}

__global__ void k_mfma(const uint8_t* A /*[32][64]*/, const uint8_t* B /*[64][32] k-major*/, float* D /*[32][32]*/, int layout, const int* sa, const int* sb) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int p = 0; p < 32; ++p) {
        int k;
        if (layout == 0) k = 32 * h + p;                       // 32 consecutive k per lane
        else k = 16 * h + (p & 15) + 32 * (p >> 4);            // two K = 32 halves, 16 consecutive k of each per lane
        a.b[p] = A[i * 64 + k];
        b.b[p] = B[k * 32 + i];
    }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

__global__ void k_cvt(const float* x, float scale, uint32_t* out) {
    const int l = threadIdx.x;
    v2s old = {0, 0};
    v2s lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * l], x[2 * l + 1], scale, false);
    v2s both = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(lo, x[2 * l], x[2 * l + 1], scale, true);
    union { v2s v; uint32_t u; } cv; cv.v = both;
    out[2 * l] = cv.u;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * l], x[2 * l + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * l + 1], x[2 * l], w, true);
    out[2 * l + 1] = (uint32_t)w;
}

__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    v8i a, b;
    for (int r = 0; r < 8; ++r) { a[r] = 0x38383838 + threadIdx.x; b[r] = 0x30303030 + r; }
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    srand(3);
    std::vector<uint8_t> A(32 * 64), B(64 * 32);
    for (auto& v : A) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }     // any e4m3 but NaN
    for (auto& v : B) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }
    // keep magnitudes moderate so that the fp32 accumulation order does not matter for the comparison
    for (auto& v : A) v = (v & 0x87) | (((v >> 3) & 3) + 6) << 3;
    for (auto& v : B) v = (v & 0x87) | (((v >> 3) & 3) + 6) << 3;
    uint8_t *dA, *dB; float* dD; int *dsa, *dsb;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    std::vector<int> sa(64, 127), sb(64, 127);
    std::vector<float> D(32 * 32);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: unit scales; mode 1: uniform non-unit scales; mode 2: per-lane scales (row / column and k-block dependent)
        for (int l = 0; l < 64; ++l) {
            sa[l] = mode == 0 ? 127 : mode == 1 ? 129 : 124 + (l % 7);
            sb[l] = mode == 0 ? 127 : mode == 1 ? 126 : 125 + (l % 5);
        }
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        for (int layout = 0; layout < 2; ++layout) {
            k_mfma<<<1, 64>>>(dA, dB, dD, layout, dsa, dsb);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, ref_max = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) {
                        // scale hypothesis: lane (i, k-block kb) of A carries the scale of A[i][32 kb .. 32 kb + 31]; same for B's column j
                        const int kb = k >> 5;
                        ref += (double)e4m3_to_f(A[i * 64 + k]) * ldexp(1.0, sa[i + 32 * kb] - 127) * (double)e4m3_to_f(B[k * 32 + j]) * ldexp(1.0, sb[j + 32 * kb] - 127);
                    }
                    worst = fmax(worst, fabs(ref - D[i * 32 + j]));
                    ref_max = fmax(ref_max, fabs(ref));
                }
            printf("mfma_scale 32x32x64 e4m3: scale mode %d, operand layout hypothesis %d (0: 32 consecutive k per lane, 1: 16 + 16): max |diff| %.3g of max |ref| %.3g -> %s\n",
                   mode, layout, worst, ref_max, worst <= 1e-5 * ref_max ? "MATCH" : "no");
        }
    }
    // cvt probes
    const int N = 128;
    std::vector<float> x(N);
    const float vals[] = {0.f, 1.f, 1.0625f, 1.1875f, 3.f, 448.f, 460.f, 480.f, 1000.f, 1e30f, 0.001953125f, 0.0009765625f, 0.0004f, -2.5f, 17.f, 0.3f};
    for (int i = 0; i < N; ++i) x[i] = vals[i % 16] * (i >= 64 ? 4.f : 1.f);
    float* dx; uint32_t* dout;
    hipMalloc(&dx, N * 4); hipMalloc(&dout, N * 4);
    hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice);
    for (float scale : {1.f, 4.f, 0.25f}) {
        k_cvt<<<1, 64>>>(dx, scale, dout);
        std::vector<uint32_t> o(N);
        hipMemcpy(o.data(), dout, N * 4, hipMemcpyDeviceToHost);
        printf("cvt_scalef32_pk_fp8_f32 scale %g: (x0, x1) -> bytes [lo word | hi word] decoded;  cvt_pk_fp8_f32 (no scale) beside it\n", scale);
        for (int l = 0; l < 8; ++l) {
            const uint32_t u = o[2 * l], w = o[2 * l + 1];
            printf("   x = (%g, %g): scaled cvt 0x%08x = [%g %g | %g %g]   plain cvt 0x%08x = [%g %g | %g %g]\n", x[2 * l], x[2 * l + 1], u, e4m3_to_f(u & 255), e4m3_to_f((u >> 8) & 255),
                   e4m3_to_f((u >> 16) & 255), e4m3_to_f(u >> 24), w, e4m3_to_f(w & 255), e4m3_to_f((w >> 8) & 255), e4m3_to_f((w >> 16) & 255), e4m3_to_f(w >> 24));
        }
    }
    // issue rate
    float* dr; unsigned long long* dc;
    hipMalloc(&dr, 64 * 4); hipMalloc(&dc, 8);
    const int iters = 4096;
    k_rate<<<1, 64>>>(dr, dc, iters);
    unsigned long long c;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("issue rate: %.1f counter ticks per v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, one wave, 4 accumulators; s_memtime counts at 100 MHz: compare with the wall clock below)\n", (double)c / (4.0 * iters));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<<<256 * 4, 64>>>(dr, dc, iters);
    hipEventRecord(e0);
    k_rate<<<256 * 4, 64>>>(dr, dc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chip: 1024 waves x %d MFMAs in %.3f ms = %.0f TFLOP/s (2 x 32 x 32 x 64 per MFMA)\n", 4 * iters, ms, 1024.0 * 4 * iters * 2 * 32 * 32 * 64 / (ms * 1e-3) / 1e12);
    return 0;
}
