// What does this part stream at, for the access mixes of the elementwise kernels?  bf16 data as 16-byte chunks, 874 MB per tensor
// (the [2,17776,12288] GELU operand).  Variants: chunks in flight per thread (U), grid size.  read-only sum / copy (1R+1W) /
// 2R+1W / 2R+2W.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, int NR, int NW>
__global__ __launch_bounds__(256) void stream(const u4* __restrict__ a, const u4* __restrict__ b, u4* __restrict__ c, u4* __restrict__ d, int64_t n, unsigned* sink) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    u4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
        u4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * stride;
            if (j < n) { x[u] = a[j]; if (NR > 1) y[u] = b[j]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * stride;
            if (j < n) {
                u4 r = x[u];
                if (NR > 1) r = r ^ y[u];
                if (NW > 0) c[j] = r;
                if (NW > 1) d[j] = r + 1u;
                if (NW == 0) acc = acc ^ r;
            }
        }
    }
    if (NW == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int U, int NR, int NW>
int run(const char* what, u4* a, u4* b, u4* c, u4* d, int64_t n, int grid, unsigned* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream<U, NR, NW><<<grid, 256>>>(a, b, c, d, n, sink);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) stream<U, NR, NW><<<grid, 256>>>(a, b, c, d, n, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-10s U=%d grid=%6d : %7.1f us  %6.2f TB/s\n", what, U, grid, ms * 1e3, (double)(NR + NW) * n * 16 / ms / 1e9);
    return 0;
}

int main() {
    const int64_t n = (int64_t)2 * 17776 * 12288 * 2 / 16;
    u4 *a, *b, *c, *d; unsigned* sink;
    CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMalloc(&c, n * 16)); CHECK(hipMalloc(&d, n * 16)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 1, n * 16)); CHECK(hipMemset(b, 2, n * 16));
    for (int grid : {2048, 4096, 8192, 32768}) {
        run<1, 1, 0>("read", a, b, c, d, n, grid, sink);  run<4, 1, 0>("read", a, b, c, d, n, grid, sink);
        run<1, 1, 1>("1R+1W", a, b, c, d, n, grid, sink); run<2, 1, 1>("1R+1W", a, b, c, d, n, grid, sink); run<4, 1, 1>("1R+1W", a, b, c, d, n, grid, sink);
        run<2, 2, 1>("2R+1W", a, b, c, d, n, grid, sink); run<4, 2, 1>("2R+1W", a, b, c, d, n, grid, sink);
        run<2, 2, 2>("2R+2W", a, b, c, d, n, grid, sink); run<4, 2, 2>("2R+2W", a, b, c, d, n, grid, sink);
    }
    return 0;
}
