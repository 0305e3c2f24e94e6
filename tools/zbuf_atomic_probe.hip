// What bounds project_zbuf (videogpa_amd/csrc/scorer.hip: one 64-bit atomicMin(depth || index) per point-view into a [T, H, W] z-buffer)?  The same access pattern at
// the reference's scale (T = 10 views of 518 x 518, N = T * 518 * 518 points) with the pieces taken apart:
//   atomics only   no loads, no projection arithmetic: pixel from a hash (scattered cloud) or from the point index (point map: neighbouring lanes -> neighbouring pixels)
//   loads only     the 24 B of xyz + rgb per point-view, reduced into a register
//   both           = the kernel's memory behaviour
//   plain stores   the z-buffer write traffic without the read-modify-write
//   wg-scope       the atomic executed in the issuing XCD's L2 (NOT coherent across the 8 XCDs -- a what-if, not an option for the product)
// hipcc --offload-arch=gfx950 -O3 tools/zbuf_atomic_probe.hip -o /tmp/zbuf_atomic_probe && /tmp/zbuf_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0 atomics (agent scope), 1 loads only, 2 loads + atomics, 3 plain stores, 4 atomics at workgroup scope
template <int MODE, bool COHERENT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ pc, const float* __restrict__ colors, int64_t N, int HW, unsigned long long* __restrict__ zbuf,
                                             float* __restrict__ sink) {
    const int t = blockIdx.y;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        float z = 1.0f + (float)(i & 1023) * 1e-3f;
        if (MODE == 1 || MODE == 2) {
            const float x = pc[3 * i], y = pc[3 * i + 1], zz = pc[3 * i + 2];
            const float c0 = colors[3 * i], c1 = colors[3 * i + 1], c2 = colors[3 * i + 2];
            acc += x + y + zz + c0 + c1 + c2;
            z += zz * 1e-6f;
        }
        if (MODE != 1) {
            // point map: point i of frame f lands next to pixel (i mod HW) of every view (a small per-view shift); scattered: anywhere
            const uint32_t pix = COHERENT ? (uint32_t)((i + 7 * t) % HW) : hash32((uint32_t)i * 31u + (uint32_t)t) % (uint32_t)HW;
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (uint32_t)i;
            unsigned long long* p = zbuf + (size_t)t * HW + pix;
            if (MODE == 3) *p = key;
            else if (MODE == 4) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicMin(p, key);
        }
    }
    if (MODE == 1 || MODE == 2) if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int T = 10, H = 518, W = 518, HW = H * W;
    const int64_t N = (int64_t)T * HW;
    float *pc, *colors, *sink;
    unsigned long long* zbuf;
    CHECK(hipMalloc(&pc, N * 12)); CHECK(hipMalloc(&colors, N * 12)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&zbuf, (size_t)T * HW * 8));
    CHECK(hipMemset(pc, 0, N * 12)); CHECK(hipMemset(colors, 0, N * 12));
    const dim3 grid(2048, T);
    const char* names[5] = {"atomicMin only (agent scope)", "loads only (24 B per point-view)", "loads + atomicMin (the kernel's pattern)", "plain 8-byte stores instead",
                            "atomicMin at workgroup scope (what-if)"};
    for (int coh = 1; coh >= 0; --coh)
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            auto launch = [&]() {
#define L(M, C) probe<M, C><<<grid, 256>>>(pc, colors, N, HW, zbuf, sink)
                if (coh) { if (mode == 0) L(0, true); if (mode == 1) L(1, true); if (mode == 2) L(2, true); if (mode == 3) L(3, true); if (mode == 4) L(4, true); }
                else     { if (mode == 0) L(0, false); if (mode == 1) L(1, false); if (mode == 2) L(2, false); if (mode == 3) L(3, false); if (mode == 4) L(4, false); }
            };
            CHECK(hipMemset(zbuf, 0xff, (size_t)T * HW * 8));
            launch();
            CHECK(hipDeviceSynchronize());
            const int reps = 20;
            CHECK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) launch();
            CHECK(hipEventRecord(b));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            ms /= reps;
            const double ops = (double)T * N;
            printf("%-10s %-42s %7.3f ms  %7.1f G point-views/s  %7.1f GB/s of (24 B loads%s)\n", coh ? "point map" : "scattered", names[mode], ms, ops / ms / 1e6,
                   ops * (mode == 0 || mode == 3 || mode == 4 ? 8.0 : mode == 1 ? 24.0 : 32.0) / ms / 1e6, mode == 1 ? "" : " + 8 B z-buffer words");
        }
    return 0;
}
