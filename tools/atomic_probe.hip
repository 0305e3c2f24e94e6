// fp32 atomic-add throughput of the fused attention-backward dQ pattern, by memory scope (tools/README.md).
//   agent scope     : global_atomic_add_f32 ... sc1  -> executed beyond the XCD's L2 (device-coherent)
//   workgroup scope : global_atomic_add_f32          -> executed IN the issuing XCD's L2 (coherent only among that XCD's CUs)
// Every workgroup reads HW_REG_XCC_ID and adds into the region of ITS XCD, so the workgroup-scope form is placement-independent.
// Pattern per workgroup (8 waves) and 64-row tile: wave w adds a 32x32 fp32 piece, 16 instructions of 2 rows x 128 B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0 agent atomics, 1 workgroup-scope atomics, 2 plain stores (traffic floor), 3 nothing (loop overhead)
__global__ __launch_bounds__(512) void probe(float* base, size_t region_floats, int nt, int regions_per_xcd, int* xcc_count) {
    const int xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID, bits [3:0]
    if (threadIdx.x == 0) atomicAdd(xcc_count + xcc, 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5;
    const int qblk = wave & 1, dblk = (wave >> 1) & 1;
    float* reg = base + ((size_t)xcc * regions_per_xcd + (blockIdx.x / 8) % regions_per_xcd) * region_floats;
    const int start = (int)(((long)(blockIdx.x / 8) * 37) % nt);
    float v = 1.0f;
    for (int i = 0; i < nt; ++i) {
        int tq = i + start; if (tq >= nt) tq -= nt;
        float* p = reg + (size_t)(tq * 64 + qblk * 32) * 64 + dblk * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (MODE == 0) __hip_atomic_fetch_add(p + row * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) __hip_atomic_fetch_add(p + row * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) p[row * 64] = v;
            else asm volatile("" ::"v"(p + row * 64));
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int S = 17776, nt = (S + 63) / 64;
    const size_t region = (size_t)nt * 64 * 64;
    int nwg = argc > 1 ? atoi(argv[1]) : 512;
    int passes = argc > 2 ? atoi(argv[2]) : 4;
    float* buf; int* cnt;
    CHECK(hipMalloc(&cnt, 8 * sizeof(int)));
    for (int rpx = 1; rpx <= 4; rpx *= 2) {
        CHECK(hipMalloc(&buf, region * 8 * rpx * sizeof(float)));
        for (int mode = 0; mode < 4; ++mode) {
            CHECK(hipMemset(buf, 0, region * 8 * rpx * sizeof(float)));
            CHECK(hipMemset(cnt, 0, 8 * sizeof(int)));
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            auto launch = [&]() {
                if (mode == 0) probe<0><<<nwg, 512>>>(buf, region, nt, rpx, cnt);
                if (mode == 1) probe<1><<<nwg, 512>>>(buf, region, nt, rpx, cnt);
                if (mode == 2) probe<2><<<nwg, 512>>>(buf, region, nt, rpx, cnt);
                if (mode == 3) probe<3><<<nwg, 512>>>(buf, region, nt, rpx, cnt);
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            for (int p = 0; p < passes; ++p) launch();
            CHECK(hipEventRecord(b));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            ms /= passes;
            const double bytes = (double)nwg * nt * 8 * 16 * 64 * 4;
            // correctness of the scoped form: total sum must equal the number of adds
            double sum = 0; int xc[8];
            if (mode <= 1) {
                std::vector<float> h(region * 8 * rpx);
                CHECK(hipMemcpy(h.data(), buf, h.size() * sizeof(float), hipMemcpyDeviceToHost));
                for (float x : h) sum += x;
            }
            CHECK(hipMemcpy(xc, cnt, sizeof(xc), hipMemcpyDeviceToHost));
            const double expect = (double)(passes + 1) * nwg * nt * 8 * 16 * 64;
            printf("regions/xcd %d mode %d (%s): %8.3f ms  %7.1f GB/s of 4-byte adds  sum/expected %.6f  wg per xcc %d %d %d %d %d %d %d %d\n", rpx, mode,
                   mode == 0 ? "agent atomics" : mode == 1 ? "workgroup-scope atomics" : mode == 2 ? "plain stores" : "no memory op", ms, bytes / ms / 1e6,
                   mode <= 1 ? sum / expect : 0.0, xc[0], xc[1], xc[2], xc[3], xc[4], xc[5], xc[6], xc[7]);
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
