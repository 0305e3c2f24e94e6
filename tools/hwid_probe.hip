// Where do the 8 waves of a 512-thread workgroup land?  Prints HW_REG_HW_ID fields per wave for a few workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hwid; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    // keep the workgroup alive for a while so that later ones land next to running ones
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
    unsigned* d; const int nb = 1024;
    hipMalloc(&d, nb * 16 * sizeof(unsigned));
    k<<<nb, 512>>>(d);
    hipDeviceSynchronize();
    static unsigned h[nb * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 9, 300, 777}) {
        printf("wg %4d:", b);
        for (int w = 0; w < 8; ++w) { unsigned x = h[(b * 8 + w) * 2]; printf("  w%d[simd %u wave %u cu %u se %u raw %08x]", w, (x >> 4) & 3, x & 15, (x >> 8) & 15, (x >> 13) & 7, x); }
        printf("\n");
    }
    int hist[16] = {0};
    for (int b = 0; b < nb; ++b) { int c[4] = {0, 0, 0, 0}; for (int w = 0; w < 8; ++w) c[(h[(b * 8 + w) * 2] >> 4) & 3]++; int ok = (c[0] == 2 && c[1] == 2 && c[2] == 2 && c[3] == 2); hist[ok]++;
        int pairs_adjacent = 1; for (int w = 0; w < 8; w += 2) if (((h[(b * 8 + w) * 2] >> 4) & 3) != ((h[(b * 8 + w + 1) * 2] >> 4) & 3)) pairs_adjacent = 0; hist[2 + pairs_adjacent]++;
        int pairs_stride4 = 1; for (int w = 0; w < 4; ++w) if (((h[(b * 8 + w) * 2] >> 4) & 3) != ((h[(b * 8 + w + 4) * 2] >> 4) & 3)) pairs_stride4 = 0; hist[4 + pairs_stride4]++; }
    printf("2 waves on each SIMD: %d of %d workgroups;  (2k,2k+1) share a SIMD: %d;  (w,w+4) share a SIMD: %d\n", hist[1], nb, hist[3], hist[5]);
    return 0;
}
