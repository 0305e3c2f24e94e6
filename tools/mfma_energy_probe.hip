// The matrix pipe's energy floor on this part: a register-resident stream of MFMAs -- no LDS, no memory, no VALU -- timed AND
// bracketed by the socket energy counter, on random and on all-zero operands.  If the hand-scheduled attention kernels and the vendor's GEMM both cost
// ~1 J per executed bf16 TFLOP on random data, is that the kernels or the silicon?  This probe gives the silicon's number.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_energy_probe.hip -o /tmp/mfma_energy_probe -ldl && /tmp/mfma_energy_probe
// Variants: operands toggling on both sides every instruction / one stationary operand per accumulator (what the attention kernels do: K, V or Q, dO
// fragments stay in registers); 1 or 2 waves per SIMD; bf16 32x32x16 and e4m3 32x32x64 (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// MODE 0: bf16, A and B both change every MFMA; 1: bf16, B stationary per accumulator; 2: e4m3 K=64, both change; 3: e4m3, B stationary
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const uint32_t* __restrict__ data, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    if (MODE < 2) {
        bf16x8 A[8], B[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const uint4 ua = *reinterpret_cast<const uint4*>(data + ((size_t)(t * 16 + f) * 4 & 0xfffffc));
            const uint4 ub = *reinterpret_cast<const uint4*>(data + ((size_t)(t * 16 + 8 + f) * 4 & 0xfffffc));
            A[f] = *reinterpret_cast<const bf16x8*>(&ua);
            B[f] = *reinterpret_cast<const bf16x8*>(&ub);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int a = u & 3;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(u * 5 + 1) & 7], MODE == 1 ? B[a] : B[(u * 3) & 7], acc[a], 0, 0, 0);
            }
        }
    } else {
        i32x8 A[4], B[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                A[f][i] = (int)data[((size_t)(t * 64 + f * 8 + i)) & 0xffffff];
                B[f][i] = (int)data[((size_t)(t * 64 + 32 + f * 8 + i)) & 0xffffff];
            }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = u & 3;
                acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[(u * 3 + 1) & 3], MODE == 3 ? B[a] : B[u & 3], acc[a], 0, 0, 0, 127, 0, 127);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[t] = s;
}

// What the operand feed costs: the same stationary-operand bf16 stream with, per group of four MFMAs, RD ds_read_b128 fragment reads (conflict-free,
// lane-linear; the fragments read ARE the A operands of the next group) and VA x {2 v_exp_f32, 1 v_cvt_pk_bf16_f32, 2 v_add_f32} (the softmax mix of the
// attention forward: 5 VALU per MFMA at VA = 4; the backward kernels carry 3 per MFMA).
template <int RD, int VA, int DM = 0>
__global__ __launch_bounds__(256, 1) void probe_mix(const uint32_t* __restrict__ data, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[16384];      // 64 KiB of random fragments
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = data[(size_t)(blockIdx.x * 16384 + i) & 0xffffff];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 A[2][4], B[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const uint4 ua = *reinterpret_cast<const uint4*>(data + ((size_t)(t * 16 + f) * 4 & 0xfffffc));
        const uint4 ub = *reinterpret_cast<const uint4*>(data + ((size_t)(t * 16 + 8 + f) * 4 & 0xfffffc));
        A[0][f] = A[1][f] = *reinterpret_cast<const bf16x8*>(&ua);
        B[f] = *reinterpret_cast<const bf16x8*>(&ub);
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = -1.0f - 0.01f * (float)((threadIdx.x + i) & 31);
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    // DM: one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) per DM groups of four MFMAs, streamed from an 8 MiB window that every
    // workgroup walks (L2 / infinity-cache resident, like the K / V tiles all q-tiles of a head re-read) into the upper 32 KiB of the LDS array
    __shared__ __attribute__((aligned(1024))) uint8_t ring[32768];
    uint32_t rsrc[4];
    {
        const uint64_t a = (uint64_t)data;
        rsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        rsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        rsrc[2] = 8u << 20;
        rsrc[3] = 0x00020000u;
    }
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 rs = {rsrc[0], rsrc[1], rsrc[2], rsrc[3]};
    uint32_t voff = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024 + (blockIdx.x & 7) * 4096;
    const uint32_t ring0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)ring + (threadIdx.x >> 6) * 8192);
    int dmc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (RD >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[nxt][0]) : "v"(addr), "n"(4096 * 0 + 16384 * (3 & 1)));
            if (RD >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[nxt][1]) : "v"(addr), "n"(4096 * 1));
            if (RD >= 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[nxt][2]) : "v"(addr), "n"(4096 * 2));
            if (RD >= 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[nxt][3]) : "v"(addr), "n"(4096 * 3));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][u], B[u], acc[u], 0, 0, 0);
                if (u < VA) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[0]), "+v"(v[1]));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[2]) : "v"(v[3]), "v"(v[4]));
                    asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(v[5]), "+v"(v[6]) : "v"(v[0]), "v"(v[1]));
                    asm volatile("v_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1" : "+v"(v[0]), "+v"(v[1]));     // keep exp's argument in range (2 more, cheap)
                }
            }
            if (RD >= 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[nxt][0]), "+v"(A[nxt][1]), "+v"(A[nxt][2]), "+v"(A[nxt][3]));
            addr ^= 2048u * (uint32_t)(g + 1);
            if (DM > 0 && (++dmc % DM) == 0) {
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(6)"
                             : "=&s"(keep) : "s"(ring0 + (uint32_t)((dmc / DM) & 7) * 1024u), "v"(voff), "s"(rs) : "memory");
                voff = (voff + 32768u) & ((8u << 20) - 1u);
            }
        }
    }
    float s = v[2] + v[5] + v[6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[t] = s;
}

typedef int (*rsmi_init_t)(uint64_t);
typedef int (*rsmi_energy_t)(uint32_t, uint64_t*, float*, uint64_t*);
static rsmi_energy_t g_energy = nullptr;
static double joules() {
    if (!g_energy) return -1.0;
    uint64_t c = 0, ts = 0;
    float res = 0.f;
    if (g_energy(0, &c, &res, &ts) != 0) return -1.0;
    return (double)c * res * 1e-6;
}

template <int MODE>
static void run(const char* name, const uint32_t* data, float* out, int waves_per_simd, double seconds) {
    const int blocks = 256 * waves_per_simd;      // 4 waves per block: one per SIMD
    const double flop_per_iter = (MODE < 2 ? 16.0 * 2 * 32 * 32 * 16 : 8.0 * 2 * 32 * 32 * 64) * 4.0 * blocks;
    int iters = 100000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MODE><<<blocks, 256>>>(data, out, iters);            // warm-up, and the time of one launch
    CHECK(hipEventRecord(e0));
    probe<MODE><<<blocks, 256>>>(data, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms1;
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    const int n = (int)(seconds * 1e3 / ms1) + 1;
    CHECK(hipDeviceSynchronize());
    const double j0 = joules();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) probe<MODE><<<blocks, 256>>>(data, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    const double j1 = joules();
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = flop_per_iter * iters * n;
    printf("%-58s %d wave/SIMD  %7.1f ms  %7.1f TFLOP/s  %7.1f W  %6.3f TFLOP/J\n", name, waves_per_simd, ms, flop / ms / 1e9, j1 > j0 ? (j1 - j0) / (ms * 1e-3) : -1.0,
           j1 > j0 ? flop / (j1 - j0) / 1e12 : -1.0);
}

template <int RD, int VA, int DM = 0>
static void run_mix(const char* name, const uint32_t* data, float* out, double seconds) {
    const int blocks = 256;
    const double flop_per_iter = 16.0 * 2 * 32 * 32 * 16 * 4.0 * blocks;
    int iters = 100000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe_mix<RD, VA, DM><<<blocks, 256>>>(data, out, iters);
    CHECK(hipEventRecord(e0));
    probe_mix<RD, VA, DM><<<blocks, 256>>>(data, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms1;
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    const int n = (int)(seconds * 1e3 / ms1) + 1;
    CHECK(hipDeviceSynchronize());
    const double j0 = joules();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) probe_mix<RD, VA, DM><<<blocks, 256>>>(data, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    const double j1 = joules();
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = flop_per_iter * iters * n;
    printf("%-58s 1 wave/SIMD  %7.1f ms  %7.1f TFLOP/s  %7.1f W  %6.3f TFLOP/J\n", name, ms, flop / ms / 1e9, j1 > j0 ? (j1 - j0) / (ms * 1e-3) : -1.0,
           j1 > j0 ? flop / (j1 - j0) / 1e12 : -1.0);
}

int main() {
    void* h = dlopen("librocm_smi64.so", RTLD_NOW);
    if (!h) h = dlopen("/opt/rocm/lib/librocm_smi64.so", RTLD_NOW);
    if (h) {
        rsmi_init_t init = (rsmi_init_t)dlsym(h, "rsmi_init");
        g_energy = (rsmi_energy_t)dlsym(h, "rsmi_dev_energy_count_get");
        if (!init || init(0) != 0) g_energy = nullptr;
    }
    if (!g_energy) printf("no energy counter (librocm_smi64): W and TFLOP/J print as -1\n");
    const size_t words = 1u << 24;
    std::vector<uint32_t> host(words);
    uint32_t* data;
    float* out;
    CHECK(hipMalloc(&data, words * 4));
    CHECK(hipMalloc(&out, 512 * 256 * 4));
    for (int pass = 0; pass < 2; ++pass) {
        // random: bf16 N(0,1)-like bit patterns (sign random, exponent around 127, mantissa random); as e4m3 bytes the same words are random finite-ish
        // values (0x7f / 0xff = NaN bytes are remapped).  zeros: no datapath toggling.
        uint64_t s = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < words; ++i) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint32_t r = (uint32_t)(s >> 32);
            uint32_t lo = (r & 0x807fu) | ((124u + ((r >> 16) & 3u)) << 7), hi = ((r >> 8) & 0x807fu) | ((124u + ((r >> 20) & 3u)) << 7);
            host[i] = pass == 0 ? (lo | (hi << 16)) : 0u;
        }
        CHECK(hipMemcpy(data, host.data(), words * 4, hipMemcpyHostToDevice));
        printf("== operands: %s\n", pass == 0 ? "random bf16 (|x| in [0.125, 2)), random bytes as e4m3" : "all zero");
        run<0>("bf16 32x32x16, both operands change every MFMA", data, out, 1, 1.5);
        run<1>("bf16 32x32x16, one operand stationary per accumulator", data, out, 1, 1.5);
        run<1>("bf16 32x32x16, one operand stationary per accumulator", data, out, 2, 1.5);
        run<2>("e4m3 32x32x64 (mfma_scale f8f6f4), both operands change", data, out, 1, 1.5);
        run<3>("e4m3 32x32x64 (mfma_scale f8f6f4), one operand stationary", data, out, 1, 1.5);
        run_mix<0, 0>("bf16, stationary B, A double-buffered (mix baseline)", data, out, 1.5);
        run_mix<2, 0>("  + 0.5 ds_read_b128 fragment per MFMA", data, out, 1.5);
        run_mix<4, 0>("  + 1 ds_read_b128 fragment per MFMA", data, out, 1.5);
        run_mix<0, 2>("  + 3.5 VALU per MFMA (exp exp cvt add add mul mul on 2 of 4)", data, out, 1.5);
        run_mix<0, 4>("  + 7 VALU per MFMA (.. on 4 of 4)", data, out, 1.5);
        run_mix<4, 2>("  + 1 fragment read + 3.5 VALU per MFMA (~ the backward kernels)", data, out, 1.5);
        run_mix<2, 4>("  + 0.5 fragment read + 7 VALU per MFMA (~ the forward)", data, out, 1.5);
        run_mix<0, 0, 4>("  + 1 KiB LDS-DMA per 16 MFMAs (~ the dK/dV stream)", data, out, 1.5);
        run_mix<0, 0, 2>("  + 1 KiB LDS-DMA per 8 MFMAs (~ the forward / dQ stream)", data, out, 1.5);
        run_mix<0, 0, 1>("  + 1 KiB LDS-DMA per 4 MFMAs (~ a 256 x 128 GEMM tile: 12 per 32)", data, out, 1.5);
    }
    return 0;
}
