// What do hipBLASLt's fused GELU epilogues cost at the cfg2 feed-forward shapes, against the plain GEMM + the stand-alone GELU pass of the step?
//   hipcc -O2 tools/hipblaslt_epilogue_time.cpp -lhipblaslt -o /tmp/hipblaslt_epilogue_time && /tmp/hipblaslt_epilogue_time
// Row-major y[M,N] = x[M,K] W[N,K]^T is the column-major product C(N x M) = A^T(N x K) B(K x M) with A = W (K x N, lda K, OP_T), B = x (K x M, ldb K).
//   forward   FF1:  BIAS  |  GELU_BIAS (g only)  |  GELU_AUX_BIAS (g and the pre-activation, both bf16)
//   backward  FF2:  dg = df W2 (DEFAULT)  |  DGELU (dh = dg * gelu'(aux) in the epilogue)
// Every heuristic solution (up to 16) is timed, interleaved in rounds so that all of them see the same power state; prints first-of-heuristic and best.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { auto _s = (x); if ((int)_s != 0) { printf("error %d at %s:%d\n", (int)_s, __FILE__, __LINE__); exit(1); } } while (0)

struct Case { const char* name; int64_t M, N, K; hipblasLtEpilogue_t ep; bool bias, aux; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6, inner = 8;
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int64_t Mf = 35840, Mb = 36864, D = 3072, F = 12288;
    const Case cases[] = {
        {"FF1 fwd  BIAS            ", Mf, F, D, HIPBLASLT_EPILOGUE_BIAS, true, false},
        {"FF1 fwd  GELU_BIAS       ", Mf, F, D, HIPBLASLT_EPILOGUE_GELU_BIAS, true, false},
        {"FF1 fwd  GELU_AUX_BIAS   ", Mf, F, D, HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, true, true},
        {"FF2 dX   DEFAULT         ", Mf, F, D, HIPBLASLT_EPILOGUE_DEFAULT, false, false},
        {"FF2 dX   DGELU           ", Mf, F, D, HIPBLASLT_EPILOGUE_DGELU, false, true},
        {"FF2 dX   DEFAULT (36864) ", Mb, F, D, HIPBLASLT_EPILOGUE_DEFAULT, false, false},
        {"FF2 dX   DGELU   (36864) ", Mb, F, D, HIPBLASLT_EPILOGUE_DGELU, false, true},
    };
    const size_t maxMN = (size_t)Mb * F;
    unsigned short *A, *B, *C, *AUX, *BIAS;
    CK(hipMalloc(&A, (size_t)F * D * 2));
    CK(hipMalloc(&B, (size_t)Mb * D * 2));
    CK(hipMalloc(&C, maxMN * 2));
    CK(hipMalloc(&AUX, maxMN * 2));
    CK(hipMalloc(&BIAS, (size_t)F * 2));
    {   // bf16 N(0, 1)-ish operands (random mantissas matter for the power the MFMAs draw): small LCG on the host, repeated
        std::vector<unsigned short> hbuf(1 << 22);
        unsigned s = 12345;
        for (auto& v : hbuf) { s = s * 1664525u + 1013904223u; unsigned e = 0x3c + ((s >> 9) & 3); v = (unsigned short)(((s >> 16) & 0x8000) | (e << 7) | ((s >> 20) & 0x7f)); }
        auto fill = [&](unsigned short* p, size_t n) { for (size_t o = 0; o < n; o += hbuf.size()) CK(hipMemcpy(p + o, hbuf.data(), std::min(hbuf.size(), n - o) * 2, hipMemcpyHostToDevice)); };
        fill(A, (size_t)F * D); fill(B, (size_t)Mb * D); fill(AUX, maxMN); fill(BIAS, F);
    }
    void* ws;
    const uint64_t wsz = 256u << 20;
    CK(hipMalloc(&ws, wsz));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        hipblasLtMatmulDesc_t d;
        CK(hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t t = HIPBLAS_OP_T, n = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &t, sizeof(t)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &n, sizeof(n)));
        CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &c.ep, sizeof(c.ep)));
        hipDataType bf = HIP_R_16BF;
        if (c.aux) {
            int64_t ld = c.N;
            void* p = AUX;
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &p, sizeof(p)));
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ld, sizeof(ld)));
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_DATA_TYPE, &bf, sizeof(bf)));
        }
        if (c.bias) {
            void* p = BIAS;
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &p, sizeof(p)));
            CK(hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bf, sizeof(bf)));
        }
        hipblasLtMatrixLayout_t a, b, cl;
        CK(hipblasLtMatrixLayoutCreate(&a, bf, c.K, c.N, c.K));
        CK(hipblasLtMatrixLayoutCreate(&b, bf, c.K, c.M, c.K));
        CK(hipblasLtMatrixLayoutCreate(&cl, bf, c.N, c.M, c.N));
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        hipblasLtMatmulHeuristicResult_t r[16];
        int found = 0;
        hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(h, d, a, b, cl, cl, pref, 16, r, &found);
        if (hs != HIPBLAS_STATUS_SUCCESS || found == 0) { printf("%s : no solution (status %d)\n", c.name, (int)hs); continue; }
        const float alpha = 1.f, beta = 0.f;
        std::vector<std::vector<float>> ms(found);
        std::vector<bool> ok(found, true);
        for (int rd = 0; rd < rounds + 1; ++rd)
            for (int s = 0; s < found; ++s) {
                if (!ok[s]) continue;
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < inner; ++i)
                    if (hipblasLtMatmul(h, d, &alpha, A, a, B, b, &beta, C, cl, C, cl, &r[s].algo, ws, wsz, st) != HIPBLAS_STATUS_SUCCESS) { ok[s] = false; break; }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float t_ms;
                CK(hipEventElapsedTime(&t_ms, e0, e1));
                if (rd > 0 && ok[s]) ms[s].push_back(t_ms / inner);      // round 0 warms every solution up
            }
        int best = -1;
        std::vector<float> med(found, 1e9f);
        for (int s = 0; s < found; ++s)
            if (ok[s] && !ms[s].empty()) {
                std::sort(ms[s].begin(), ms[s].end());
                med[s] = ms[s][ms[s].size() / 2];
                if (best < 0 || med[s] < med[best]) best = s;
            }
        const double fl = 2.0 * c.M * c.N * c.K;
        printf("%s M %5ld N %5ld K %5ld : %2d solutions; heuristic-first %.3f ms (%.0f TF/s), best #%d %.3f ms (%.0f TF/s)\n", c.name, (long)c.M, (long)c.N, (long)c.K, found,
               med[0], fl / med[0] * 1e-9, best, med[best], fl / med[best] * 1e-9);
        fflush(stdout);
        hipblasLtMatmulPreferenceDestroy(pref);
        hipblasLtMatrixLayoutDestroy(a); hipblasLtMatrixLayoutDestroy(b); hipblasLtMatrixLayoutDestroy(cl);
        hipblasLtMatmulDescDestroy(d);
    }
    return 0;
}
