// Which hipBLASLt epilogues have bf16 solutions on this device at the feed-forward shapes?  hipcc tools/hipblaslt_probe.cpp -lhipblaslt -o tools/hipblaslt_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
int main() {
    hipblasLtHandle_t h;
    printf("create %d\n", (int)hipblasLtCreate(&h));
    const int64_t shapes[3][3] = {{35552, 12288, 3072}, {300, 1024, 256}, {35552, 12288, 3072}};
    const hipblasLtEpilogue_t eps[] = {HIPBLASLT_EPILOGUE_DEFAULT, HIPBLASLT_EPILOGUE_BIAS, HIPBLASLT_EPILOGUE_GELU_BIAS, HIPBLASLT_EPILOGUE_GELU_AUX,
                                       HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, HIPBLASLT_EPILOGUE_DGELU};
    const char* names[] = {"DEFAULT", "BIAS", "GELU_BIAS", "GELU_AUX", "GELU_AUX_BIAS", "DGELU"};
    for (int s = 0; s < 2; ++s)
        for (int e = 0; e < 6; ++e)
            for (int auxt = 0; auxt < 3; ++auxt) {
                const int64_t M = shapes[s][0], N = shapes[s][1], K = shapes[s][2];
                hipblasLtMatmulDesc_t d;
                hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F);
                hipblasOperation_t t = HIPBLAS_OP_T, n = HIPBLAS_OP_N;
                hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &t, sizeof(t));
                hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &n, sizeof(n));
                hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &eps[e], sizeof(eps[e]));
                hipDataType bf = HIP_R_16BF, f32 = HIP_R_32F;
                const bool has_aux = e >= 3;
                if (!has_aux && auxt > 0) { hipblasLtMatmulDescDestroy(d); continue; }
                if (has_aux) {
                    int64_t ld = N;
                    void* p = (void*)256;
                    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &p, sizeof(p));
                    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ld, sizeof(ld));
                    if (auxt == 1) hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_DATA_TYPE, &bf, sizeof(bf));
                    if (auxt == 2) hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_DATA_TYPE, &f32, sizeof(f32));
                }
                if (e == 1 || e == 2 || e == 4) {
                    void* p = (void*)256;
                    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &p, sizeof(p));
                    hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bf, sizeof(bf));
                }
                hipblasLtMatrixLayout_t a, b, c;
                hipblasLtMatrixLayoutCreate(&a, bf, K, N, K);
                hipblasLtMatrixLayoutCreate(&b, bf, K, M, K);
                hipblasLtMatrixLayoutCreate(&c, bf, N, M, N);
                hipblasLtMatmulPreference_t pref;
                hipblasLtMatmulPreferenceCreate(&pref);
                uint64_t ws = 64u << 20;
                hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
                hipblasLtMatmulHeuristicResult_t r[4];
                int found = 0;
                hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, d, a, b, c, c, pref, 4, r, &found);
                printf("M %6ld N %6ld K %5ld  %-14s aux_type %s : status %d found %d\n", (long)M, (long)N, (long)K, names[e],
                       auxt == 0 ? "unset" : auxt == 1 ? "bf16 " : "f32  ", (int)st, found);
            }
    return 0;
}
