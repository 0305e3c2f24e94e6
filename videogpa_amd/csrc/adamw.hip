// Fused optimizer step over ONE flat fp32 buffer of all LoRA parameters (66 M values at r=64): global L2 gradient
// norm (deterministic two-level fp64 reduction), clip-by-norm, AdamW -- two launches, no host sync.
// Replaces Lightning's gradient_clip_val=1.0 + torch.optim.AdamW(lr) at train/CogVideoX-5B/03_train.py:208-213,266.
// Semantics = torch.nn.utils.clip_grad_norm_ (coef = max_norm / (norm + 1e-6), clamped to 1) followed by
// torch.optim.AdamW (decoupled weight decay, bias-corrected moments).
#include "common.h"

#define OPT_THREADS 256
#define OPT_MAX_BLOCKS 1024

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    __shared__ double smem[16];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * OPT_THREADS) acc += g[i] * g[i];
    const double s = block_sum<double>((double)acc, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_finish_kernel(const double* __restrict__ partial, int nblk, float grad_scale,
                                                                     float* __restrict__ norm_out) {
    __shared__ double smem[16];
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += OPT_THREADS) s += partial[i];
    s = block_sum<double>(s, smem);
    if (threadIdx.x == 0) norm_out[0] = (float)(sqrt(s) * (double)grad_scale);
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                                              float wd, float bc1, float bc2_sqrt, float grad_scale, float max_norm,
                                                              const float* __restrict__ total_norm) {
    float coef = grad_scale;
    if (max_norm > 0.f && total_norm) {
        const float c = max_norm / (total_norm[0] + 1e-6f);
        coef *= fminf(c, 1.0f);
    }
    const float step_size = lr / bc1;
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = ga[j] * coef;
            pa[j] *= 1.f - lr * wd;
            ma[j] = beta1 * ma[j] + (1.f - beta1) * gj;
            va[j] = beta2 * va[j] + (1.f - beta2) * gj * gj;
            pa[j] -= step_size * ma[j] / (sqrtf(va[j]) / bc2_sqrt + eps);
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * OPT_THREADS) {
        const float gj = g[i] * coef;
        float pj = p[i] * (1.f - lr * wd);
        const float mj = beta1 * m[i] + (1.f - beta1) * gj;
        const float vj = beta2 * v[i] + (1.f - beta2) * gj * gj;
        pj -= step_size * mj / (sqrtf(vj) / bc2_sqrt + eps);
        p[i] = pj; m[i] = mj; v[i] = vj;
    }
}

static inline int opt_blocks(int64_t n) {
    int64_t nb = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
    if (nb < 1) nb = 1;
    if (nb > OPT_MAX_BLOCKS) nb = OPT_MAX_BLOCKS;
    return (int)nb;
}

extern "C" {

size_t vgpa_grad_norm_workspace_bytes(void) { return OPT_MAX_BLOCKS * sizeof(double); }

// norm_out[0] = grad_scale * ||grad||_2   (grad_scale folds the 1/world_size of the summed all-reduce)
int32_t vgpa_grad_norm(const float* grad, int64_t n, float grad_scale, float* norm_out, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (!grad || !norm_out || !workspace || n <= 0 || ((uintptr_t)grad & 15)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_grad_norm_workspace_bytes()) return VGPA_ERR_WORKSPACE;
    const int nb = opt_blocks(n);
    VGPA_LAUNCH(sumsq_partial_kernel, dim3(nb), dim3(OPT_THREADS), 0, stream, grad, n, (double*)workspace);
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(sumsq_finish_kernel, dim3(1), dim3(OPT_THREADS), 0, stream, (const double*)workspace, nb, grad_scale, norm_out);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

// One AdamW step (step >= 1) on flat fp32 buffers; gradient is used as grad * grad_scale * min(1, max_norm/(norm+1e-6)).
int32_t vgpa_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int64_t step, float grad_scale, float max_norm, const float* total_norm,
                        hipStream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return VGPA_ERR_INVALID;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return VGPA_ERR_INVALID;
    if (max_norm > 0.f && !total_norm) return VGPA_ERR_INVALID;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    VGPA_LAUNCH(adamw_kernel, dim3(opt_blocks(n)), dim3(OPT_THREADS), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, bc2_sqrt, grad_scale, max_norm, total_norm);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
