// Diffusion-DPO loss, fused forward reduction + backward (HBM-bound).
// Replaces train/loss.py:53-121 (DPOLoss.forward) and its autograd backward.
//   e_x[b]   = mean_n (v_x[b,n] - tgt_x[b,n])^2          for x in {win, lose, win_ref, lose_ref}
//   logit[b] = beta * ((e_wref - e_w) - (e_lref - e_l))
// Pass 1 streams the six tensors once (16-byte loads), fp32 per-thread partials, fp64 block/grid
// reduction in a fixed order (deterministic).  Pass 2 (one block) finishes the scalars.
#include "common.h"

#define LOSS_THREADS 256
#define LOSS_MAX_BLOCKS 256

template <int DT, bool ROUND_DIFF>
__global__ __launch_bounds__(LOSS_THREADS) void dpo_err_partial_kernel(
    const void* __restrict__ vw, const void* __restrict__ vl, const void* __restrict__ vwr,
    const void* __restrict__ vlr, const void* __restrict__ tw, const void* __restrict__ tl,
    int64_t N, int64_t s_pred, int64_t s_ref, int64_t s_tgt, int vec_ok, double* __restrict__ partial) {
    __shared__ double smem[16];
    const int b = blockIdx.y;
    const size_t op = (size_t)b * s_pred, orf = (size_t)b * s_ref, ot = (size_t)b * s_tgt;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t n8 = vec_ok ? (N >> 3) : 0;
    for (int64_t c = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; c < n8; c += (int64_t)gridDim.x * LOSS_THREADS) {
        float a[8], r[8], t[8];
        const size_t i = (size_t)c << 3;
        load8<DT>(tw, ot + i, t);
        load8<DT>(vw, op + i, a);
        load8<DT>(vwr, orf + i, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d0 = a[j] - t[j], d1 = r[j] - t[j];
            if (ROUND_DIFF) { d0 = round_bf16(d0); d1 = round_bf16(d1); }
            acc[0] += d0 * d0; acc[2] += d1 * d1;
        }
        load8<DT>(tl, ot + i, t);
        load8<DT>(vl, op + i, a);
        load8<DT>(vlr, orf + i, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d0 = a[j] - t[j], d1 = r[j] - t[j];
            if (ROUND_DIFF) { d0 = round_bf16(d0); d1 = round_bf16(d1); }
            acc[1] += d0 * d0; acc[3] += d1 * d1;
        }
    }
    // scalar tail (or everything when the vector path is not usable)
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * LOSS_THREADS) {
        float t0 = load1<DT>(tw, ot + i), t1 = load1<DT>(tl, ot + i);
        float d[4] = {load1<DT>(vw, op + i) - t0, load1<DT>(vl, op + i) - t1, load1<DT>(vwr, orf + i) - t0, load1<DT>(vlr, orf + i) - t1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ROUND_DIFF) d[k] = round_bf16(d[k]);
            acc[k] += d[k] * d[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double s = block_sum<double>((double)acc[k], smem);
        if (threadIdx.x == 0) partial[((size_t)b * gridDim.x + blockIdx.x) * 4 + k] = s;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void dpo_finish_kernel(const double* __restrict__ partial, int nblk, int B, double invN,
                                                                    float beta, float label_smoothing, int loss_type,
                                                                    float* __restrict__ out5, float* __restrict__ dlogit,
                                                                    float* __restrict__ errs) {
    __shared__ double smem[16];
    double loss = 0, margin = 0, wr = 0, lr = 0, acc = 0;
    for (int b = 0; b < B; ++b) {
        double e[4];
        for (int k = 0; k < 4; ++k) {
            double s = 0;
            for (int i = threadIdx.x; i < nblk; i += LOSS_THREADS) s += partial[((size_t)b * nblk + i) * 4 + k];
            e[k] = block_sum<double>(s, smem) * invN;
        }
        if (threadIdx.x == 0) {
            // the reference holds e_* as fp32; round here so logits see the same quantisation
            float ew = (float)e[0], el = (float)e[1], erw = (float)e[2], erl = (float)e[3];
            if (errs) { errs[b * 4 + 0] = ew; errs[b * 4 + 1] = el; errs[b * 4 + 2] = erw; errs[b * 4 + 3] = erl; }
            float x = beta * ((erw - ew) - (erl - el));
            double xd = x, li, dl;
            if (loss_type == 0) {
                double sp = fmax(-xd, 0.0) + log1p(exp(-fabs(xd)));  // softplus(-x) = -logsigmoid(x)
                double sig = 1.0 / (1.0 + exp(-xd));
                double y = 1.0 - (double)label_smoothing;
                li = (1.0 - y) * xd + sp;
                dl = sig - y;
            } else {
                li = fmax(1.0 - xd, 0.0);
                dl = (xd < 1.0) ? -1.0 : 0.0;
            }
            loss += li;
            dlogit[b] = (float)(dl / B);
            wr += -(double)ew; lr += -(double)el; margin += (double)(el - ew);
            acc += (ew < el) ? 1.0 : 0.0;
        }
    }
    if (threadIdx.x == 0) {
        out5[0] = (float)(loss / B);
        out5[1] = (float)(margin / B);
        out5[2] = (float)(wr / B);
        out5[3] = (float)(lr / B);
        out5[4] = (float)(acc / B);
    }
}

template <int DT, bool ROUND_DIFF>
__global__ __launch_bounds__(LOSS_THREADS) void dpo_bwd_kernel(const void* __restrict__ vw, const void* __restrict__ vl,
                                                                 const void* __restrict__ tw, const void* __restrict__ tl,
                                                                 int64_t N, int64_t s_pred, int64_t s_tgt, int64_t s_grad, int vec_ok,
                                                                 const float* __restrict__ dlogit, const float* __restrict__ grad_out,
                                                                 float beta, float two_over_n, void* __restrict__ gw, void* __restrict__ gl) {
    const int b = blockIdx.y;
    const float g = grad_out ? grad_out[0] : 1.0f;
    // dlogit/de_w = -beta, dlogit/de_l = +beta; de/dv = 2 (v - t) / N
    const float cw = g * dlogit[b] * (-beta) * two_over_n;
    const float cl = g * dlogit[b] * (beta) * two_over_n;
    const size_t op = (size_t)b * s_pred, ot = (size_t)b * s_tgt, og = (size_t)b * s_grad;
    const int64_t n8 = vec_ok ? (N >> 3) : 0;
    for (int64_t c = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; c < n8; c += (int64_t)gridDim.x * LOSS_THREADS) {
        float a[8], t[8], o[8];
        const size_t i = (size_t)c << 3;
        load8<DT>(vw, op + i, a);
        load8<DT>(tw, ot + i, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = a[j] - t[j]; if (ROUND_DIFF) d = round_bf16(d); o[j] = cw * d; }
        store8<DT>(gw, og + i, o);
        load8<DT>(vl, op + i, a);
        load8<DT>(tl, ot + i, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = a[j] - t[j]; if (ROUND_DIFF) d = round_bf16(d); o[j] = cl * d; }
        store8<DT>(gl, og + i, o);
    }
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * LOSS_THREADS) {
        float d0 = load1<DT>(vw, op + i) - load1<DT>(tw, ot + i);
        float d1 = load1<DT>(vl, op + i) - load1<DT>(tl, ot + i);
        if (ROUND_DIFF) { d0 = round_bf16(d0); d1 = round_bf16(d1); }
        store1<DT>(gw, og + i, cw * d0);
        store1<DT>(gl, og + i, cl * d1);
    }
}

static inline int loss_blocks(int64_t N) {
    int64_t nb = (N + (int64_t)LOSS_THREADS * 8 - 1) / ((int64_t)LOSS_THREADS * 8);
    if (nb < 1) nb = 1;
    if (nb > LOSS_MAX_BLOCKS) nb = LOSS_MAX_BLOCKS;
    return (int)nb;
}

static inline int aligned_ok(const void* p, int dtype) {
    return ((uintptr_t)p % (dtype == VGPA_DTYPE_BF16 ? 16 : 32)) == 0;
}

extern "C" {

size_t vgpa_dpo_loss_workspace_bytes(int64_t B) { return (size_t)B * LOSS_MAX_BLOCKS * 4 * sizeof(double); }

int32_t vgpa_dpo_loss_fwd(const void* v_win, const void* v_lose, const void* v_win_ref, const void* v_lose_ref,
                          const void* tgt_win, const void* tgt_lose, int64_t B, int64_t N, int64_t stride_pred,
                          int64_t stride_ref, int64_t stride_tgt, int32_t dtype, float beta, float label_smoothing,
                          int32_t loss_type, int32_t flags, float* out5, float* dlogit, float* errs, void* workspace,
                          size_t ws_bytes, hipStream_t stream) {
    if (!v_win || !v_lose || !v_win_ref || !v_lose_ref || !tgt_win || !tgt_lose || !out5 || !dlogit || !workspace) return VGPA_ERR_INVALID;
    if (B <= 0 || N <= 0 || B > 65535 || (dtype != VGPA_DTYPE_F32 && dtype != VGPA_DTYPE_BF16) || (loss_type != 0 && loss_type != 1)) return VGPA_ERR_INVALID;
    if (ws_bytes < vgpa_dpo_loss_workspace_bytes(B)) return VGPA_ERR_WORKSPACE;
    const int nblk = loss_blocks(N);
    const int vec_ok = aligned_ok(v_win, dtype) && aligned_ok(v_lose, dtype) && aligned_ok(v_win_ref, dtype) && aligned_ok(v_lose_ref, dtype) &&
                       aligned_ok(tgt_win, dtype) && aligned_ok(tgt_lose, dtype) && (stride_pred % 8 == 0) && (stride_ref % 8 == 0) && (stride_tgt % 8 == 0);
    dim3 grid(nblk, (unsigned)B);
    double* partial = (double*)workspace;
    const bool rd = (flags & 1) != 0;
#define LAUNCH(DT, RD) VGPA_LAUNCH((dpo_err_partial_kernel<DT, RD>), grid, dim3(LOSS_THREADS), 0, stream, v_win, v_lose, v_win_ref, \
                                          v_lose_ref, tgt_win, tgt_lose, N, stride_pred, stride_ref, stride_tgt, vec_ok, partial)
    if (dtype == VGPA_DTYPE_BF16) { if (rd) LAUNCH(VGPA_DTYPE_BF16, true); else LAUNCH(VGPA_DTYPE_BF16, false); }
    else { if (rd) LAUNCH(VGPA_DTYPE_F32, true); else LAUNCH(VGPA_DTYPE_F32, false); }
#undef LAUNCH
    VGPA_CHECK_LAUNCH();
    VGPA_LAUNCH(dpo_finish_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, partial, nblk, (int)B, 1.0 / (double)N, beta,
                       label_smoothing, loss_type, out5, dlogit, errs);
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

int32_t vgpa_dpo_loss_bwd(const void* v_win, const void* v_lose, const void* tgt_win, const void* tgt_lose, int64_t B, int64_t N,
                          int64_t stride_pred, int64_t stride_tgt, int64_t stride_grad, int32_t dtype, float beta, int32_t flags,
                          const float* dlogit, const float* grad_out, void* grad_win, void* grad_lose, hipStream_t stream) {
    if (!v_win || !v_lose || !tgt_win || !tgt_lose || !dlogit || !grad_win || !grad_lose) return VGPA_ERR_INVALID;
    if (B <= 0 || N <= 0 || B > 65535 || (dtype != VGPA_DTYPE_F32 && dtype != VGPA_DTYPE_BF16)) return VGPA_ERR_INVALID;
    const int vec_ok = aligned_ok(v_win, dtype) && aligned_ok(v_lose, dtype) && aligned_ok(tgt_win, dtype) && aligned_ok(tgt_lose, dtype) &&
                       aligned_ok(grad_win, dtype) && aligned_ok(grad_lose, dtype) && (stride_pred % 8 == 0) && (stride_tgt % 8 == 0) && (stride_grad % 8 == 0);
    dim3 grid(loss_blocks(N), (unsigned)B);
    const float two_over_n = (float)(2.0 / (double)N);
    const bool rd = (flags & 1) != 0;
#define LAUNCH(DT, RD) VGPA_LAUNCH((dpo_bwd_kernel<DT, RD>), grid, dim3(LOSS_THREADS), 0, stream, v_win, v_lose, tgt_win, tgt_lose, N, \
                                          stride_pred, stride_tgt, stride_grad, vec_ok, dlogit, grad_out, beta, two_over_n, grad_win, grad_lose)
    if (dtype == VGPA_DTYPE_BF16) { if (rd) LAUNCH(VGPA_DTYPE_BF16, true); else LAUNCH(VGPA_DTYPE_BF16, false); }
    else { if (rd) LAUNCH(VGPA_DTYPE_F32, true); else LAUNCH(VGPA_DTYPE_F32, false); }
#undef LAUNCH
    VGPA_CHECK_LAUNCH();
    return VGPA_OK;
}

}  // extern "C"
