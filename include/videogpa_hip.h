/* videogpa_hip.h -- C ABI of libvgpa_hip.so: the MI355X (gfx950) kernels behind the VideoGPA DPO hot path.
 *
 * Boundary contract: plain pointers and sizes, no torch types; the CALLER owns every buffer
 * (outputs and workspaces); every call is asynchronous on `stream`, re-entrant, holds no global state; returns
 * 0 on success or a negative VGPA_ERR_* code and never throws.  One process per GPU.
 *
 * The reference (Hongyang-Du/VideoGPA) is pure Python and reaches these computations through PyTorch / diffusers /
 * PEFT objects; each entry point below cites the reference interface (file:line) it stands behind.  The Python
 * host side that mirrors those object APIs is `videogpa_amd/` (ctypes binding: videogpa_amd/_lib.py; the binding a
 * reference maintainer would add is shown in INTEGRATION.md).
 *
 * dtype codes: 0 = float32, 1 = bfloat16.  "BHS strides" = int64[3] element strides {batch, head, token} of a
 * [B, H, S, 64] bf16 view whose last dimension is contiguous (multiples of 8; base pointers 16-byte aligned).
 */
#ifndef VIDEOGPA_HIP_H
#define VIDEOGPA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vgpa_stream_t; /* == hipStream_t */

#define VGPA_OK 0
#define VGPA_ERR_INVALID (-1)
#define VGPA_ERR_LAUNCH (-2)
#define VGPA_ERR_WORKSPACE (-3)

/* ---- Diffusion-DPO loss: train/loss.py:53-121 (DPOLoss.forward) + its autograd backward -------------------------
 * Each input is [B, N] with its own per-sample stride (paired layout [B,2,N]: win = base, lose = base + N, stride 2N).
 * out5 = {loss, reward_margin, winner_reward, loser_reward, accuracy}; dlogit[B] = dloss/dlogit_b (kept for bwd);
 * errs[B,4] = {e_win, e_lose, e_win_ref, e_lose_ref} (may be NULL).  loss_type 0 = sigmoid (label_smoothing
 * honoured), 1 = hinge.  flags bit0: round (v - tgt) to bf16 before squaring (what bf16 autocast does upstream). */
size_t vgpa_dpo_loss_workspace_bytes(int64_t B);
int32_t vgpa_dpo_loss_fwd(const void* v_win, const void* v_lose, const void* v_win_ref, const void* v_lose_ref,
                          const void* tgt_win, const void* tgt_lose, int64_t B, int64_t N, int64_t stride_pred,
                          int64_t stride_ref, int64_t stride_tgt, int32_t dtype, float beta, float label_smoothing,
                          int32_t loss_type, int32_t flags, float* out5, float* dlogit, float* errs, void* workspace,
                          size_t ws_bytes, vgpa_stream_t stream);
int32_t vgpa_dpo_loss_bwd(const void* v_win, const void* v_lose, const void* tgt_win, const void* tgt_lose, int64_t B,
                          int64_t N, int64_t stride_pred, int64_t stride_tgt, int64_t stride_grad, int32_t dtype, float beta,
                          int32_t flags, const float* dlogit, const float* grad_out, void* grad_win, void* grad_lose,
                          vgpa_stream_t stream);

/* ---- noising + v-targets: scheduler.add_noise / get_velocity, train/CogVideoX-5B/03_train.py:129-130,154-155 ------
 * x_pair [B,2,N], noise [B,N] shared by the pair, t [B]; tables = sqrt(abar), sqrt(1-abar) (fp32 [T]). */
int32_t vgpa_noise_velocity_paired(const void* x_pair, const void* noise, const int64_t* t, const float* sqrt_abar,
                                   const float* sqrt_1m_abar, int64_t B, int64_t N, int32_t num_train_timesteps,
                                   int32_t dtype, void* x_noisy_pair, void* v_target_pair, vgpa_stream_t stream);

/* flow-matching variant (train/Wan2.2-TI2V-5B/03_train.py:103-116): x_t = (1-sigma) x + sigma eps (fp32 when xt_f32, as
 * torch's promotion gives), v = eps - x; sigma fp32 [B]. */
int32_t vgpa_flow_noise_velocity_paired(const void* x_pair, const void* noise, const float* sigma, int64_t B, int64_t N,
                                        int32_t dtype, int32_t xt_f32, void* x_noisy_pair, void* v_target_pair,
                                        vgpa_stream_t stream);

/* ---- AdaLN-Zero pieces of CogVideoXBlock (diffusers CogVideoXLayerNormZero / AdaLayerNorm / LayerNorm), reached
 * from train/CogVideoX-5B/03_train.py:134-151.  x, out, dy, dx: bf16 [B,S,D], text tokens first (rows < text_len).
 * Per-range fp32 vectors [B,D] with a common batch stride; all four modulation pointers NULL = plain LayerNorm. */
int32_t vgpa_ln_modulate_fwd(const void* x, const float* ln_w, const float* ln_b, const float* shift_v,
                             const float* scale1p_v, const float* shift_t, const float* scale1p_t, int64_t mod_stride,
                             int64_t B, int64_t S, int64_t D, int64_t text_len, float eps, void* out, float* mean,
                             float* rstd, vgpa_stream_t stream);
int32_t vgpa_ln_modulate_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* ln_w,
                             const float* scale1p_v, const float* scale1p_t, int64_t mod_stride, int64_t B, int64_t S,
                             int64_t D, int64_t text_len, const void* dres, void* dx, vgpa_stream_t stream);
/* fused gated residual + following LN-modulate (the pair diffusers runs as `hidden += gate * out` then `norm(hidden)`):
 *   fwd: x_new = x + gate[range]*y ; n = LN(x_new) * (w (1+scale)) + (b (1+scale) + shift)      (y NULL: n = LN-mod(x) only)
 *   bwd: dx = dres + LN'(dn) ; dy = gate[range]*dx                                               (dres / dy may be NULL) */
int32_t vgpa_residual_ln_fwd(const void* x, const void* y, const float* gate_v, const float* gate_t, int64_t gate_stride,
                             const float* ln_w, const float* ln_b, const float* shift_v, const float* scale1p_v,
                             const float* shift_t, const float* scale1p_t, int64_t mod_stride, int64_t B, int64_t S, int64_t D,
                             int64_t text_len, float eps, void* x_new, void* n_out,
                             int64_t n_stride /* row stride of n_out in elements, >= D: n may be the head of a wider [rows, D+R]
                                                 buffer whose tail carries LoRA down-projections */,
                             float* mean, float* rstd, vgpa_stream_t stream);
int32_t vgpa_residual_ln_bwd(const void* dn, const void* x_new, const float* mean, const float* rstd, const float* ln_w,
                             const float* scale1p_v, const float* scale1p_t, int64_t mod_stride, const float* gate_v,
                             const float* gate_t, int64_t gate_stride, const void* dres, int64_t B, int64_t S, int64_t D,
                             int64_t text_len, void* dx, void* dy, int64_t dy_stride /* row stride of dy, >= D */,
                             vgpa_stream_t stream);
/* out = x + gate[range] * y (x NULL: out = gate * y, the backward of the y branch) */
int32_t vgpa_gate_residual(const void* x, const void* y, const float* gate_v, const float* gate_t, int64_t mod_stride,
                           int64_t B, int64_t S, int64_t D, int64_t text_len, void* out, vgpa_stream_t stream);
/* FeedForward activation "gelu-approximate" (tanh), bf16, n elements */
int32_t vgpa_gelu_tanh_fwd(const void* u, int64_t n, void* out, vgpa_stream_t stream);
int32_t vgpa_gelu_tanh_bwd(const void* u, const void* dy, int64_t n, void* du, vgpa_stream_t stream);

/* ---- QK-norm (LayerNorm over head_dim 64) + 3D RoPE (attn.norm_q/norm_k, apply_rotary_emb in
 * CogVideoXAttnProcessor2_0; RoPE only when image_rotary_emb is given: generate/CogVideoX-5B.py:72-77).
 * rope_cos/rope_sin: fp32 [S - text_len, 64] or NULL.  q_out is additionally multiplied by q_out_scale (the attention
 * kernels take q pre-multiplied by scale*log2(e)); the backward returns d/d(q_in) for the UNscaled normalised query. */
int32_t vgpa_qknorm_rope_fwd(const void* q_in, const void* k_in, void* q_out, void* k_out, const int64_t* qin_strides,
                             const int64_t* kin_strides, const int64_t* qout_strides, const int64_t* kout_strides,
                             const float* wq, const float* bq, const float* wk, const float* bk, const float* rope_cos,
                             const float* rope_sin, int64_t text_len, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                             float eps, float q_out_scale, int32_t rope_mode, vgpa_stream_t stream);
int32_t vgpa_qknorm_rope_bwd(const void* dq_out, const void* dk_out, const void* q_in, const void* k_in, void* dq_in,
                             void* dk_in, const int64_t* dqout_strides, const int64_t* dkout_strides,
                             const int64_t* qin_strides, const int64_t* kin_strides, const int64_t* dqin_strides,
                             const int64_t* dkin_strides, const float* wq, const float* wk, const float* rope_cos,
                             const float* rope_sin, int64_t text_len, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                             float eps, int32_t rope_mode, vgpa_stream_t stream);

/* ---- 3D full attention, non-causal, head_dim 64 (F.scaled_dot_product_attention in the same processor) ----------
 * CONTRACT: q is PRE-MULTIPLIED by scale*log2(e) in every entry point below; dq is the gradient w.r.t. the unscaled q.
 * lse2 = log2 sum_k exp2(q.k), fp32 [B,H,S]; delta = rowsum(dO * O), fp32 [B,H,S]. */
int32_t vgpa_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                      const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H,
                      int64_t S, int64_t head_dim, float scale, vgpa_stream_t stream);
/* The same with a caller-owned workspace (>= vgpa_attn_fwd_workspace_bytes): when the number of (head, 256-row strip)
 * tasks leaves a mostly empty last scheduling round on the device, those leftover tasks are cut into key-range chunks
 * (second small launch + merge) instead.  split_mode: -1 automatic, 0 never, k >= 2 force k chunks for every task. */
size_t vgpa_attn_fwd_workspace_bytes(int64_t B, int64_t H, int64_t S);
int32_t vgpa_attn_fwd_ws(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                         const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H,
                         int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes,
                         vgpa_stream_t stream);
size_t vgpa_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t S);
int32_t vgpa_attn_bwd_delta(const void* o, const void* d_o, const int64_t* o_strides, const int64_t* do_strides, float* delta,
                            int64_t B, int64_t H, int64_t S, int64_t head_dim, vgpa_stream_t stream);
/* The same from the output as the forward's residual tensor completes it (o_res / res_kind as vgpa_attn_fwd_w1_res wrote them; o_res may be
 * NULL).  delta stands for rowsum(P o dP), which equals rowsum(dO o O) for the UNROUNDED O only; from the bf16 O alone (what flash-attention
 * backwards, torch's included, do) every row's dS stops summing to zero and dQ picks up a coherent error. */
int32_t vgpa_attn_bwd_delta_res(const void* o, const void* o_res, int32_t res_kind, const void* d_o, const int64_t* o_strides, const int64_t* ores_strides,
                                const int64_t* do_strides, float* delta, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                                vgpa_stream_t stream);
int32_t vgpa_attn_bwd_dkv(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                          void* dk, void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                          const int64_t* do_strides, const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H,
                          int64_t S, int64_t head_dim, float scale, vgpa_stream_t stream);
int32_t vgpa_attn_bwd_dq(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                         void* dq, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                         const int64_t* do_strides, const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                         float scale, vgpa_stream_t stream);
/* fused alternative to bwd_dkv + bwd_dq (5 instead of 7 matrix products per score block): dK, dV as above, dQ added with
 * fp32 atomics into dq_f32 = fp32 [B,H,S,64] contiguous, which the CALLER MUST ZERO first. */
/* dQ on the "w1" structure (attention_w1.hip: one wave per SIMD, 512-register waves, LDS-DMA ring, hand-scheduled main
 * loop); arguments and results as vgpa_attn_bwd_dq_ws (workspace >= vgpa_attn_bwd_split_workspace_bytes, may be NULL). */
int32_t vgpa_attn_bwd_dq_w1(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                            void* dq, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                            const int64_t* do_strides, const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                            float scale, int32_t split_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* Forward on the "w1" structure (attention_w1.hip); arguments and results as vgpa_attn_fwd_ws, the workspace
 * (>= vgpa_attn_fwd_w1_workspace_bytes) is required.  Scores are shifted per row by M' = min(b, m_s + 64), b = the bound |q| max|k|, m_s = the row's maximum
 * over 64 keys spread evenly over the sequence, instead of a running maximum; 256-row strips whose sum overflows or comes too close to underflow (a row whose true
 * maximum lies > ~176 log2 units above the sampled one) are redone by the online-softmax kernel in the same call.  lse2 is formed from the sum of the bf16-ROUNDED
 * weights (the ones the PV product multiplies: O is an exact convex combination of V rows); it differs from the exact value by a row's weighted mean rounding error
 * (<= 2^-8 relative, i.e. 5.6e-3 in log2 units, on a one-hot row; ~ 2e-3 / sqrt(n) on a row spread over n keys: tests/attn_tol.py). */
size_t vgpa_attn_fwd_w1_workspace_bytes(int64_t B, int64_t H, int64_t S);
int32_t vgpa_attn_fwd_w1(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides,
                         const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, int64_t B, int64_t H, int64_t S,
                         int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* vgpa_attn_fwd_w1 that also writes what the bf16 rounding of the output dropped ([B,H,S,64] view o_res with its own ELEMENT strides; NULL = plain
 * vgpa_attn_fwd_w1), for the backward's delta (vgpa_attn_bwd_prep_w1_res / vgpa_attn_bwd_delta_res).  res_kind:
 *   VGPA_RES_BF16 (1)  bf16 elements: O_fp32 - bf16(O);
 *   VGPA_RES_8    (2)  uint8 elements: eight further mantissa bits, 128 + clamp(rint((O_fp32 - bf16(O)) * 2^8 / ulp(bf16(O))), -128, 127).
 * Either way (o, o_res) carries the output to 2^-17 relative; the 8-bit form costs half the bytes (1 per output element). */
#define VGPA_RES_NONE 0
#define VGPA_RES_BF16 1
#define VGPA_RES_8 2
int32_t vgpa_attn_fwd_w1_res(const void* q, const void* k, const void* v, void* o, void* o_res, int32_t res_kind, float* lse2, const int64_t* q_strides,
                             const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* ores_strides,
                             int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace,
                             size_t ws_bytes, vgpa_stream_t stream);
/* Redo accounting of vgpa_attn_fwd_w1 / _w1_res: after the call the workspace holds, behind its first B*H uint32 words (max_k |k|^2 per head), one int32 per
 * (batch, head, 256-row strip): non-zero = the strip held a row outside what the shifted loop represents (see vgpa_attn_fwd_w1) and it was redone by the online-softmax kernel inside the same call.  Results never depend on it; time does.
 * vgpa_attn_fwd_online_res: same arguments, results and workspace, EVERY strip on the online-softmax kernel -- the faster call when most strips would be flagged
 * (one sweep instead of two).  The host side (transformer.AttentionCore) reads the count on a layer's first calls and switches that layer. */
int32_t vgpa_attn_fwd_online_res(const void* q, const void* k, const void* v, void* o, void* o_res, int32_t res_kind, float* lse2, const int64_t* q_strides,
                                 const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, const int64_t* ores_strides,
                                 int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace,
                                 size_t ws_bytes, vgpa_stream_t stream);
/* w1 backward, step 1 and step 2: vgpa_attn_bwd_prep_w1 writes delta (fp32 [B,H,S], as vgpa_attn_bwd_delta) and the statistics
 * planes stats = fp32 [B,H,2,S] = {-lse2, -delta}; vgpa_attn_bwd_dkv_w1 = vgpa_attn_bwd_dkv_ws on the w1 structure, reading `stats`. */
int32_t vgpa_attn_bwd_prep_w1(const void* o, const void* d_o, const float* lse2, const int64_t* o_strides, const int64_t* do_strides,
                              float* delta, float* stats, int64_t B, int64_t H, int64_t S, int64_t head_dim, vgpa_stream_t stream);
int32_t vgpa_attn_bwd_prep_w1_res(const void* o, const void* o_res, int32_t res_kind, const void* d_o, const float* lse2, const int64_t* o_strides,
                                  const int64_t* ores_strides, const int64_t* do_strides, float* delta, float* stats, int64_t B, int64_t H,
                                  int64_t S, int64_t head_dim, vgpa_stream_t stream);
int32_t vgpa_attn_bwd_dkv_w1(const void* q, const void* k, const void* v, const void* d_o, const float* stats, void* dk, void* dv,
                             const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, const int64_t* do_strides,
                             const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                             float scale, int32_t split_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* vgpa_attn_bwd_dkv / _dq with a caller-owned workspace (>= vgpa_attn_bwd_split_workspace_bytes, may be shared by the two
 * calls): the leftover tasks of a mostly empty last scheduling round are cut into chunks along the streamed axis (second
 * small launch + fp32 merge), as in vgpa_attn_fwd_ws.  split_mode: -1 automatic, 0 never, k >= 2 force k chunks. */
size_t vgpa_attn_bwd_split_workspace_bytes(int64_t B, int64_t H, int64_t S);
int32_t vgpa_attn_bwd_dkv_ws(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                             void* dk, void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                             const int64_t* do_strides, const int64_t* dk_strides, const int64_t* dv_strides, int64_t B, int64_t H,
                             int64_t S, int64_t head_dim, float scale, int32_t split_mode, void* workspace, size_t ws_bytes,
                             vgpa_stream_t stream);
int32_t vgpa_attn_bwd_dq_ws(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                            void* dq, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                            const int64_t* do_strides, const int64_t* dq_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim,
                            float scale, int32_t split_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
#ifdef VGPA_VARIANTS /* measured-slower experiment (one-kernel backward, dQ by fp32 atomics): exported by variant builds only
                      * (tools/build_variant.sh); dq_f32 = fp32 [B,H,S,64] contiguous, zeroed by the caller */
int32_t vgpa_attn_bwd_fused(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                            float* dq_f32, void* dk, void* dv, const int64_t* q_strides, const int64_t* k_strides,
                            const int64_t* v_strides, const int64_t* do_strides, const int64_t* dk_strides,
                            const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale,
                            vgpa_stream_t stream);
#endif
int32_t vgpa_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq,
                      void* dk, void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                      const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                      const int64_t* dv_strides, int64_t B, int64_t H, int64_t S, int64_t head_dim, float scale, void* workspace,
                      size_t ws_bytes, vgpa_stream_t stream);

/* ---- LoRA A.B contractions (peft Linear.forward `lora_B(lora_A(x)) * scaling` and its backward for the adapters of
 * train/CogVideoX-5B/03_train.py:102-106).  bf16 row-major operands with explicit row strides (elements).
 *   down  : T[M,R]  = X[M,K] A[R,K]^T                     K % 64 == 0, R <= 256
 *   up_add: Y[M,N]  = (accumulate ? Y : 0) + s * T[M,rp] Bw[N,rp]^T   rp in {16,32,48,64,96,128,192}, N % 8 == 0
 *   grad  : G[P,Q] += s * U[M,P]^T V[M,Q]                 G fp32, caller-zeroed (fp32 atomics) */
int32_t vgpa_lora_down(const void* X, int64_t ldx, const void* A, void* T, int64_t ldt, int64_t M, int64_t K, int64_t R,
                       vgpa_stream_t stream);
int32_t vgpa_lora_up_add(void* Y, int64_t ldy, const void* T, int64_t ldt, const void* Bw, int64_t ldb, float s, int64_t M,
                         int64_t N, int64_t rp, int32_t accumulate, vgpa_stream_t stream);
int32_t vgpa_lora_grad(const void* U, int64_t ldu, const void* V, int64_t ldv, float* G, int64_t ldg, float s, int64_t M,
                       int64_t P, int64_t Q, vgpa_stream_t stream);
/* the same product, bit-reproducible: per-row-range partials in the caller's workspace + an ordered merge; G is overwritten */
size_t vgpa_lora_grad_workspace_bytes(int64_t M, int64_t P, int64_t Q);
int32_t vgpa_lora_grad_ws(const void* U, int64_t ldu, const void* V, int64_t ldv, float* G, int64_t ldg, float s, int64_t M,
                          int64_t P, int64_t Q, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* adapter refresh of the K-extended projection operands after an optimizer step (PEFT re-reads lora_A / lora_B in every forward,
 * peft/tuners/lora/layer.py Linear.forward; here the extended GEMM operands cache them): a = bf16(A[r, K]), sB = bf16(float(bf16(B[Dn, r])) * s)
 * -> a_cat [r, K], wt_tail[k * ld_wt + rr] = a (the [K, N + R] operand's tail columns), w_tail[n * ld_w + rr] = sB (the [N, K + R] operand's
 * tail columns), sbt [rr * Dn + n] = sB.  A, B fp32 contiguous, outputs bf16. */
int32_t vgpa_lora_ext_refresh(const float* A, const float* B, float s, int64_t r, int64_t K, int64_t Dn, void* a_cat, void* wt_tail, int64_t ld_wt,
                              void* w_tail, int64_t ld_w, void* sbt, vgpa_stream_t stream);

/* ---- feed-forward GEMM with fused epilogue: diffusers FeedForward(activation_fn="gelu-approximate") inside CogVideoXBlock
 * (the reference reaches it through diffusers cogvideox_transformer_3d.py; train/CogVideoX-5B/utils.py:255-289 drives the blocks).
 * C[M, N] = epi(X[M, K] W[N, K]^T + bias[N]), bf16 in / fp32 accumulate / bf16 out; N % 128 == 0, K % 64 == 0, ld* in elements.
 *   epilogue 0: identity   1: gelu_tanh (aux != NULL: the bf16 pre-activation is also stored to aux)   2: C = acc * gelu_tanh'(aux) */
#ifdef VGPA_VARIANTS   /* probe: profiles/r03_gemm_probe.txt */
int32_t vgpa_gemm_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, const void* bias, void* C, int64_t ldc, void* aux,
                       int64_t ldaux, int32_t M, int32_t N, int32_t K, int32_t epilogue, vgpa_stream_t stream);
#endif

/* ---- head_dim-128 attention with separate query / key lengths: the self- and cross-attention of the Wan2.2-TI2V-5B denoiser
 * (train/Wan2.2-TI2V-5B/03_train.py:150-163; WanModel comes from the un-vendored Wan2.2 checkout: 24 heads x 128, text length 512).
 * q, o, d_o, dq: [B, H, Sq, 128] views; k, v, dk, dv: [B, H, Skv, 128] views; *_strides = element strides {batch, head, token},
 * last dim contiguous.  lse2 [B, H, Sq] fp32 = log2-domain log-sum-exp written by the forward. */
/* With a workspace (vgpa_attn128_fwd_workspace_bytes) and Skv >= 1024 the forward runs on the one-wave-per-SIMD / LDS-DMA structure
 * (row-bound softmax shift, flagged strips redone with a running max); workspace NULL: the compiler-scheduled kernel. */
size_t vgpa_attn128_fwd_workspace_bytes(int64_t B, int64_t H, int64_t Sq);
/* o_res8 (optional; uint8 [B, H, Sq, 128] view with its own element strides, NULL = not written): eight further mantissa bits of every output value
 * (VGPA_RES_8 of vgpa_attn_fwd_w1_res) for the backward's delta = rowsum(dO o O) -- pass the same tensor to vgpa_attn128_bwd. */
int32_t vgpa_attn128_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides, const int64_t* k_strides,
                         const int64_t* v_strides, const int64_t* o_strides, void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H,
                         int64_t Sq, int64_t Skv, float scale, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* The head_dim-128 forward on OCP-e4m3 matrix operands (BASELINE configs[4] "fp8 MFMA path"): q (with scale * log2 e folded in), k and v are quantised
 * with one power-of-two scale per (batch, head) and tensor, v transposed, by two prep kernels; both products run as v_mfma_scale_f32_32x32x64_f8f6f4
 * with the scales -- and a per-tile, per-row power of two for the softmax weights -- on the instruction's E8M0 operands; row sums and lse2 stay fp32.
 * Arguments and results as vgpa_attn128_fwd; the workspace (>= vgpa_attn128_fwd_f8_workspace_bytes, 256-byte aligned) is REQUIRED and is scratch.
 * The softmax shift of a row is M' = b - n, b = |q8 row| max|k8 row| and n = floor(max(0, b - (m_s + 64))) with m_s the row's maximum over 64 keys spread evenly
 * over the sweep (as vgpa_attn_fwd_w1; an INTEGER step off the bound, so the e4m3 bits of the weights do not depend on it); 256-row strips with a row it cannot
 * represent (row sum outside [2^-100, 2^118), M' > 1024, a non-finite accumulator) are redone by the bf16 running-max kernel inside the call.
 * Forward only.  q_deq / k_deq / v_deq (optional, all three or none; bf16 [B,H,S,128] views with their stride triples): the operands the products really ran
 * on, dequantised EXACTLY (an e4m3 value times a power of two is a bf16 number): k8 2^ek, v8 2^ev and q8 2^eq -- the query PRE-SCALED by scale * log2 e.
 * vgpa_attn128_bwd_prescaled run on THEM with this call's lse2, output and o_res8 is the straight-through gradient of this forward: its recomputed scores are
 * this call's bit for bit, its softmax weights this call's p / l (rows sum to one) and delta = rowsum(dO o O) matches them. */
size_t vgpa_attn128_fwd_f8_workspace_bytes(int64_t B, int64_t H, int64_t Sq, int64_t Skv);
int32_t vgpa_attn128_fwd_f8(const void* q, const void* k, const void* v, void* o, float* lse2, const int64_t* q_strides, const int64_t* k_strides,
                            const int64_t* v_strides, const int64_t* o_strides, void* o_res8, const int64_t* ores_strides, void* q_deq, void* k_deq,
                            void* v_deq, const int64_t* qd_strides, const int64_t* kd_strides, const int64_t* vd_strides, int64_t B, int64_t H,
                            int64_t Sq, int64_t Skv, float scale, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* workspace (vgpa_attn128_bwd_workspace_bytes): delta + the statistics planes.  dkv_mode -1 = automatic (w1 dK/dV kernel from 1024 queries on),
 * 0 = compiler-scheduled kernel, 1 = w1 kernel */
size_t vgpa_attn128_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Sq);
int32_t vgpa_attn128_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                         void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                         const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                         const int64_t* dv_strides, const void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H, int64_t Sq, int64_t Skv,
                         float scale, int32_t dkv_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* vgpa_attn128_bwd for a query that arrives PRE-SCALED by scale * log2 e (q = vgpa_attn128_fwd_f8's q_deq; k = k_deq, v = v_deq): scores are q.k as they stand
 * (log2 units); dq = scale dS k is the gradient w.r.t. the UNscaled query, dk = ln 2 dS^T q (= scale dS^T q_unscaled).  Everything else as vgpa_attn128_bwd. */
int32_t vgpa_attn128_bwd_prescaled(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse2, void* dq, void* dk,
                                   void* dv, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                   const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                                   const int64_t* dv_strides, const void* o_res8, const int64_t* ores_strides, int64_t B, int64_t H, int64_t Sq, int64_t Skv,
                                   float scale, int32_t dkv_mode, void* workspace, size_t ws_bytes, vgpa_stream_t stream);

/* ---- row kernels of the Wan2.2 denoiser block (WanAttentionBlock / WanRMSNorm / rope_apply of the Wan2.2 checkout imported at
 * train/Wan2.2-TI2V-5B/03_train.py:43-48).  fp32 residual stream, bf16 matmul operands.  Per-token modulation as a table: row
 * gid[row] of a [groups, mod_stride] fp32 table (shift / scale / gate point at their chunk's column offset); gid NULL = row 0.
 *   ln_mod_fwd : out(bf16, row stride out_ld >= D: the LoRA tail of the consuming projection may follow each row) = ([round_bf16] LN_eps(x)) * ln_w
 *                + ln_b, then * (1 + scale[g]) + shift[g]   (affine / modulation optional; x_dtype VGPA_DTYPE_F32 | VGPA_DTYPE_BF16); with q8 / q8_scale the
 *                same rows are ALSO (or, out NULL, only) written as the e4m3 operand of the fp8 feed-forward GEMM: q8 [rows, D] bytes, q8_scale [rows]
 *                fp32, bit-identical to vgpa_quant_fp8_rows of the bf16 output
 *   ln_mod_bwd : dx(fp32) = [dres +] LN-backward(dy (1 + scale) ln_w)
 *   gate_residual: out(fp32) = [x +] y(bf16) * gate[g]  (gate NULL = 1; out may alias x)
 *   gate_bwd   : dy(bf16, row stride ld_dy) = dout(fp32) * gate[g];   gate_bwd_q8: the same rows as e4m3 + per-row scale (fp8 dX GEMM operand)
 *   rms_rope   : n = bf16(u rsqrt(mean(u^2) + eps)) over the whole row [heads * head_dim]; y = bf16(n w); interleaved pairs of every head
 *                rotated by the angle at rope_{cos,sin}[(row % L) * head_dim/2 + pair]  (NULL: no rotation); u / out / dout / du are row-strided
 *                (ld_* in elements) so that q and k are read from and their gradients written into the fused [rows, 3 D (+ LoRA tail)] buffers */
int32_t vgpa_wan_ln_mod_fwd(const void* x, int32_t x_dtype, const int32_t* gid, const float* ln_w, const float* ln_b, const float* shift,
                            const float* scale, int64_t mod_stride, int64_t rows, int64_t D, float eps, int32_t round_xhat, void* out, int64_t out_ld,
                            void* q8, float* q8_scale, float* mean, float* rstd, vgpa_stream_t stream);
int32_t vgpa_wan_ln_mod_bwd(const void* dy, const void* x, int32_t x_dtype, const float* mean, const float* rstd, const int32_t* gid, const float* ln_w,
                            const float* scale, int64_t mod_stride, int64_t rows, int64_t D, const float* dres, float* dx, vgpa_stream_t stream);
/* the same two row kernels with an fp32 result / fp32 incoming gradient: WanModel's output head (upstream Head.forward runs LN, modulation and
 * the projection in fp32; reference call site train/Wan2.2-TI2V-5B/03_train.py:227-233 through WanModel.forward) */
int32_t vgpa_wan_ln_mod_fwd_f32(const float* x, const int32_t* gid, const float* shift, const float* scale, int64_t mod_stride, int64_t rows, int64_t D, float eps,
                                float* out, float* mean, float* rstd, vgpa_stream_t stream);
int32_t vgpa_wan_ln_mod_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const int32_t* gid, const float* scale, int64_t mod_stride,
                                int64_t rows, int64_t D, float* dx, vgpa_stream_t stream);
/* a block's [gated residual add -> LayerNorm] pair in one pass (upstream WanAttentionBlock.forward: `x = x + y * e[2]` followed by norm3 / norm2 of that x):
 *   gate_ln_mod_fwd : xo(fp32) = x + y(bf16) * gate[g] (gate NULL = 1); then exactly ln_mod_fwd of xo (bf16 and / or e4m3 result)
 *   ln_mod_bwd_gate : dx = [dres +] LN-backward(...) as ln_mod_bwd, and dy_prev(bf16, row stride ld_dy_prev) = dx * gate_prev[g] (NULL = 1): the gradient of
 *                     the y of that residual add.  gate / gate_prev index the same table rows as shift / scale (mod_stride). */
int32_t vgpa_wan_gate_ln_mod_fwd(const float* x, const void* y, const int32_t* gid, const float* gate, const float* ln_w, const float* ln_b, const float* shift,
                                 const float* scale, int64_t mod_stride, int64_t rows, int64_t D, float eps, float* xo, void* out, int64_t out_ld, void* q8,
                                 float* q8_scale, float* mean, float* rstd, vgpa_stream_t stream);
int32_t vgpa_wan_ln_mod_bwd_gate(const void* dy, const float* x, const float* mean, const float* rstd, const int32_t* gid, const float* ln_w, const float* scale,
                                 int64_t mod_stride, int64_t rows, int64_t D, const float* dres, float* dx, const float* gate_prev, void* dy_prev,
                                 int64_t ld_dy_prev, vgpa_stream_t stream);
int32_t vgpa_wan_gate_residual(const float* x, const void* y, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, float* out,
                               vgpa_stream_t stream);
int32_t vgpa_wan_gate_bwd(const float* dout, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, void* dy, int64_t ld_dy,
                          vgpa_stream_t stream);
int32_t vgpa_wan_gate_bwd_q8(const float* dout, const int32_t* gid, const float* gate, int64_t mod_stride, int64_t rows, int64_t D, void* q8,
                             float* q8_scale, vgpa_stream_t stream);
int32_t vgpa_wan_rms_rope_fwd(const void* u, int64_t ld_u, const void* w, const float* rope_cos, const float* rope_sin, int64_t L, int64_t head_dim,
                              int64_t rows, int64_t D, float eps, void* out, int64_t ld_out, float* rstd, vgpa_stream_t stream);
int32_t vgpa_wan_rms_rope_bwd(const void* dout, int64_t ld_dout, const void* u, int64_t ld_u, const float* rstd, const void* w, const float* rope_cos,
                              const float* rope_sin, int64_t L, int64_t head_dim, int64_t rows, int64_t D, void* du, int64_t ld_du, vgpa_stream_t stream);

/* ---- fp8 operand preparation for the frozen feed-forward GEMMs of the Wan2.2 path (BASELINE.json configs[4]: "fp8 MFMA path"):
 * per-row dynamic quantisation of bf16 rows to OCP e4m3: scale[m] = amax(row) / 448 (1 for a zero row), q = e4m3(x / scale), RNE, saturating.
 * x [M, K] bf16 with row stride ldx (elements); q [M, K] bytes, contiguous; scale [M] fp32. */
int32_t vgpa_quant_fp8_rows(const void* x, int64_t ldx, void* q, float* scale, int64_t M, int64_t K, vgpa_stream_t stream);
/* the feed-forward's activation written as that operand by its producer: q8 = e4m3(bf16(gelu_tanh(u)) / scale) resp. of bf16(dy * gelu_tanh'(u));
 * u / dy [rows, K] bf16 contiguous, K <= 16384; bit-identical to vgpa_gelu_tanh_{fwd,bwd} followed by vgpa_quant_fp8_rows */
int32_t vgpa_gelu_tanh_fwd_q8(const void* u, int64_t rows, int64_t K, void* q8, float* q8_scale, vgpa_stream_t stream);
int32_t vgpa_gelu_tanh_bwd_q8(const void* u, const void* dy, int64_t rows, int64_t K, void* q8, float* q8_scale, vgpa_stream_t stream);

/* ---- VGGT input preprocessing: utils/model_utils.py:16-85 preprocess_images_from_numpy (PIL bicubic resize to width 518 /
 * longer side 518, ToTensor, centre crop or white pad).  frames uint8 [T, H, W, 3] -> out float32 [T, 3, out_h, out_w].
 * mode 0 = "crop", 1 = "pad".  vgpa_preprocess_shape is host arithmetic only (:36-48, :54-71). */
int32_t vgpa_preprocess_shape(int32_t H, int32_t W, int32_t mode, int32_t* out_h, int32_t* out_w);
size_t vgpa_preprocess_workspace_bytes(int32_t T, int32_t H, int32_t W, int32_t mode);
int32_t vgpa_preprocess_frames(const void* frames, int32_t T, int32_t H, int32_t W, int32_t mode, float* out, void* workspace,
                               size_t ws_bytes, vgpa_stream_t stream);

/* ---- optimizer on one flat fp32 buffer of all LoRA parameters: gradient_clip_val=1.0 + torch.optim.AdamW,
 * train/CogVideoX-5B/03_train.py:208-213,266.  norm_out[0] = grad_scale * ||grad||_2. */
size_t vgpa_grad_norm_workspace_bytes(void);
int32_t vgpa_grad_norm(const float* grad, int64_t n, float grad_scale, float* norm_out, void* workspace, size_t ws_bytes,
                       vgpa_stream_t stream);
int32_t vgpa_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step, float grad_scale, float max_norm,
                        const float* total_norm, vgpa_stream_t stream);

/* ---- geometry-consistency scorer ---------------------------------------------------------------------------------
 * project_points: utils/projection_utils.py:12-51 (+ :57-101 batch loop and [-1,1] output) with the confidence filter
 * of utils/pointcloud_utils.py:47-73 as a per-point predicate.  pc/colors fp32 [N,3], conf fp32 [N] or NULL,
 * K fp32 [T,3,3], E fp32 [T,e_rows,4] (e_rows 3 or 4).  canvas u8 [T,H,W,3] and/or out_f fp32 [T,3,H,W]. */
size_t vgpa_project_points_workspace_bytes(int64_t T, int64_t H, int64_t W);
int32_t vgpa_project_points(const float* pc, const float* colors, const float* conf, float conf_thr,
                            const float* conf_thr_dev /* device fp32[1] or NULL: overrides conf_thr */, const float* K,
                            const float* E, int32_t e_rows, int64_t N, int64_t T, int64_t H, int64_t W, uint8_t* canvas,
                            float* out_f, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* MSEMetric.compute, metrics/mse.py:14-54.  dtype 0 f32 / 2 u8; layout 0 [T,C,H,W] / 1 [T,H,W,C]; is_tensor selects
 * the torch.Tensor vs numpy range heuristics. */
size_t vgpa_frame_mse_workspace_bytes(void);
int32_t vgpa_frame_mse(const void* gt, int32_t gt_dtype, int32_t gt_layout, int32_t gt_is_tensor, const void* rep,
                       int32_t rep_dtype, int32_t rep_layout, int32_t rep_is_tensor, int64_t T, int64_t C, int64_t H,
                       int64_t W, float* out, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* compute_motion_score_vectorized, metrics/consistency_score.py:8-40 */
int32_t vgpa_motion_score(const float* E, int32_t e_rows, int64_t T, float* out, vgpa_stream_t stream);
/* kornia find_fundamental (8-point) + sampson_epipolar_distance as used by metrics/epipolar.py:197-213 */
int32_t vgpa_epipolar_sampson(const float* p1, const float* p2, const int64_t* offsets, int64_t n_pairs, float* err_out,
                              float* F_out, vgpa_stream_t stream);

/* Confidence cut of get_colored_pointcloud, utils/pointcloud_utils.py:44-73: thr_out[0] (device) = k-th largest valid
 * confidence, k = max(1, ceil(n_valid * (1 - conf_thres / 100))); -inf for conf_thres <= 0 or no valid point.  On-device
 * radix select (replaces torch.topk + .item()). */
size_t vgpa_conf_threshold_workspace_bytes(void);
int32_t vgpa_conf_threshold(const float* conf, int64_t N, float conf_thres, float* thr_out, void* workspace, size_t ws_bytes,
                            vgpa_stream_t stream);
/* MSEMetric / PSNRMetric.compute incl. the bilinear-resize branch, metrics/mse.py:14-29,56-80: rep [T,C,H2,W2] is resized to
 * gt's [H,W] (F.interpolate bilinear, align_corners=False).  psnr 0: mse; 1: 10 log10(1/mse), 100 when mse == 0. */
size_t vgpa_frame_metric_workspace_bytes(void);
int32_t vgpa_frame_metric(const void* gt, int32_t gt_dtype, int32_t gt_layout, int32_t gt_is_tensor, const void* rep,
                          int32_t rep_dtype, int32_t rep_layout, int32_t rep_is_tensor, int64_t T, int64_t C, int64_t H,
                          int64_t W, int64_t H2, int64_t W2, int32_t psnr, float* out, void* workspace, size_t ws_bytes,
                          vgpa_stream_t stream);
/* Input side of LPIPSMetric.compute, metrics/lpips.py:21-63 (the LPIPS network itself is the caller's): frames -> fp32
 * [T,C,Ho,Wo] in [-1,1] by the reference's range rules, bilinearly resized when (Ho,Wo) != (H,W). */
int32_t vgpa_frames_to_pm1(const void* src, int32_t dtype, int32_t layout, int32_t is_tensor, int64_t T, int64_t C, int64_t H,
                           int64_t W, int64_t Ho, int64_t Wo, float* out, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* MVCSMetric.compute, metrics/mvcs.py:12-114.  depth fp32 [T,H,W]; K fp32 [T,k_dim,k_dim] (k_dim 3|4); E fp32
 * [T,e_rows(3|4),4] world-to-camera; out[0] = exp(-mean over valid consecutive pairs of the masked depth MSE). */
size_t vgpa_mvcs_workspace_bytes(int64_t T);
int32_t vgpa_mvcs(const float* depth, const float* K, int32_t k_dim, const float* E, int32_t e_rows, int64_t T, int64_t H,
                  int64_t W, float* out, void* workspace, size_t ws_bytes, vgpa_stream_t stream);
/* DA3 world points, pipelines/process_video.py:151-156 (affine_inverse + unproject_depth,
 * depth_anything_3/utils/geometry.py:54-59,434-497).  depth fp32 [T,H,W], K fp32 [T,3,3], E fp32 [T,e_rows,4] -> world
 * fp32 [T,H,W,3]. */
int32_t vgpa_unproject_depth(const float* depth, const float* K, const float* E, int32_t e_rows, int64_t T, int64_t H,
                             int64_t W, float* world, vgpa_stream_t stream);
/* pose_encoding_to_extri_intri("absT_quaR_FoV"), vggt/utils/pose_enc.py:62-124: pe fp32 [n,9] -> ext fp32 [n,3,4],
 * intr fp32 [n,3,3] or NULL. */
int32_t vgpa_pose_decode(const float* pose_enc, int64_t n, float image_h, float image_w, float* ext, float* intr,
                         vgpa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VIDEOGPA_HIP_H */
