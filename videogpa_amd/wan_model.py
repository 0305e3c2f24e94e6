"""`WanModel` -- the Wan2.2 denoiser the reference trains (train/Wan2.2-TI2V-5B/03_train.py:43-48 imports `wan.modules.model.WanModel`
from a sibling Wan2.2 checkout; :139-168 loads it, wraps q/k/v/o with LoRA, checkpoints each block) -- on the MI355X kernels.

The Wan2.2 source is NOT in the reference tree (un-vendored), so this module restates its published architecture; module and
parameter names follow the upstream state dict (patch_embedding, text_embedding.{0,2}, time_embedding.{0,2}, time_projection.1,
blocks.N.{norm3, self_attn.{q,k,v,o,norm_q,norm_k}, cross_attn.{...}, ffn.{0,2}, modulation}, head.{head, modulation}) so that an
upstream checkpoint loads with `load_state_dict` and PEFT's target names 'q', 'k', 'v', 'o' hit the same linears.  Parity is against
oracle/wan.py (a plain torch restatement), which is UNPINNED for the same reason.

MI355X-first choices (everything else is the upstream arithmetic):
  * per-token modulation ([B, L, 6, C] fp32 upstream, 1.4 GB per sample at 18480 tokens) is a table over the DISTINCT timesteps of
    the batch (2 per sample for TI2V) + an int32 group id per token; the time MLP runs on the distinct values only;
  * fp32 residual stream, bf16 GEMM operands, fused LN + modulation, RMS-norm + weight + RoPE, gate + residual row kernels
    (csrc/wan.hip); attention is csrc/attention_hd128.hip (self: Sq = Skv = L; cross: Skv = text_len);
  * q / k / v of the self-attention are ONE projection [3 dim, dim] with the (up to three) LoRA adapters riding the GEMM as extra K
    (ops.LoraExt): the LN kernel writes its rows with the adapters' tail columns behind them, q / k are normalised out of the fused output
    by stride, v is read in place, and the backward assembles d(qkv) in one buffer (_SelfAttnFn); the o projections and the cross-attention's
    q get their tails from the attention kernel / the gate's backward the same way (no operand copies anywhere);
  * the residual path's gradient of each branch is added inside the LN backward kernel (ln_mod passthrough, _FfnFp8Fn);
  * enable_fp8(): the frozen feed-forward runs on e4m3 operands written by their producers (csrc/fp8.hip), one autograd node per branch;
  * no block recompute by default: 288 GB hold the activations of the full-size pair step (enable_gradient_checkpointing(True, stride=k)
    restores the reference's behaviour, every k-th block).
Call convention as upstream:  model(list of [C,F,H,W], t=[B] or [B, seq_len], context=list of [n, text_dim], seq_len=int) -> list of
[C_out,F,H,W] fp32.  All samples of a call must share one latent shape (the reference's batches do), and seq_len must equal the token
count (it does: 03_train.py:176-179 computes it from the latent)."""
import json
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .transformer import _adapter, _f32, _parts

_F32, _BF16 = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr_dtype(x):
    if x.dtype == torch.float32:
        return _F32
    if x.dtype == torch.bfloat16:
        return _BF16
    raise TypeError(f"videogpa_amd.wan_model: rows must be fp32 or bf16, got {x.dtype}")


def _modulation_is_frozen(tab, who):
    """ONE policy for every consumer of a modulation table (block LN / gate kernels, feed-forward node, output head): the row kernels return no gradient
    for shift / scale / gate, so a table that would need one -- `modulation`, the time embedding or the time projection left trainable -- raises instead of
    training with a silently missing gradient.  The DPO path freezes the base model (LoRA targets are q / k / v / o: train/Wan2.2-TI2V-5B/03_train.py:143-148,
    get_peft_model does the freezing); a bare WanModel needs `.requires_grad_(False)` before a grad-enabled forward."""
    if tab.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError(f"videogpa_amd.wan_model.{who}: the modulation table requires a gradient, which the HIP row kernels do not produce -- "
                                  "freeze the base model (model.requires_grad_(False), or wrap it with get_peft_model) before a grad-enabled forward")


class _LnModFn(torch.autograd.Function):
    """bf16( LN_eps(x) [rounded to bf16] * ln_w + ln_b, then * (1 + scale[gid]) + shift[gid] );   x [rows, D] fp32 or bf16.
    pad > 0: the result is the head of a [rows, D + pad] buffer (ops._padded_empty) whose tail the consuming projection fills with its LoRA
    down-projection, so the K-extended GEMM (ops.LoraExt) reads its operand in place.
    passthrough: x is returned as a second output for the residual path of the same branch (x' = x + gate * f(LN(x))); the backward then gets
    the residual path's gradient as an argument and adds it inside the LN backward kernel (`dres`) instead of autograd adding two [rows, D] fp32
    tensors in a pass of its own."""

    @staticmethod
    def forward(ctx, x, gid, ln_w, ln_b, shift, scale, eps, round_xhat, pad, passthrough):
        rows, D = x.shape
        x = x.contiguous()
        out = ops._padded_empty((rows,), D, pad, torch.bfloat16, x.device) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ms = 0 if shift is None else shift.stride(0)
        ops._timed("wan_ln_mod_fwd", (x.element_size() + 2.0) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_fwd", x, _ptr_dtype(x), gid, ln_w, ln_b, shift, scale, ms, rows, D, float(eps), int(round_xhat), out, D + pad, None, None,
            mean, rstd, _stream()), "byte")
        ctx.save_for_backward(x, mean, rstd, gid, ln_w, scale)
        ctx.ms = ms
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None, not as a zero tensor to be added
        if passthrough:
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, mean, rstd, gid, ln_w, scale = ctx.saved_tensors
        rows, D = x.shape
        if dy is None:
            return (None if dres is None else dres.to(x.dtype)), None, None, None, None, None, None, None, None, None
        if dres is not None and (dres.dtype != torch.float32 or not dres.is_contiguous()):
            dres = dres.float().contiguous()
        dx = torch.empty(rows, D, dtype=torch.float32, device=x.device)
        ops._timed("wan_ln_mod_bwd", (x.element_size() + (6.0 if dres is None else 10.0)) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_bwd", dy.contiguous(), x, _ptr_dtype(x), mean, rstd, gid, ln_w, scale, ctx.ms, rows, D, dres, dx, _stream()), "byte")
        return dx.to(x.dtype), None, None, None, None, None, None, None, None, None


class _HeadLnModFn(torch.autograd.Function):
    """fp32( LN_eps(x) * (1 + scale[gid]) + shift[gid] ),  x [rows, D] fp32: the output head's normalisation (csrc/wan.hip, the fp32-result form of the
    block kernels).  The modulation table takes no gradient (see _modulation_is_frozen: checked by the caller, as in the blocks)."""

    @staticmethod
    def forward(ctx, x, gid, shift, scale, eps):
        rows, D = x.shape
        x = x.contiguous()
        out = torch.empty(rows, D, dtype=torch.float32, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ms = shift.stride(0)
        ops._timed("wan_ln_mod_fwd", 8.0 * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_fwd_f32", x, gid, shift, scale, ms, rows, D, float(eps), out, mean, rstd, _stream()), "byte")
        ctx.save_for_backward(x, mean, rstd, gid, scale)
        ctx.ms = ms
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gid, scale = ctx.saved_tensors
        rows, D = x.shape
        dx = torch.empty(rows, D, dtype=torch.float32, device=x.device)
        ops._timed("wan_ln_mod_bwd", 12.0 * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_bwd_f32", dy.float().contiguous(), x, mean, rstd, gid, scale, ctx.ms, rows, D, dx, _stream()), "byte")
        return dx, None, None, None, None


class _GateLnFn(torch.autograd.Function):
    """A block's [gated residual add -> LayerNorm] pair as one node and one pass each way (csrc/wan.hip, GR / GB forms of the LN kernels):
         x' = x + y * gate[gid]  (fp32; gate None: 1),   h = bf16( LN_eps(x') * ln_w + ln_b, then * (1 + scale[gid]) + shift[gid] )      -> (h, x')
    x' feeds the branch's own residual add further down, so the backward gets both gradients, forms dx' = dres + LN-backward(dh) and, in the same kernel,
    dy = bf16(dx' * gate[gid]) -- written as the head of a buffer `dy_pad` wider (the LoRA tail of the output projection's backward GEMM).  Bit-identical to
    wan_gate_residual followed by wan_ln_mod (tests/test_gpu_wan_kernels.py); 12 instead of 16 bytes per element forward, 16 instead of 20 backward."""

    @staticmethod
    def forward(ctx, x, y, gid, gate, ln_w, ln_b, shift, scale, eps, pad, dy_pad):
        if x.dtype != torch.float32 or y.dtype != torch.bfloat16 or x.shape != y.shape:
            raise TypeError(f"gate_ln: fp32 residual stream and a bf16 branch output of the same shape, got {x.dtype} {tuple(x.shape)} / {y.dtype} {tuple(y.shape)}")
        rows, D = y.shape
        x, y = x.contiguous(), y.contiguous()
        xo = torch.empty(rows, D, dtype=torch.float32, device=y.device)
        out = ops._padded_empty((rows,), D, pad, torch.bfloat16, y.device) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=y.device)
        mean = torch.empty(rows, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        ms = gate.stride(0) if gate is not None else (shift.stride(0) if shift is not None else 0)
        if gate is not None and shift is not None and gate.stride(0) != shift.stride(0):
            raise ValueError("gate and shift / scale must be columns of one modulation table")
        ops._timed("wan_gate_ln_fwd", 12.0 * rows * D, lambda: _lib.call(
            "vgpa_wan_gate_ln_mod_fwd", x, y, gid, gate, ln_w, ln_b, shift, scale, ms, rows, D, float(eps), xo, out, D + pad, None, None, mean, rstd, _stream()), "byte")
        ctx.save_for_backward(xo, mean, rstd, gid, ln_w, scale, gate)
        ctx.ms, ctx.dy_pad = ms, dy_pad
        ctx.set_materialize_grads(False)
        return out, xo

    @staticmethod
    def backward(ctx, dh, dres):
        xo, mean, rstd, gid, ln_w, scale, gate = ctx.saved_tensors
        rows, D = xo.shape
        pad = ctx.dy_pad
        if dh is None:        # the LN output was not used: only the residual path carries a gradient (the arithmetic of _GateResidualFn.backward)
            if dres is None:
                return (None,) * 11
            dres = dres.float().contiguous()
            dy = ops._padded_empty((rows,), D, pad, torch.bfloat16, xo.device) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=xo.device)
            ops._timed("wan_gate_bwd", 6.0 * rows * D, lambda: _lib.call("vgpa_wan_gate_bwd", dres, gid, gate, ctx.ms, rows, D, dy, D + pad, _stream()), "byte")
            return (dres, dy) + (None,) * 9
        if dres is not None and (dres.dtype != torch.float32 or not dres.is_contiguous()):
            dres = dres.float().contiguous()
        dx = torch.empty(rows, D, dtype=torch.float32, device=xo.device)
        dy = ops._padded_empty((rows,), D, pad, torch.bfloat16, xo.device) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=xo.device)
        ops._timed("wan_ln_gate_bwd", (12.0 if dres is None else 16.0) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_bwd_gate", dh.contiguous(), xo, mean, rstd, gid, ln_w, scale, ctx.ms, rows, D, dres, dx, gate, dy, D + pad, _stream()), "byte")
        return dx, dy, None, None, None, None, None, None, None, None, None


class _GateResidualFn(torch.autograd.Function):
    """fp32: x + y(bf16) * gate[gid]      (gate None: 1).  dy_pad: the gradient of y is returned as the head of a buffer that much wider (the
    LoRA tail of the output projection's backward GEMM)."""

    @staticmethod
    def forward(ctx, x, y, gid, gate, dy_pad):
        rows, D = y.shape
        out = torch.empty(rows, D, dtype=torch.float32, device=y.device)
        ms = 0 if gate is None else gate.stride(0)
        ops._timed("wan_gate_residual", 10.0 * rows * D, lambda: _lib.call(
            "vgpa_wan_gate_residual", x.contiguous(), y.contiguous(), gid, gate, ms, rows, D, out, _stream()), "byte")
        ctx.save_for_backward(gid, gate)
        ctx.ms, ctx.dy_pad = ms, dy_pad
        return out

    @staticmethod
    def backward(ctx, dout):
        gid, gate = ctx.saved_tensors
        rows, D = dout.shape
        pad = ctx.dy_pad
        dy = ops._padded_empty((rows,), D, pad, torch.bfloat16, dout.device) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=dout.device)
        dout = dout.contiguous()
        ops._timed("wan_gate_bwd", 6.0 * rows * D, lambda: _lib.call("vgpa_wan_gate_bwd", dout, gid, gate, ctx.ms, rows, D, dy, D + pad, _stream()), "byte")
        return dout, dy, None, None, None


def _rows2(t, D):
    """[.., D] tensor with a contiguous last dim and uniformly strided rows -> ([rows, D] view, row stride); copies only if it has to"""
    if t.stride(-1) != 1:
        t = t.contiguous()
    try:
        t2 = t.view(-1, D)
    except RuntimeError:
        t2 = t.contiguous().view(-1, D)
    return t2, t2.stride(0)


def _rms_rope_fwd_raw(u2, ld_u, w, cos, sin, L, head_dim, eps, out2, ld_out, rstd):
    rows, D = u2.shape
    ops._timed("wan_rms_rope_fwd", 4.0 * rows * D, lambda: _lib.call(
        "vgpa_wan_rms_rope_fwd", u2, ld_u, w, cos, sin, L, head_dim, rows, D, float(eps), out2, ld_out, rstd, _stream()), "byte")


def _rms_rope_bwd_raw(dout2, ld_dout, u2, ld_u, rstd, w, cos, sin, L, head_dim, du2, ld_du):
    rows, D = u2.shape
    ops._timed("wan_rms_rope_bwd", 6.0 * rows * D, lambda: _lib.call(
        "vgpa_wan_rms_rope_bwd", dout2, ld_dout, u2, ld_u, rstd, w, cos, sin, L, head_dim, rows, D, du2, ld_du, _stream()), "byte")


class _RmsRopeFn(torch.autograd.Function):
    """WanRMSNorm over the full row, bf16 weight, RoPE per head:  u [B, L, D] bf16 (rows may be strided) -> [B, L, D] bf16.  grad_pad: du is the
    head of a buffer that much wider (LoRA tail of the producing projection's backward GEMM)."""

    @staticmethod
    def forward(ctx, u, w, cos, sin, head_dim, eps, grad_pad):
        B, L, D = u.shape
        u2, ld_u = _rows2(u, D)
        out = torch.empty(B, L, D, dtype=torch.bfloat16, device=u.device)
        rstd = torch.empty(B * L, dtype=torch.float32, device=u.device)
        _rms_rope_fwd_raw(u2, ld_u, w, cos, sin, L, head_dim, eps, out, D, rstd)
        ctx.save_for_backward(u2, rstd, w, cos, sin)
        ctx.meta = (B, L, D, head_dim, grad_pad)
        return out

    @staticmethod
    def backward(ctx, dout):
        u2, rstd, w, cos, sin = ctx.saved_tensors
        B, L, D, head_dim, pad = ctx.meta
        du = ops._padded_empty((B, L), D, pad, torch.bfloat16, u2.device) if pad else torch.empty(B, L, D, dtype=torch.bfloat16, device=u2.device)
        d2, ld_d = _rows2(dout, D)
        _rms_rope_bwd_raw(d2, ld_d, u2, u2.stride(0), rstd, w, cos, sin, L, head_dim, du, D + pad)
        return du, None, None, None, None, None, None


class _SelfAttnFn(torch.autograd.Function):
    """qkv [B, L, 3 D] (the fused q/k/v projection's output) -> attention output [B, L, D]: WanRMSNorm + RoPE on the q and k slices, v read from the
    buffer by stride, head_dim-128 attention; the backward writes the three gradients straight into one [B, L, 3 D (+ grad_pad)] buffer (dv by the
    attention kernel, dq / dk by the norm backward), so the fused projection's backward GEMM needs no concatenation.  o_pad / grad_pad: LoRA tails of
    the output projection's forward resp. the q/k/v projection's backward operand (ops.LoraExt)."""

    @staticmethod
    def forward(ctx, qkv, wq, wk, cos, sin, H, eps, o_pad, grad_pad, f8=False, precise_delta=None, f8_policy=None):
        B, L, W = qkv.shape
        D = W // 3
        hd = D // H
        q2, ld = _rows2(qkv, W)
        qn = torch.empty(B, L, D, dtype=torch.bfloat16, device=qkv.device)
        kn = torch.empty_like(qn)
        rq = torch.empty(B * L, dtype=torch.float32, device=qkv.device)
        rk = torch.empty_like(rq)
        _rms_rope_fwd_raw(q2[:, :D], ld, wq, cos, sin, L, hd, eps, qn, D, rq)
        _rms_rope_fwd_raw(q2[:, D:2 * D], ld, wk, cos, sin, L, hd, eps, kn, D, rk)
        heads = lambda t: t.unflatten(-1, (H, hd)).permute(0, 2, 1, 3)
        v = heads(q2.view(B, L, W)[:, :, 2 * D:])
        # eight further mantissa bits of the output for the backward's delta (ops.py "Precise delta"), where a backward will run
        o_res8 = torch.empty(B, L, D, dtype=torch.uint8, device=qkv.device) if precise_delta and ctx.needs_input_grad[0] else None
        # e4m3 forward with a backward to come: the backward runs on the operands the forward's products really used (dequantised to bf16 by the forward's
        # own quantisation pass), so that its recomputed softmax weights ARE the forward's and delta = rowsum(dO o O) matches them -- the straight-through
        # gradient of this forward.  They replace the normalised q / k among the saved tensors; the dequantised v is the one tensor this adds.
        if f8 and f8_policy is not None:      # enable_fp8(attention="auto"): the layer's first call decides from the score range whether e4m3 scores are accurate enough
            f8 = f8_policy.use_f8(qn, kn, H, hd ** -0.5)
        deq = None
        if ops.attention128_uses_f8(f8, L) and ctx.needs_input_grad[0]:
            deq = tuple(torch.empty(B, L, D, dtype=torch.bfloat16, device=qkv.device) for _ in range(3))
        o, lse = ops.attention128_fwd_raw(heads(qn), heads(kn), v, hd ** -0.5, o_pad, f8=f8, o_res8=o_res8, deq=deq)      # f8: e4m3 matrix operands (enable_fp8(attention=True))
        if deq is None:
            ctx.save_for_backward(q2, qn, kn, None, o, lse, rq, rk, wq, wk, cos, sin, o_res8)
        else:
            ctx.save_for_backward(q2, deq[0], deq[1], deq[2], o, lse, rq, rk, wq, wk, cos, sin, o_res8)
        ctx.meta = (B, L, D, H, hd, grad_pad)
        return o.permute(0, 2, 1, 3).flatten(2)          # [B, L, D]: a view of the token-major storage

    @staticmethod
    def backward(ctx, do):
        q2, qn, kn, vd, o, lse, rq, rk, wq, wk, cos, sin, o_res8 = ctx.saved_tensors      # (qn, kn, vd): the e4m3 forward's dequantised operands when it ran
        B, L, D, H, hd, pad = ctx.meta
        W = 3 * D
        heads = lambda t: t.unflatten(-1, (H, hd)).permute(0, 2, 1, 3)
        do = do if do.stride(-1) == 1 else do.contiguous()
        dqkv = ops._padded_empty((B, L), W, pad, torch.bfloat16, q2.device) if pad else torch.empty(B, L, W, dtype=torch.bfloat16, device=q2.device)
        dqn = torch.empty(B, L, D, dtype=torch.bfloat16, device=q2.device)
        dkn = torch.empty_like(dqn)
        v = heads(q2.view(B, L, W)[:, :, 2 * D:]) if vd is None else heads(vd)
        # vd is not None: (qn, kn, vd) are the e4m3 forward's own operands, qn pre-scaled by hd^-1/2 log2(e) (exact): the prescaled backward recomputes its scores bit for bit
        ops.attention128_bwd_raw(heads(qn), heads(kn), v, o, heads(do), lse, heads(dqn), heads(dkn), heads(dqkv[:, :, 2 * D:]), hd ** -0.5, o_res8=o_res8,
                                 q_prescaled=vd is not None)
        d2 = dqkv.view(B * L, W)
        _rms_rope_bwd_raw(dqn, D, q2[:, :D], q2.stride(0), rq, wq, cos, sin, L, hd, d2[:, :D], d2.stride(0))
        _rms_rope_bwd_raw(dkn, D, q2[:, D:2 * D], q2.stride(0), rk, wk, cos, sin, L, hd, d2[:, D:2 * D], d2.stride(0))
        return dqkv, None, None, None, None, None, None, None, None, None, None, None


class _FfnFp8Fn(torch.autograd.Function):
    """One feed-forward branch of a block with e4m3 GEMM operands, x' = x + gate[g] * W2 gelu(W1 ln_mod(x) + b1) + b2, as ONE autograd node: every
    fp8 operand is written by its producer (LN + modulation, GELU, the gate's backward, GELU's backward -- csrc/wan.hip, csrc/fp8.hip; bit-identical
    to the bf16 producer followed by vgpa_quant_fp8_rows), the vendor fp8 GEMM (hipBLASLt through torch._scaled_mm) does the four products, and the
    backward adds the residual path's gradient inside the LN backward kernel (dres)."""

    @staticmethod
    def forward(ctx, x, gid, shift, scale, gate, eps, W1, b1, W2, b2, pre=None, pre_dy_pad=0):
        """pre (bf16 [rows, D]): the branch input is x + pre -- the ungated residual add of the cross-attention in front of this branch, taken into the
        LN pass (gate_ln_mod_fwd); the backward then also returns pre's gradient (its head in a buffer pre_dy_pad wider)."""
        rows, D = x.shape
        x = x.contiguous()
        dev = x.device
        w1, w2 = ops.Fp8Weight.of(W1), ops.Fp8Weight.of(W2)
        hq = torch.empty(rows, D, dtype=torch.float8_e4m3fn, device=dev)
        hs = torch.empty(rows, 1, dtype=torch.float32, device=dev)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        ms = shift.stride(0)
        ctx.pre_dy_pad = None if pre is None else int(pre_dy_pad)
        if pre is not None and (pre.dtype != torch.bfloat16 or pre.shape != x.shape):
            raise TypeError(f"ffn_fp8: `pre` is the bf16 output of the branch in front, shape {tuple(x.shape)}; got {pre.dtype} {tuple(pre.shape)}")
        if pre is not None:
            xin, x = x, torch.empty(rows, D, dtype=torch.float32, device=dev)
            ops._timed("wan_gate_ln_fwd", 11.0 * rows * D, lambda: _lib.call(
                "vgpa_wan_gate_ln_mod_fwd", xin, pre.contiguous(), gid, None, None, None, shift, scale, ms, rows, D, float(eps), x, None, D, hq, hs, mean, rstd,
                _stream()), "byte")
        else:
            ops._timed("wan_ln_mod_fwd", (x.element_size() + 1.0) * rows * D, lambda: _lib.call(
                "vgpa_wan_ln_mod_fwd", x, _ptr_dtype(x), gid, None, None, shift, scale, ms, rows, D, float(eps), 0, None, D, hq, hs, mean, rstd, _stream()), "byte")
        u = ops._fp8_gemm(hq, hs, w1.q, w1.s, b1)
        gq, gs = ops.gelu_tanh_fwd_q8(u)
        y = ops._fp8_gemm(gq, gs, w2.q, w2.s, b2)
        out = torch.empty(rows, D, dtype=torch.float32, device=dev)
        ops._timed("wan_gate_residual", 10.0 * rows * D, lambda: _lib.call("vgpa_wan_gate_residual", x, y, gid, gate, gate.stride(0), rows, D, out, _stream()), "byte")
        ctx.save_for_backward(x, mean, rstd, u, gid, scale, gate)
        ctx.w, ctx.ms = (w1, w2), ms
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mean, rstd, u, gid, scale, gate = ctx.saved_tensors
        w1, w2 = ctx.w
        rows, D = x.shape
        dev = x.device
        dout = dout.contiguous()
        dq = torch.empty(rows, D, dtype=torch.float8_e4m3fn, device=dev)
        ds = torch.empty(rows, 1, dtype=torch.float32, device=dev)
        ops._timed("wan_gate_bwd", 5.0 * rows * D, lambda: _lib.call("vgpa_wan_gate_bwd_q8", dout, gid, gate, gate.stride(0), rows, D, dq, ds, _stream()), "byte")
        dg = ops._fp8_gemm(dq, ds, w2.qt, w2.st)
        duq, dus = ops.gelu_tanh_bwd_q8(u, dg)
        dh = ops._fp8_gemm(duq, dus, w1.qt, w1.st)
        dx = torch.empty(rows, D, dtype=torch.float32, device=dev)
        if ctx.pre_dy_pad is not None:
            pad = ctx.pre_dy_pad
            dpre = ops._padded_empty((rows,), D, pad, torch.bfloat16, dev) if pad else torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
            ops._timed("wan_ln_gate_bwd", 16.0 * rows * D, lambda: _lib.call(
                "vgpa_wan_ln_mod_bwd_gate", dh, x, mean, rstd, gid, None, scale, ctx.ms, rows, D, dout, dx, None, dpre, D + pad, _stream()), "byte")
            return dx, None, None, None, None, None, None, None, None, None, dpre, None
        ops._timed("wan_ln_mod_bwd", (x.element_size() + 10.0) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_bwd", dh, x, _ptr_dtype(x), mean, rstd, gid, None, scale, ctx.ms, rows, D, dout, dx, _stream()), "byte")
        return dx, None, None, None, None, None, None, None, None, None, None, None


def ln_mod(x, gid=None, ln_w=None, ln_b=None, shift=None, scale=None, eps=1e-6, round_xhat=False, pad=0, passthrough=False):
    """passthrough=True returns (LN output, x): feed THAT x into the branch's residual add (see _LnModFn)"""
    return _LnModFn.apply(x, gid, ln_w, ln_b, shift, scale, eps, round_xhat, int(pad), bool(passthrough))


def gate_residual(x, y, gid=None, gate=None, dy_pad=0):
    return _GateResidualFn.apply(x, y, gid, gate, int(dy_pad))


def gate_ln(x, y, gid=None, gate=None, ln_w=None, ln_b=None, shift=None, scale=None, eps=1e-6, pad=0, dy_pad=0):
    """(LN output of x + y * gate, that sum): gate_residual + ln_mod(passthrough=True) in one pass each way (see _GateLnFn)"""
    return _GateLnFn.apply(x, y, gid, gate, ln_w, ln_b, shift, scale, eps, int(pad), int(dy_pad))


def rms_rope(u, w, cos=None, sin=None, head_dim=128, eps=1e-6, grad_pad=0):
    return _RmsRopeFn.apply(u, w, cos, sin, head_dim, eps, int(grad_pad))


def ffn_fp8(x, gid, shift, scale, gate, eps, W1, b1, W2, b2, pre=None, pre_dy_pad=0):
    """x fp32 [rows, D] (+ pre, bf16: a residual add taken into the LN pass) -> x' + gate[g] * FFN(ln_mod(x')) with e4m3 GEMM operands (frozen weights)"""
    if W1.requires_grad or W2.requires_grad or (b1 is not None and b1.requires_grad) or (b2 is not None and b2.requires_grad):
        raise RuntimeError("videogpa_amd: the fp8 path is for frozen projections only")
    if x.dtype != torch.float32:
        raise TypeError("ffn_fp8: fp32 residual stream")
    return _FfnFp8Fn.apply(x, gid, shift, scale, gate, eps, W1, b1, W2, b2, pre, int(pre_dy_pad))


def sinusoidal_embedding_1d(dim, position):
    """upstream sinusoidal_embedding_1d: float64 outer product, [cos | sin]"""
    half = dim // 2
    position = position.to(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, dtype=torch.float64, device=position.device).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_tables(grid, head_dim, device, theta=10000.0):
    """cos / sin [f*h*w, head_dim/2] fp32 of upstream rope_params + rope_apply: the head's complex pairs split [c - 2(c//3), c//3, c//3]
    over (frame, row, column), each axis with its own frequency ladder 1 / theta^(2i / axis_dim); angles in float64."""
    f, h, w = grid
    d = head_dim
    dims = (d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6))
    ang = []
    for n, ax_dim, shape in ((f, dims[0], (f, 1, 1)), (h, dims[1], (1, h, 1)), (w, dims[2], (1, 1, w))):
        fr = 1.0 / torch.pow(theta, torch.arange(0, ax_dim, 2, dtype=torch.float64, device=device).div(ax_dim))
        a = torch.outer(torch.arange(n, dtype=torch.float64, device=device), fr)
        ang.append(a.view(*shape, -1).expand(f, h, w, -1))
    ang = torch.cat(ang, dim=-1).reshape(f * h * w, d // 2)
    return torch.cos(ang).float().contiguous(), torch.sin(ang).float().contiguous()


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim, self.eps = dim, num_heads, dim // num_heads, eps
        if tuple(window_size) != (-1, -1):
            raise NotImplementedError("windowed attention is not on the training path")
        if self.head_dim != 128:
            raise NotImplementedError("csrc/attention_hd128.hip: head_dim 128 (every released Wan model)")
        self.q, self.k, self.v, self.o = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else None
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else None
        self._fused = None
        self._qkv_ext = None
        self.fp8_attn = False
        self.f8_policy = None                 # ops.F8AttnPolicy with enable_fp8(attention="auto")
        self.precise_delta = "int8" if ops.precise_delta_default() else None     # ops.py "Precise delta"; WanModel.set_precise_delta changes it per model

    def _heads(self, t, B, S):
        return t.view(B, S, self.num_heads, self.head_dim).permute(0, 2, 1, 3)

    def _norm(self, norm, u, rope, grad_pad=0):
        if norm is None:
            if rope is not None:
                raise NotImplementedError("RoPE without QK-norm")
            return u
        return rms_rope(u, norm.weight, rope[0] if rope else None, rope[1] if rope else None, self.head_dim, norm.eps, grad_pad)

    # ---- the q / k / v linears as ONE projection (the CogVideoX path's AttentionCore, transformer.py): [3 dim, dim] weight cached while the frozen
    # base weights are unchanged, the (up to three) adapters riding the GEMM as extra K (ops.LoraExt)
    def fused_qkv(self):
        ws = [_parts(m)[0] for m in (self.q, self.k, self.v)]
        key = tuple((w.data_ptr(), w._version, w.dtype, w.device) for w in ws)
        if self._fused is None or self._fused[0] != key:
            bs = [_parts(m)[1] for m in (self.q, self.k, self.v)]
            W = torch.cat([w.detach() for w in ws], dim=0)
            b = torch.cat([x.detach() for x in bs], dim=0) if bs[0] is not None else None
            self._fused = (key, W, b)
        return self._fused[1], self._fused[2]

    def lora_state(self):
        """(adapters of q / k / v, adapter of o, enabled): all wrapped linears of one attention share the on / off state"""
        qa = [_adapter(m) for m in (self.q, self.k, self.v)]
        oa = _adapter(self.o)
        on = [e for a, e in qa + [oa] if a is not None]
        return [a for a, _ in qa], oa[0], (bool(on) and all(on))

    def pads(self):
        """(in_pad, out_pad): LoRA tail widths behind this attention's input rows (q/k/v adapters) and behind its output rows (o adapter)"""
        qa, oa, _ = self.lora_state()
        act = [a for a in qa if a is not None]
        return (len(act) * ops._pad_rank(act[0][0].shape[0]) if act else 0), (ops._pad_rank(oa[0].shape[0]) if oa is not None else 0)

    def forward(self, x, B, L, rope):
        """x [B*L, dim] bf16 (head of a buffer with pads()[0] tail columns when adapters are mounted) -> [B*L, dim] bf16"""
        if self.norm_q is None:
            raise NotImplementedError("the fused self-attention path normalises q and k (qk_norm=True in every released Wan model)")
        W, b = self.fused_qkv()
        qa, _, on = self.lora_state()
        in_pad, out_pad = self.pads()
        if in_pad:
            if self._qkv_ext is None:
                self._qkv_ext = ops.LoraExt()
            qkv = ops.linear_lora_ext(x, W, b, self._qkv_ext, qa, enabled=on)
        else:
            qkv = ops.frozen_linear(x, W, b)
        o = _SelfAttnFn.apply(qkv.view(B, L, 3 * self.dim), self.norm_q.weight, self.norm_k.weight, rope[0] if rope else None, rope[1] if rope else None,
                              self.num_heads, self.norm_q.eps, out_pad, in_pad, bool(self.fp8_attn), self.precise_delta, self.f8_policy)
        return self.o(o.reshape(B * L, self.dim))


class WanCrossAttention(WanSelfAttention):
    def pads(self):
        qa, oa = _adapter(self.q)[0], _adapter(self.o)[0]
        return (ops._pad_rank(qa[0].shape[0]) if qa is not None else 0), (ops._pad_rank(oa[0].shape[0]) if oa is not None else 0)

    def forward(self, x, context, B, L):
        """x [B*L, dim] bf16, context [B, T, dim] bf16 (every one of the T text positions is attended: upstream passes k_lens=None)"""
        T = context.shape[1]
        in_pad, out_pad = self.pads()
        q = self._norm(self.norm_q, self.q(x).view(B, L, self.dim), None, grad_pad=in_pad)
        k = self._norm(self.norm_k, self.k(context.reshape(B * T, self.dim)).view(B, T, self.dim), None)
        v = self.v(context.reshape(B * T, self.dim)).view(B, T, self.dim)
        o = ops.attention128(self._heads(q, B, L), self._heads(k, B, T), self._heads(v, B, T), o_pad=out_pad, precise_delta=self.precise_delta)
        return self.o(o.permute(0, 2, 1, 3).reshape(B * L, self.dim))


class WanAttentionBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanCrossAttention(dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self.fp8_ffn = False

    def forward(self, x, e0, gid, B, L, rope, context):
        """x [B*L, dim]: bf16 in the first block (the patch embedding's output), fp32 afterwards; e0 [G, 6, dim] fp32; -> fp32"""
        tab = (self.modulation.float() + e0).contiguous()          # [G, 6, dim] fp32: upstream adds under autocast(float32)
        _modulation_is_frozen(tab, "WanAttentionBlock")
        first = x.dtype == torch.bfloat16                           # norm1(x).type_as(x) rounds only while the stream is still bf16
        sa_in, sa_out = self.self_attn.pads()
        ca_in, ca_out = self.cross_attn.pads()
        if first:         # block 0 takes the bf16 patch embedding; the stream is fp32 from its first residual add on
            h = ln_mod(x, gid, None, None, tab[:, 0], tab[:, 1], self.eps, round_xhat=True, pad=sa_in)
            x = x.float()
        else:
            h, x = ln_mod(x, gid, None, None, tab[:, 0], tab[:, 1], self.eps, pad=sa_in, passthrough=True)
        # every gated residual add is followed by a LayerNorm of its result: the two run as one pass (gate_ln) wherever both ends are in this block
        sa = self.self_attn(h, B, L, rope)
        if isinstance(self.norm3, nn.Identity):
            x = gate_residual(x, sa, gid, tab[:, 2], dy_pad=sa_out)
            h = x.to(torch.bfloat16)
        else:
            h, x = gate_ln(x, sa, gid, tab[:, 2], _f32(self.norm3.weight), _f32(self.norm3.bias), None, None, self.eps, pad=ca_in, dy_pad=sa_out)
        ca = self.cross_attn(h, context, B, L)
        f0, f2 = self.ffn[0], self.ffn[2]
        if self.fp8_ffn and f0.weight.shape[0] <= 16384:       # one feed-forward row per workgroup in the GELU -> e4m3 kernels (csrc/fp8.hip)
            return ffn_fp8(x, gid, tab[:, 3], tab[:, 4], tab[:, 5], self.eps, f0.weight, f0.bias, f2.weight, f2.bias, pre=ca, pre_dy_pad=ca_out)
        lin = ops.frozen_linear_fp8 if self.fp8_ffn else ops.frozen_linear
        h, x = gate_ln(x, ca, gid, None, None, None, tab[:, 3], tab[:, 4], self.eps, dy_pad=ca_out)
        y = lin(ops.gelu_tanh(lin(h, f0.weight, f0.bias)), f2.weight, f2.bias)
        return gate_residual(x, y, gid, tab[:, 5])


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, out_dim * math.prod(patch_size))
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e, gid):
        """x [B*L, dim] fp32, e [G, dim] fp32 -> [B*L, out] fp32: upstream runs the whole head under autocast(float32).  LN + per-token modulation
        is the row kernel of the blocks with an fp32 result (the modulation table is indexed inside the kernel: no [L, dim] gather), then one
        fp32 library GEMM."""
        tab = (self.modulation.float() + e[:, None]).contiguous()   # [G, 2, dim]
        _modulation_is_frozen(tab, "Head")
        h = _HeadLnModFn.apply(x, gid, tab[:, 0], tab[:, 1], self.eps)
        w, b = self.head.weight, self.head.bias
        return F.linear(h, w.float() if w.requires_grad else _f32(w), b.float() if b.requires_grad else _f32(b))


class WanModel(nn.Module):
    def __init__(self, model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256, text_dim=4096,
                 out_dim=48, num_heads=24, num_layers=30, window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6):
        super().__init__()
        self.gemm_rows_per_sample = True      # batched samples go through the bf16 vendor GEMMs one sample per call (ops.gemm_rows_per_call)
        assert model_type in ("t2v", "i2v", "ti2v", "s2v")
        self.model_type, self.patch_size, self.text_len, self.in_dim, self.dim, self.ffn_dim = model_type, tuple(patch_size), text_len, in_dim, dim, ffn_dim
        self.freq_dim, self.text_dim, self.out_dim, self.num_heads, self.num_layers, self.eps = freq_dim, text_dim, out_dim, num_heads, num_layers, eps
        # ConfigMixin's `model.config`: the registered constructor arguments (attribute and item access)
        from .transformer import _Config
        self.config = _Config(model_type=model_type, patch_size=list(patch_size), text_len=text_len, in_dim=in_dim, dim=dim, ffn_dim=ffn_dim, freq_dim=freq_dim,
                              text_dim=text_dim, out_dim=out_dim, num_heads=num_heads, num_layers=num_layers, window_size=list(window_size), qk_norm=qk_norm,
                              cross_attn_norm=cross_attn_norm, eps=eps)
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([WanAttentionBlock(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps) for _ in range(num_layers)])
        self.head = Head(dim, out_dim, self.patch_size, eps)
        self.gradient_checkpointing = False
        self.checkpoint_stride = 1
        self._rope = {}
        self.init_weights()

    # ------------------------------------------------------------------ upstream (diffusers ModelMixin / ConfigMixin) loading protocol
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"
    _CONFIG_KEYS = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim", "num_heads",
                    "num_layers", "window_size", "qk_norm", "cross_attn_norm", "eps")

    @classmethod
    def from_config(cls, config, **kw):
        cfg = {k: v for k, v in dict(config).items() if k in cls._CONFIG_KEYS}
        cfg.update(kw)
        for k in ("patch_size", "window_size"):
            if k in cfg:
                cfg[k] = tuple(cfg[k])
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        """`WanModel.from_pretrained(config['model_path'])` -- how the reference builds policy AND reference model
        (train/Wan2.2-TI2V-5B/03_train.py:140,166; generate/Wan2.2-TI2V-5B.py).  Upstream WanModel is a diffusers ModelMixin with
        @register_to_config, so a checkpoint directory holds `config.json` (the constructor arguments + `_class_name` / `_diffusers_version`)
        and the weights as `diffusion_pytorch_model.safetensors` or as shards `diffusion_pytorch_model-0000i-of-0000n.safetensors` listed by
        `diffusion_pytorch_model.safetensors.index.json` (`weight_map`: parameter name -> shard file).  Parameter names are the upstream ones
        (module docstring), loaded strictly.  dtype as diffusers' ModelMixin.from_pretrained: torch_dtype=None gives float32 parameters whatever the
        checkpoint stores (the reference casts right after: `.to(torch.bfloat16)`, 03_train.py:141), a torch.dtype casts to it, and "auto" keeps the stored
        dtype of every tensor (mixed checkpoints included).  No hub access: `path` must be a local directory."""
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder else path
        if not os.path.isdir(root):
            raise FileNotFoundError(f"WanModel.from_pretrained: {root!r} is not a local directory (there is no hub access on this path)")
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        unknown = sorted(k for k in cfg if k not in cls._CONFIG_KEYS and not k.startswith("_"))
        if unknown:
            raise ValueError(f"WanModel.from_pretrained: config keys this model does not implement: {unknown}")
        index = os.path.join(root, cls.weights_name + ".index.json")
        if os.path.isfile(index):
            with open(index) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        elif os.path.isfile(os.path.join(root, cls.weights_name)):
            files = [cls.weights_name]
        else:
            files = sorted(f for f in os.listdir(root) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors weights under {root}")
        sd = {}
        for fn in files:
            sd.update(load_file(os.path.join(root, fn)))
        with torch.device("meta"):
            model = cls.from_config(cfg, **kw)
        if torch_dtype != "auto":
            want = torch.float32 if torch_dtype is None else torch_dtype
            sd = {k: (v.to(want) if v.is_floating_point() else v) for k, v in sd.items()}      # per tensor, before the assign: never two full copies
        model.load_state_dict(sd, strict=True, assign=True)          # meta skeleton + assign: no second copy of a 5 B-parameter model, no random init
        model._rope = {}
        model.eval()                                                  # diffusers' from_pretrained returns the model in eval mode
        return model

    def save_pretrained(self, path, max_shard_size=None, safe_serialization=True):
        """the upstream on-disk layout (see from_pretrained); `max_shard_size` bytes (int) splits the weights into indexed shards"""
        from safetensors.torch import save_file
        if not safe_serialization:
            raise NotImplementedError("safetensors only")
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "WanModel"
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if max_shard_size is None:
            save_file(sd, os.path.join(path, self.weights_name))
            return
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            b = v.numel() * v.element_size()
            if cur and size + b > int(max_shard_size):
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += b
        shards.append(cur)
        stem = self.weights_name[:-len(".safetensors")]
        wmap = {}
        for i, sh in enumerate(shards):
            fn = f"{stem}-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(path, fn))
            wmap.update({k: fn for k in sh})
        with open(os.path.join(path, self.weights_name + ".index.json"), "w") as f:
            json.dump({"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": wmap}, f, indent=2)

    def init_weights(self):
        """upstream init_weights: xavier on linears, patch embedding, normal(0.02) on the two embedding MLPs, zero output head"""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for m in self.text_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
        for m in self.time_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
        nn.init.zeros_(self.head.head.weight)

    def enable_gradient_checkpointing(self, enabled=True, stride=1):
        """the reference wraps every block's forward in torch.utils.checkpoint (03_train.py:150-159: sized for 80 GB parts).  `stride` k
        recomputes only every k-th block; 288 GB of HBM3E hold ALL activations of the full-size pair step (30 blocks x 2 samples x
        18 480 tokens: DESIGN section 4.4), so the MI355X default of the trainer / bench is enabled=False -- one forward in five saved."""
        self.gradient_checkpointing = bool(enabled)
        self.checkpoint_stride = max(1, int(stride))

    def enable_fp8(self, enabled=True, attention=None):
        """BASELINE.json configs[4] "fp8 MFMA path".  (i) The frozen feed-forward projections (61 % of the linear FLOPs per token) take OCP-e4m3 operands --
        per-row dynamic activation scales (csrc/fp8.hip), per-output-row weight scales, fp32 accumulation, bf16 out, forward and dX (the vendor's fp8 GEMM).
        (ii) attention (default: as `enabled`): the self-attention FORWARD -- reference pass and policy pass alike, so the Diffusion-DPO identity loss = ln 2
        at B = 0 stays exact -- runs the hand-written e4m3 kernel (csrc/attention_hd128.hip attn128_fwd_f8_kernel: both products as
        v_mfma_scale_f32_32x32x64_f8f6f4, power-of-two scales on the instruction's E8M0 operands); its backward runs the bf16 kernels on the forward's own operands, dequantised
        (_SelfAttnFn: the straight-through gradient of the e4m3 forward -- softmax rows that sum to one, delta consistent with them); the 512-key
        cross-attention stays bf16.
        attention="auto": each self-attention layer decides at its first call whether e4m3 SCORES are accurate enough on its data (ops.F8AttnPolicy: the score error
        grows with |q| |k|; a layer whose estimated rms score error exceeds `F8AttnPolicy.threshold` = 0.5 log2 units keeps the bf16 forward) -- for checkpoints
        whose QK-norm gains are not known to be small; True pins the e4m3 kernel (what bench.py measures).
        The LoRA-carrying q/k/v/o projections stay in bf16."""
        if attention not in (None, True, False, "auto"):
            raise ValueError(f'enable_fp8: attention is True, False, None (= enabled) or "auto", got {attention!r}')
        for blk in self.blocks:
            blk.fp8_ffn = enabled
            blk.self_attn.fp8_attn = bool(enabled if attention is None else attention)
            blk.self_attn.f8_policy = ops.F8AttnPolicy() if attention == "auto" else None

    def fp8_attention_report(self):
        """per block: whether the self-attention forward runs e4m3 and, under attention="auto", the estimated score error the decision was taken on"""
        return [{"block": i, "fp8_attn": bool(b.self_attn.fp8_attn) and (b.self_attn.f8_policy is None or b.self_attn.f8_policy.mode == "f8"),
                 "estimated_score_error_log2": None if b.self_attn.f8_policy is None else b.self_attn.f8_policy.estimated_score_error} for i, b in enumerate(self.blocks)]

    def set_precise_delta(self, mode="int8"):
        """what both attentions of every block of THIS model keep of their output beyond its bf16 rounding for the backward's delta: "int8" (default, one
        byte per output element) or None (the textbook flash-attention backward) -- ops.py "Precise delta" has the why"""
        if mode not in (None, "int8"):
            raise ValueError(f'precise_delta: "int8" or None, got {mode!r}')
        for blk in self.blocks:
            blk.self_attn.precise_delta = blk.cross_attn.precise_delta = mode

    def _rope_tables(self, grid, device):
        key = (tuple(grid), str(device))
        if key not in self._rope:
            self._rope = {key: rope_tables(grid, self.dim // self.num_heads, device)}
        return self._rope[key]

    def forward(self, x, t, context, seq_len, y=None, timestep_groups=None):
        """upstream call convention (module docstring).  A grad-enabled forward needs the base model FROZEN: gradients reach the LoRA adapters (and the
        inputs), not the modulation tables / time embedding (_modulation_is_frozen raises otherwise)."""
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        xb = torch.stack(list(x))
        B, C, Fr, H, W = xb.shape
        pt, ph, pw = self.patch_size
        f, h, w = Fr // pt, H // ph, W // pw
        L = f * h * w
        if seq_len != L:
            raise NotImplementedError(f"seq_len {seq_len} != token count {L}: padded sequences are not on the training path (03_train.py:176-179)")
        rows_per_call = L if B > 1 and self.gemm_rows_per_sample else 0      # one vendor-GEMM call per sample (ops.gemm_rows_per_call has the measurement)
        dev = xb.device
        wdt = self.patch_embedding.weight.dtype
        # patch embedding: Conv3d with kernel = stride = patch  ==  one GEMM over the (c, pt, ph, pw) patch vectors
        patches = xb.view(B, C, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * L, C * pt * ph * pw).to(wdt)
        tok = F.linear(patches, self.patch_embedding.weight.view(self.dim, -1), self.patch_embedding.bias)           # [B*L, dim]
        # time embedding on the distinct timesteps only -- grouped WITHOUT a host round trip (torch.unique would sort and synchronise in every forward):
        #   t [B]     one group per sample;
        #   t [B, L]  the TI2V form (03_train.py:119-125,181-187): a sample's tokens carry its first token's value (the clean first latent frame: 0) or
        #             ONE other value -> groups (2b, 2b + 1), table values (t[b, 0], that other value); checked on the device, asynchronously.
        #   anything richer: pass timestep_groups=(values [G], gid [B*L] int32).
        if timestep_groups is not None:
            tvals, gid = timestep_groups[0].reshape(-1).float(), timestep_groups[1].reshape(-1).to(torch.int32).contiguous()
        elif t.dim() == 1:
            tvals = t.reshape(B).float()
            gid = torch.arange(B, device=dev, dtype=torch.int32)[:, None].expand(B, L).reshape(-1).contiguous()
        else:
            tf = t.reshape(B, L).float()
            first = tf[:, :1]
            is_other = tf != first
            other = torch.where(is_other, tf, tf.new_full((), float("-inf"))).amax(dim=1, keepdim=True)
            other = torch.where(torch.isinf(other), first, other)         # a sample whose tokens all share one value
            torch._assert_async((~is_other | (tf == other)).all())      # at most two distinct values per sample
            tvals = torch.cat([first, other], dim=1).reshape(-1)
            gid = (2 * torch.arange(B, device=dev, dtype=torch.int32)[:, None] + is_other.to(torch.int32)).reshape(-1).contiguous()
        te = self.time_embedding
        e = F.linear(sinusoidal_embedding_1d(self.freq_dim, tvals).float(), te[0].weight.float(), te[0].bias.float())
        e = F.linear(F.silu(e), te[2].weight.float(), te[2].bias.float())                                             # [G, dim]
        e0 = F.linear(F.silu(e), self.time_projection[1].weight.float(), self.time_projection[1].bias.float()).view(-1, 6, self.dim)
        # text embedding
        ctx = torch.stack([torch.cat([u, u.new_zeros(self.text_len - u.size(0), u.size(1))]) for u in context]).to(wdt)
        tx = self.text_embedding
        ctx = ops.frozen_linear(ops.gelu_tanh(ops.frozen_linear(ctx.view(-1, self.text_dim), tx[0].weight, tx[0].bias)), tx[2].weight, tx[2].bias)
        ctx = ctx.view(B, self.text_len, self.dim)
        rope = self._rope_tables((f, h, w), dev)
        xs = tok

        def run_block(blk, *a):       # the setting is scoped to this forward; a checkpointed block's recomputation (autograd thread) re-enters it here
            with ops.gemm_rows_per_call(rows_per_call):
                return blk(*a)
        for i, blk in enumerate(self.blocks):
            if self.gradient_checkpointing and torch.is_grad_enabled() and i % self.checkpoint_stride == 0:
                from torch.utils.checkpoint import checkpoint
                xs = checkpoint(run_block, blk, xs, e0, gid, B, L, rope, ctx, use_reentrant=False)
            else:
                xs = run_block(blk, xs, e0, gid, B, L, rope, ctx)
        out = self.head(xs.float(), e, gid)                                                                             # [B*L, out_dim * prod(patch)]
        # unpatchify: [f, h, w, pt, ph, pw, c] -> [c, f pt, h ph, w pw]
        c = self.out_dim
        out = out.view(B, f, h, w, pt, ph, pw, c).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, c, f * pt, h * ph, w * pw)
        return [u.float() for u in out]
