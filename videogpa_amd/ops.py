"""torch.autograd wrappers over the C-ABI HIP kernels.  PyTorch here is plumbing (device memory, streams,
autograd graph); every op below runs a hand-written gfx950 kernel through `_lib.call` and raises if it cannot."""
import contextlib as _contextlib
import contextvars as _contextvars
import ctypes
import os as _os

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype=None, contiguous=True):
    if not t.is_cuda:
        raise RuntimeError("videogpa_amd ops need GPU tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t


def _i64x3(*v):
    return (ctypes.c_int64 * 3)(*v)


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches on the launch stream (bench.py's live roofline leg).
    Disabled (None) by default: zero overhead in the product path.

    Every event pair is a packet between two kernels that keeps the next launch from overlapping the tail of the previous one: timing all
    ~1500 launches of a cfg2 step costs 1.3 % of the step (2.301 vs 2.331 s).  `full_steps` / `always`: after `full_steps` calls of next_step()
    only the kernels named in `always` (the dominant ones the roofline is computed from) are still timed; the others were sampled on the
    first steps of the timed region and run unobserved afterwards."""

    def __init__(self, full_steps=None, always=()):
        self.records = {}
        self.full_steps, self.always = full_steps, frozenset(always)
        self.step = 0
        self.steps_seen = {}

    def next_step(self):
        self.step += 1

    def run(self, name, work, fn, unit="flop"):
        if self.full_steps is not None and self.step >= self.full_steps and name not in self.always:
            fn()
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.records.setdefault(name, []).append((a, b, work, unit))
        self.steps_seen.setdefault(name, set()).add(self.step)

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in recs]
            work = sum(r[2] for r in recs) / len(recs)      # launches of one kernel can differ in size (q,k,v vs out LoRA)
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms), "work_per_launch": work,
                         "unit": recs[0][3], "steps": len(self.steps_seen[name])}
        return out


TIMER = None   # set to a KernelTimer() to time launches


def _timed(name, work, fn, unit="flop"):
    """work = algorithmic FLOPs (unit "flop", MFMA-bound kernels) or algorithmic HBM bytes = unique reads + writes
    (unit "byte", HBM-bound kernels) of the launch -- DESIGN.md section 4."""
    if TIMER is None:
        fn()
    else:
        TIMER.run(name, work, fn, unit)


# --------------------------------------------------------------------------------------------- DPO loss
class _DPOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_win, v_lose, v_win_ref, v_lose_ref, tgt_win, tgt_lose, beta, label_smoothing, loss_type, round_diff):
        ts = [v_win, v_lose, v_win_ref, v_lose_ref, tgt_win, tgt_lose]
        dt = v_win.dtype
        if dt not in _DT:
            raise RuntimeError(f"dpo_loss: unsupported dtype {dt}")
        ts = [_req(t.contiguous() if not t.is_contiguous() else t, dt) for t in ts]
        B = ts[0].shape[0]
        N = ts[0].numel() // B
        for t in ts:
            if t.shape != ts[0].shape:
                raise RuntimeError("dpo_loss: shape mismatch")
        dev = v_win.device
        out5 = torch.empty(5, dtype=torch.float32, device=dev)
        dlogit = torch.empty(B, dtype=torch.float32, device=dev)
        errs = torch.empty(B, 4, dtype=torch.float32, device=dev)
        ws_bytes = _lib.query("vgpa_dpo_loss_workspace_bytes", B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        flags = 1 if round_diff else 0
        _lib.call("vgpa_dpo_loss_fwd", *ts, B, N, N, N, N, _DT[dt], float(beta), float(label_smoothing), int(loss_type), flags,
                  out5, dlogit, errs, ws, ws_bytes, _stream())
        ctx.save_for_backward(ts[0], ts[1], ts[4], ts[5], dlogit)
        ctx.meta = (B, N, _DT[dt], float(beta), flags)
        ctx.mark_non_differentiable(errs)
        return out5[0], out5[1], out5[2], out5[3], out5[4], errs

    @staticmethod
    def backward(ctx, g_loss, g_margin, g_wr, g_lr, g_acc, g_errs):
        v_win, v_lose, tgt_win, tgt_lose, dlogit = ctx.saved_tensors
        B, N, dt, beta, flags = ctx.meta
        gw = torch.empty_like(v_win)
        gl = torch.empty_like(v_lose)
        g = g_loss.to(torch.float32).contiguous()
        _lib.call("vgpa_dpo_loss_bwd", v_win, v_lose, tgt_win, tgt_lose, B, N, N, N, N, dt, beta, flags, dlogit, g, gw, gl, _stream())
        return gw, gl, None, None, None, None, None, None, None, None


def dpo_loss(v_win, v_lose, v_win_ref, v_lose_ref, tgt_win, tgt_lose, beta=1.0, label_smoothing=0.0, loss_type="sigmoid",
             round_diff=False):
    """-> (loss, reward_margin, winner_reward, loser_reward, accuracy, errs[B,4]); only `loss` carries grad
    (the reference's logged scalars are detached by use; train/loss.py:116-121)."""
    lt = {"sigmoid": 0, "hinge": 1}.get(loss_type)
    if lt is None:
        raise ValueError(f"Unknown loss type: {loss_type}")
    return _DPOLossFn.apply(v_win, v_lose, v_win_ref, v_lose_ref, tgt_win, tgt_lose, beta, label_smoothing, lt, round_diff)


class _DPOLossPairedFn(torch.autograd.Function):
    """Paired layout: v_pair / vref_pair / tgt_pair are [B,2,...] (win, lose); no de-interleaving copies."""

    @staticmethod
    def forward(ctx, v_pair, vref_pair, tgt_pair, beta, label_smoothing, loss_type, round_diff):
        dt = v_pair.dtype
        if dt not in _DT:
            raise RuntimeError(f"dpo_loss: unsupported dtype {dt}")
        for t in (v_pair, vref_pair, tgt_pair):
            _req(t, dt)
            if t.shape != v_pair.shape or t.shape[1] != 2:
                raise RuntimeError("dpo_loss_paired: expected [B,2,...] tensors of one shape")
        B = v_pair.shape[0]
        N = v_pair.numel() // (2 * B)
        dev = v_pair.device
        out5 = torch.empty(5, dtype=torch.float32, device=dev)
        dlogit = torch.empty(B, dtype=torch.float32, device=dev)
        errs = torch.empty(B, 4, dtype=torch.float32, device=dev)
        ws_bytes = _lib.query("vgpa_dpo_loss_workspace_bytes", B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        flags = 1 if round_diff else 0
        _lib.call("vgpa_dpo_loss_fwd", v_pair[:, 0], v_pair[:, 1], vref_pair[:, 0], vref_pair[:, 1], tgt_pair[:, 0], tgt_pair[:, 1],
                  B, N, 2 * N, 2 * N, 2 * N, _DT[dt], float(beta), float(label_smoothing), int(loss_type), flags, out5, dlogit, errs, ws,
                  ws_bytes, _stream())
        ctx.save_for_backward(v_pair, tgt_pair, dlogit)
        ctx.meta = (B, N, _DT[dt], float(beta), flags)
        ctx.mark_non_differentiable(errs)
        return out5[0], out5[1], out5[2], out5[3], out5[4], errs

    @staticmethod
    def backward(ctx, g_loss, g_margin, g_wr, g_lr, g_acc, g_errs):
        v_pair, tgt_pair, dlogit = ctx.saved_tensors
        B, N, dt, beta, flags = ctx.meta
        gp = torch.empty_like(v_pair)
        g = g_loss.to(torch.float32).contiguous()
        _lib.call("vgpa_dpo_loss_bwd", v_pair[:, 0], v_pair[:, 1], tgt_pair[:, 0], tgt_pair[:, 1], B, N, 2 * N, 2 * N, 2 * N, dt, beta, flags,
                  dlogit, g, gp[:, 0], gp[:, 1], _stream())
        return gp, None, None, None, None, None, None


def dpo_loss_paired(v_pair, vref_pair, tgt_pair, beta=1.0, label_smoothing=0.0, loss_type="sigmoid", round_diff=False):
    lt = {"sigmoid": 0, "hinge": 1}.get(loss_type)
    if lt is None:
        raise ValueError(f"Unknown loss type: {loss_type}")
    return _DPOLossPairedFn.apply(v_pair, vref_pair, tgt_pair, beta, label_smoothing, lt, round_diff)


# --------------------------------------------------------------------------------------------- optimizer
def grad_norm(flat_grad, grad_scale=1.0, out=None):
    _req(flat_grad, torch.float32)
    out = torch.empty(1, dtype=torch.float32, device=flat_grad.device) if out is None else out
    ws_bytes = _lib.query("vgpa_grad_norm_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=flat_grad.device)
    _lib.call("vgpa_grad_norm", flat_grad, flat_grad.numel(), float(grad_scale), out, ws, ws_bytes, _stream())
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, max_norm=0.0, total_norm=None):
    for t in (param, grad, exp_avg, exp_avg_sq):
        _req(t, torch.float32)
    _lib.call("vgpa_adamw_step", param, grad, exp_avg, exp_avg_sq, param.numel(), float(lr), float(beta1), float(beta2), float(eps),
              float(weight_decay), int(step), float(grad_scale), float(max_norm), total_norm, _stream())


# --------------------------------------------------------------------------------------------- noise / velocity
def noise_velocity_paired(x_pair, noise, t, sqrt_abar, sqrt_1m_abar):
    """x_pair [B,2,...], noise [B,...] (shared by the pair), t [B] int64 -> (x_noisy_pair, v_target_pair)."""
    dt = x_pair.dtype
    _req(x_pair, dt), _req(noise, dt), _req(t, torch.int64), _req(sqrt_abar, torch.float32), _req(sqrt_1m_abar, torch.float32)
    B = x_pair.shape[0]
    if x_pair.shape[1] != 2 or x_pair.shape[2:] != noise.shape[1:] or noise.shape[0] != B:
        raise RuntimeError("noise_velocity_paired: expected x_pair [B,2,...] and noise [B,...]")
    N = noise.numel() // B
    xt = torch.empty_like(x_pair)
    v = torch.empty_like(x_pair)
    _lib.call("vgpa_noise_velocity_paired", x_pair, noise, t, sqrt_abar, sqrt_1m_abar, B, N, sqrt_abar.numel(), _DT[dt], xt, v, _stream())
    return xt, v


def flow_sigma(timesteps, num_train_timesteps=1000, shift=5.0):
    """get_sigma_from_timestep (train/Wan2.2-TI2V-5B/03_train.py:103-106)."""
    s = timesteps.float() / num_train_timesteps
    return shift * s / (1 + (shift - 1) * s)


def flow_noise_velocity_paired(x_pair, noise, sigma, xt_fp32=True):
    """Flow-matching noising + velocity target over the paired layout: x_pair [B,2,...], noise [B,...], sigma [B] fp32 ->
    (x_t pair [fp32 like the reference's promoted result, or latent dtype], v-target pair = noise - x)."""
    dt = x_pair.dtype
    _req(x_pair, dt), _req(noise, dt), _req(sigma, torch.float32)
    B = x_pair.shape[0]
    N = noise.numel() // B
    xt = torch.empty(x_pair.shape, dtype=torch.float32 if (xt_fp32 or dt == torch.float32) else dt, device=x_pair.device)
    v = torch.empty_like(x_pair)
    _lib.call("vgpa_flow_noise_velocity_paired", x_pair, noise, sigma, B, N, _DT[dt], 1 if xt.dtype == torch.float32 else 0, xt, v, _stream())
    return xt, v


# --------------------------------------------------------------------------------------------- AdaLN-Zero pieces
class _LNModulateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, mod, text_len, eps):
        # x [B,S,D] bf16; ln_w/ln_b fp32 [D]; mod fp32 [B,4,D] = (shift_v, 1+scale_v, shift_t, 1+scale_t) or None
        _req(x, torch.bfloat16), _req(ln_w, torch.float32), _req(ln_b, torch.float32)
        B, S, D = x.shape
        out = torch.empty_like(x)
        mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if mod is not None:
            _req(mod, torch.float32)
            sv, s1v, st, s1t = mod[:, 0], mod[:, 1], mod[:, 2], mod[:, 3]
            stride = mod.stride(0)
        else:
            sv = s1v = st = s1t = None
            stride = 0
        _lib.call("vgpa_ln_modulate_fwd", x, ln_w, ln_b, sv, s1v, st, s1t, stride, B, S, D, text_len, float(eps), out, mean, rstd, _stream())
        ctx.save_for_backward(x, mean, rstd, ln_w, mod)
        ctx.text_len = text_len
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, ln_w, mod = ctx.saved_tensors
        B, S, D = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        if mod is not None:
            s1v, s1t, stride = mod[:, 1], mod[:, 3], mod.stride(0)
        else:
            s1v = s1t = None
            stride = 0
        _lib.call("vgpa_ln_modulate_bwd", dy, x, mean, rstd, ln_w, s1v, s1t, stride, B, S, D, ctx.text_len, None, dx, _stream())
        return dx, None, None, None, None, None


def ln_modulate(x, ln_w, ln_b, mod=None, text_len=0, eps=1e-5, n_pad=0):
    return _ResidualLNFn.apply(x, None, None, ln_w, ln_b, mod, text_len, eps, n_pad, 0)[1]


def ln_modulate_recompute(x, ln_w, ln_b, mod=None, text_len=0, eps=1e-5):
    """The LN-modulate output again, outside autograd: the residual+LN kernel with no residual branch normalises the (already bf16) x exactly
    as the fused pass that produced it did, so the result is bit-identical to the n that pass returned (lean activations)."""
    B, S, D = x.shape
    n = torch.empty_like(x)
    mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    sv, s1v, st, s1t, mstride = _mod_ptrs(mod)
    _timed("ln_modulate_fwd", 4.0 * B * S * D, lambda: _lib.call("vgpa_residual_ln_fwd", x, None, None, None, 0, ln_w, ln_b, sv, s1v, st, s1t, mstride, B, S, D,
                                                                  text_len, float(eps), None, n, D, mean, rstd, _stream()), "byte")
    return n


def ln_modulate_v1(x, ln_w, ln_b, mod=None, text_len=0, eps=1e-5):
    """First-generation LN-modulate kernels (kept for A/B and as a second implementation under test)."""
    return _LNModulateFn.apply(x, ln_w, ln_b, mod, text_len, eps)


def _mod_ptrs(mod):
    if mod is None:
        return None, None, None, None, 0
    return mod[:, 0], mod[:, 1], mod[:, 2], mod[:, 3], mod.stride(0)


class _ResidualLNFn(torch.autograd.Function):
    """(x, y) -> (x_new = x + gate*y, n = LN(x_new)*alpha + beta) in one pass; backward fuses the residual-path add, the
    LN backward and the gate multiply.  y / gates may be None: plain LN-modulate of x (x_new is then x itself)."""

    @staticmethod
    def forward(ctx, x, y, gates, ln_w, ln_b, mod, text_len, eps, n_pad, dy_pad):
        """n_pad: n is returned as the first D columns of a fresh [B,S,D+n_pad] buffer (its consumer, linear_lora_ext, puts the
        LoRA down-projections into the tail and runs ONE GEMM over K = D+n_pad); dy_pad: the same for the gradient of y."""
        _req(x, torch.bfloat16), _req(ln_w, torch.float32), _req(ln_b, torch.float32)
        B, S, D = x.shape
        n = _padded_empty((B, S), D, n_pad, x.dtype, x.device) if n_pad else _empty_rows(x.shape, x.dtype, x.device)
        mean = torch.empty(B, S, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        sv, s1v, st, s1t, mstride = _mod_ptrs(mod)
        if y is not None:
            _req(y, torch.bfloat16), _req(gates, torch.float32)
            x_new = torch.empty_like(x)
            gv, gt, gstride = gates[:, 0], gates[:, 1], gates.stride(0)
        else:
            x_new, gv, gt, gstride = None, None, None, 0
        _timed("residual_ln_fwd" if y is not None else "ln_modulate_fwd", (8.0 if y is not None else 4.0) * B * S * D,
               lambda: _lib.call("vgpa_residual_ln_fwd", x, y, gv, gt, gstride, ln_w, ln_b, sv, s1v, st, s1t, mstride, B, S, D, text_len,
                                 float(eps), x_new, n, D + n_pad, mean, rstd, _stream()), "byte")
        xs = x if y is None else x_new
        ctx.save_for_backward(xs, mean, rstd, ln_w, mod, gates if y is not None else None)
        ctx.text_len = text_len
        ctx.has_y = y is not None
        ctx.dy_pad = dy_pad
        return x_new, n          # x_new is None when there is no y (plain LN-modulate)

    @staticmethod
    def backward(ctx, dx_new, dn):
        xs, mean, rstd, ln_w, mod, gates = ctx.saved_tensors
        B, S, D = xs.shape
        if dn is None:
            dn = torch.zeros_like(xs)
        dn = dn.contiguous()
        dres = None if dx_new is None else dx_new.contiguous()
        dx = torch.empty_like(xs)
        _, s1v, _, s1t, mstride = _mod_ptrs(mod)
        dyp = ctx.dy_pad
        if ctx.has_y:
            dy = _padded_empty((B, S), D, dyp, xs.dtype, xs.device) if dyp else _empty_rows(xs.shape, xs.dtype, xs.device)
            gv, gt, gstride = gates[:, 0], gates[:, 1], gates.stride(0)
        else:
            dy, gv, gt, gstride = None, None, None, 0
        passes = 3 + (1 if dres is not None else 0) + (1 if ctx.has_y else 0)     # dn, x, dx (+ dres) (+ dy)
        _timed("residual_ln_bwd" if ctx.has_y else "ln_modulate_bwd", 2.0 * passes * B * S * D,
               lambda: _lib.call("vgpa_residual_ln_bwd", dn, xs, mean, rstd, ln_w, s1v, s1t, mstride, gv, gt, gstride, dres, B, S, D,
                                 ctx.text_len, dx, dy, D + dyp, _stream()), "byte")
        return dx, dy, None, None, None, None, None, None, None, None


def residual_ln(x, y, gates, ln_w, ln_b, mod=None, text_len=0, eps=1e-5, n_pad=0, dy_pad=0):
    """-> (x + gate*y, LN-modulate of that).  One HIP pass forward, one backward."""
    return _ResidualLNFn.apply(x, y, gates, ln_w, ln_b, mod, text_len, eps, n_pad, dy_pad)


class _GateResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, gates, text_len):
        # out = x + gate[range] * y ; gates fp32 [B,2,D] = (gate_v, gate_t)
        _req(x, torch.bfloat16), _req(y, torch.bfloat16), _req(gates, torch.float32)
        B, S, D = x.shape
        out = torch.empty_like(x)
        _lib.call("vgpa_gate_residual", x, y, gates[:, 0], gates[:, 1], gates.stride(0), B, S, D, text_len, out, _stream())
        ctx.save_for_backward(gates)
        ctx.text_len = text_len
        return out

    @staticmethod
    def backward(ctx, dout):
        (gates,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, S, D = dout.shape
        dy = _empty_rows(dout.shape, dout.dtype, dout.device)
        _lib.call("vgpa_gate_residual", None, dout, gates[:, 0], gates[:, 1], gates.stride(0), B, S, D, ctx.text_len, dy, _stream())
        return dout, dy, None, None


def gate_residual(x, y, gates, text_len):
    return _GateResidualFn.apply(x, y, gates, text_len)


class _GeluTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        _req(u, torch.bfloat16)
        out = _empty_rows(u.shape, u.dtype, u.device, wide=True)      # feeds the long-K down-projection
        _timed("gelu_tanh_fwd", 4.0 * u.numel(), lambda: _lib.call("vgpa_gelu_tanh_fwd", u, u.numel(), out, _stream()), "byte")
        ctx.save_for_backward(u)
        return out

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = dy.contiguous()
        du = _empty_rows(u.shape, u.dtype, u.device, wide=True)
        _timed("gelu_tanh_bwd", 6.0 * u.numel(), lambda: _lib.call("vgpa_gelu_tanh_bwd", u, dy, u.numel(), du, _stream()), "byte")
        return du


def gelu_tanh(u):
    return _GeluTanhFn.apply(u)


# --------------------------------------------------------------------------------------------- linear (+ LoRA)
def _pad_rank(r):
    rp = 16
    while rp < r:
        rp *= 2
    return rp


def lora_down(x2, a_cat, out=None):
    """T[M,R] = x2[M,K] @ a_cat[R,K]^T on MFMA (x2 may be a column slice: row stride = x2.stride(0))."""
    M, K = x2.shape
    R = a_cat.shape[0]
    t = torch.empty(M, R, dtype=torch.bfloat16, device=x2.device) if out is None else out
    _timed("lora_down_kernel", 2.0 * M * (K + R), lambda: _lib.call("vgpa_lora_down", x2, x2.stride(0), a_cat, t, t.stride(0), M, K, R, _stream()),
           "byte")
    return t


def lora_up_add(y2, t, bw, s, accumulate=True):
    """y2[M,N] (+)= s * t[M,rp] @ bw[N,rp]^T in place (y2 / t may be column slices)."""
    M, N = y2.shape
    _timed("lora_up_add_kernel", (4.0 if accumulate else 2.0) * M * N + 2.0 * M * t.shape[1],
           lambda: _lib.call("vgpa_lora_up_add", y2, y2.stride(0), t, t.stride(0), bw, bw.stride(0), float(s), M, N, t.shape[1],
                             1 if accumulate else 0, _stream()), "byte")


def lora_grad(u, v, s=1.0):
    """fp32 G[P,Q] = s * u[M,P]^T @ v[M,Q] (contraction over tokens)."""
    M, P = u.shape
    Q = v.shape[1]
    g = torch.empty(P, Q, dtype=torch.float32, device=u.device)
    ws_bytes = _lib.query("vgpa_lora_grad_workspace_bytes", M, P, Q)       # per-row-range partials + ordered merge: bit-reproducible
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=u.device)
    _timed("lora_grad_kernel", 2.0 * M * (P + Q), lambda: _lib.call("vgpa_lora_grad_ws", u, u.stride(0), v, v.stride(0), g, Q, float(s), M, P, Q,
                                                                     ws, ws_bytes, _stream()), "byte")
    return g


# ---------------------------------------------------------------------------------------------------------------- vendor-GEMM solution selection
# The dense projections are hipBLASLt's (north_star: MFMA by hand for attention and LoRA only): 31 % of the cfg2 step.  torch asks hipBLASLt's heuristic for ONE
# solution per shape; the library holds hundreds that are valid for it.  tools/gemm_tune.py times every one of them at the shapes of the step (PyTorch TunableOp:
# the same hipblasLtMatmul call, algo chosen by index) and writes the winners to videogpa_amd/tuned/tunableop_gfx950.csv; use_tuned_gemms() makes torch pick them.
# The file is keyed by (torch, ROCm, hipBLASLt versions, gfx arch): on any other stack torch ignores it and the default heuristic runs -- never a wrong kernel.
TUNED_GEMM_FILE = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuned", "tunableop_gfx950.csv")
_TUNED_GEMMS = None          # None: undecided; else {"enabled": bool, "file": path | None, "entries": int}


def use_tuned_gemms(enabled=True, path=None):
    """Select the vendor-GEMM solutions found by tools/gemm_tune.py (PyTorch TunableOp reading a results file, tuning itself OFF: an unknown shape runs the default
    heuristic's kernel).  Called once by the trainers (config key "tuned_gemms", default True); process-wide because torch's GEMM dispatch is.  -> the decision, logged
    by the caller.  A process that already runs TunableOp its own way (PYTORCH_TUNABLEOP_ENABLED set) is left alone."""
    global _TUNED_GEMMS
    import torch.cuda.tunable as tn
    if _os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        gpu = torch.cuda.is_available()
        _TUNED_GEMMS = {"enabled": tn.is_enabled() if gpu else False, "file": tn.get_filename() if gpu else None, "entries": None,
                        "note": "PYTORCH_TUNABLEOP_* set by the caller: left as is"}
        return _TUNED_GEMMS
    path = path or TUNED_GEMM_FILE
    if not torch.cuda.is_available():
        _TUNED_GEMMS = {"enabled": False, "file": None, "entries": 0}
        return _TUNED_GEMMS
    if not enabled or not _os.path.isfile(path):
        if _TUNED_GEMMS is not None and _TUNED_GEMMS.get("enabled"):
            tn.enable(False)
        _TUNED_GEMMS = {"enabled": False, "file": None, "entries": 0}
        return _TUNED_GEMMS
    if _TUNED_GEMMS is not None and _TUNED_GEMMS.get("enabled") and _TUNED_GEMMS.get("file") == path:
        return _TUNED_GEMMS
    with open(path) as f:
        entries = sum(1 for ln in f if ln.startswith(("Gemm", "ScaledGemm")))
    tn.enable(True)
    tn.tuning_enable(False)
    tn.record_untuned_enable(False)
    tn.set_filename(path, False)
    _TUNED_GEMMS = {"enabled": True, "file": path, "entries": entries}
    return _TUNED_GEMMS


def tuned_gemms_state():
    return _TUNED_GEMMS


# dX = dY W of a frozen projection: torch.autograd issues it as an "NN" GEMM, which hipBLASLt runs 4-25 % slower at these
# shapes than the "TN" form x W^T (tools/gemm_bench.py: fused q,k,v 1.91 -> 1.44 ms, FF 2.24 -> 2.07 / 2.08 -> 1.89 ms).  The
# base weights never change during LoRA training, so a transposed copy is kept next to each weight (11 GB for CogVideoX-5B,
# made at the first backward) and dX is computed as dY (W^T)^T.
# Dispatch constants below (this one, GEMM_SPLIT_*, GEMM_ROW_SLACK, ATTN_*) are MODULE constants: a measurement tool flips them from Python before its first call
# (tools/*.py); nothing here reads the environment.  The environment carries only the three model-level settings -- VGPA_PRECISE_DELTA (default of the per-model
# precise_delta), VGPA_DP_COLLECTIVE (optim.FlatAdamW), lean activations are a trainer config key -- plus VGPA_LIB (which build of the library to load) and
# VGPA_FORCE_DIST (run the N > 1 code path on one GPU: tests).
_WT_CACHE = True


def _transposed(W):
    if not _WT_CACHE:
        return None
    c = getattr(W, "_vgpa_wt", None)
    if c is None or c[0] != W._version or c[1].device != W.device or c[1].dtype != W.dtype:
        c = (W._version, W.detach().t().contiguous())
        W._vgpa_wt = c          # lives and dies with the weight tensor
    return c[1]


# Rows per vendor-GEMM call (0 = one call).  hipBLASLt's bf16 kernels lose ~10 % at M = 36 960 (two Wan2.2 samples as one batch) against two calls of
# M = 18 480, also inside the power-limited step: cfg5 bf16 GEMMs 300 -> 272 ms per step, measured A/B in one session (profiles/r04_gemm_split_ab.txt);
# at the CogVideoX shapes (M = 35 552) the same split changes nothing or costs (feed-forward shapes: +20 ms), and the fp8 GEMMs do not care.  So the
# split is set by the model that knows its rows per sample (WanModel.forward) and applies only to row counts that are a multiple of it.
# GEMM_SPLIT_OVERRIDE (tools: an int overrides the models' choice, 0 = never split); GEMM_SPLIT_EXT_ONLY restricts the split to the LoRA-extended GEMMs.
GEMM_SPLIT_OVERRIDE = None
GEMM_SPLIT_EXT_ONLY = False
# The setting is SCOPED, not process state: a model wraps its forward in `with gemm_rows_per_call(L):` (a contextvar: other models / threads of the process
# see 0), every autograd node below records the value it ran its forward GEMM with (ctx.rows_per_call) and runs its backward GEMMs with the same one -- the
# autograd engine's threads never consult the context.  A checkpointed block re-enters the context in its recomputation (WanModel.forward).
_GEMM_ROWS = _contextvars.ContextVar("vgpa_gemm_rows_per_call", default=0)


@_contextlib.contextmanager
def gemm_rows_per_call(rows):
    """for the duration of a model's forward: vendor GEMMs whose row count is a multiple (>= 2x) of `rows` run one call per `rows` rows (0: one call)"""
    tok = _GEMM_ROWS.set(int(rows))
    try:
        yield
    finally:
        _GEMM_ROWS.reset(tok)


def current_gemm_rows():
    return int(GEMM_SPLIT_OVERRIDE) if GEMM_SPLIT_OVERRIDE is not None else _GEMM_ROWS.get()


def _slack_rows(x2):
    """rows of [M, K] storage behind x2 that _empty_rows (directly, or under a _padded_empty head view) allocated for the vendor GEMM to run over: the
    row count it may cover (>= M), or M when x2 is anything else -- a caller's view of some other buffer is never read or written past its shape."""
    base = x2 if x2._base is None else x2._base
    tag = getattr(base, "_vgpa_rows", None)
    if tag is None or x2.dim() != 2 or x2.stride(1) != 1:
        return x2.shape[0]
    rows, Mp, width = tag          # logical rows, storage rows, row width of the buffer
    if x2.shape[0] != rows or x2.stride(0) != width or x2.data_ptr() != base.data_ptr() or x2.shape[1] > width:
        return x2.shape[0]
    return Mp


def _linear_rows(x2, W, bias, ext=False, split=0):
    M = x2.shape[0]
    if GEMM_ROW_SLACK and 16384 <= M <= 65536 and x2.is_cuda and x2.dim() == 2 and x2.stride(1) == 1 and (split <= 0 or M % split):
        # the operand was allocated with row slack (_empty_rows, recognised by its tag): run the GEMM over the padded row count -- rows are independent, the
        # extra output rows are never read -- because hipBLASLt is 1-17 % faster at M = 35 840 / 36 864 than at 35 552 (tools/gemm_m_probe.py)
        room = _slack_rows(x2)
        for Mp in (gemm_rows(M, W.shape[0], W.shape[1]), gemm_rows(M)):
            if M < Mp <= room:
                xp = torch.as_strided(x2, (Mp, x2.shape[1]), x2.stride(), x2.storage_offset())
                return torch.nn.functional.linear(xp, W, bias)[:M]
    if split <= 0 or M < 2 * split or M % split or not x2.is_cuda or x2.dim() != 2 or (GEMM_SPLIT_EXT_ONLY and not ext):
        return torch.nn.functional.linear(x2, W, bias)
    out = torch.empty(M, W.shape[0], dtype=x2.dtype, device=x2.device)
    for a in range(0, M, split):
        b = a + split
        if bias is None:
            torch.mm(x2[a:b], W.t(), out=out[a:b])
        else:
            torch.addmm(bias, x2[a:b], W.t(), out=out[a:b])
    return out


def _gemm(x2, W, bias=None, ext=False, split=None):
    """x2 [M,K] @ W[N,K]^T (+ bias) through hipBLASLt (torch), timed like the hand-written kernels when bench.py asks for it.  split: rows per call
    (None: the enclosing gemm_rows_per_call context -- forward passes; backward passes hand in what their node recorded)."""
    split = current_gemm_rows() if split is None else split
    if TIMER is None:
        return _linear_rows(x2, W, bias, ext, split)
    out = []
    _timed("hipblaslt_gemm (vendor)", 2.0 * x2.shape[0] * W.shape[0] * W.shape[1], lambda: out.append(_linear_rows(x2, W, bias, ext, split)))
    return out[0]


def _frozen_dx(dy2, W, split=0):
    Wt = _transposed(W)
    return dy2 @ W if Wt is None else _gemm(dy2, Wt, split=split)


class _FrozenLinearFn(torch.autograd.Function):
    """y = x W^T + b with W, b frozen: backward is dX only, through the cached transposed weight."""

    @staticmethod
    def forward(ctx, x, W, bias):
        ctx.save_for_backward(W)
        ctx.rows_per_call = current_gemm_rows()
        return _gemm(x.reshape(-1, x.shape[-1]), W, bias, split=ctx.rows_per_call).view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (W,) = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        return _frozen_dx(dy2, W, ctx.rows_per_call).view(*dy.shape[:-1], W.shape[1]), None, None


def frozen_linear(x, W, bias):
    """F.linear for a projection whose weight / bias do not train (falls back to F.linear otherwise)."""
    if W.requires_grad or (bias is not None and bias.requires_grad):
        return torch.nn.functional.linear(x, W, bias)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return _gemm(x.reshape(-1, x.shape[-1]), W, bias).view(*x.shape[:-1], W.shape[0])
    return _FrozenLinearFn.apply(x, W, bias)


# The fused AdamW kernel writes adapter values through raw pointers (no autograd version bump): FlatAdamW.step calls
# bump_adapter_epoch() so that the bf16 copies below are refreshed exactly once per optimizer step.
ADAPTER_EPOCH = 0


def bump_adapter_epoch():
    global ADAPTER_EPOCH
    ADAPTER_EPOCH += 1


# Row slack for vendor-GEMM operands.  hipBLASLt's bf16 kernels are markedly faster when the row count is a multiple of 1024 (35 840 instead of cfg2's 35 552:
# fused q/k/v projection -6.7 %, attention-out projection -17 %, FF1 -2 %, for 0.8 % more rows; tools/gemm_m_probe.py, profiles/r04_gemm_m_probe.txt).  GEMM rows are
# independent, so the activation buffers that feed GEMMs are allocated with that many rows of STORAGE (logical shape unchanged, contents of the slack never read
# by anything but the GEMM, whose extra output rows nobody reads) and _linear_rows runs over the padded count.  Only when the padding is <= 1.25 % of the rows.
GEMM_ROW_SLACK = True


def gemm_rows(M, N=None, K=None, wide=False):
    """the row count a vendor GEMM with M real rows is run over.  Multiples of 1024 when that costs <= 1.25 % more rows; the long-K, narrow-N shape
    (feed-forward down-projection and the dX of the up-projection: N = 3072, K = 12 288) is 5 % faster still at a multiple of 4096 (36 864 for
    35 552: profiles/r04_gemm_m_probe.txt), taken when it costs <= 4 %.  wide=True: the largest count any shape may ask for (allocation size)."""
    if not GEMM_ROW_SLACK or M < 16384 or M > 65536:      # measured: gain at 35 552 rows (cfg2 / cfg3), loss at 82 052 (cfg4: 82 944 rows cost +26 ms of GEMM time per step)
        return M
    Mp = (M + 1023) // 1024 * 1024
    Mp = Mp if (Mp - M) * 80 <= M else M
    if wide or (K is not None and N is not None and K >= 8192 and N <= 4096 and K >= 3 * N):
        Mw = (M + 4095) // 4096 * 4096
        if (Mw - M) * 25 <= M:
            return Mw
    return Mp


def _empty_rows(shape, dtype, device, wide=False):
    """torch.empty(shape) whose storage has room for gemm_rows(rows) rows of the last dimension (rows = product of the leading dimensions); not a view"""
    shape = tuple(int(v) for v in shape)
    rows = 1
    for v in shape[:-1]:
        rows *= v
    Mp = gemm_rows(rows, wide=wide)
    if Mp == rows or str(device).startswith("cpu"):
        return torch.empty(shape, dtype=dtype, device=device)
    t = torch.empty(Mp * shape[-1], dtype=dtype, device=device)
    t = t.resize_(shape)
    t._vgpa_rows = (rows, Mp, shape[-1])      # what _slack_rows recognises: ONLY buffers made here are ever run past their logical row count
    return t


def _padded_empty(shape, D, pad, dtype, device):
    """A fresh [..., D + pad] buffer whose tail is zero, marked as a padded operand buffer; returns its [..., :D] head view.  The
    producers of LoRA-carrying GEMM operands (residual_ln, attention, their backwards) allocate through this; `_padded_base`
    recognises ONLY buffers made here (a caller's slice of some wider tensor is never written into)."""
    buf = _empty_rows((*shape, D + pad), dtype, device)
    buf[..., D:].zero_()          # fresh tensor, no autograd history: the reference pass's LoRA tail is zero without touching a view later
    buf._vgpa_pad = (D, pad)
    return buf[..., :D]


def _padded_base(t2, width):
    """If the 2-D tensor `t2` [M, K] is the head of a buffer made by `_padded_empty` with total width `width`, return that buffer as
    [M, width]; else None."""
    b = t2._base
    if b is None or getattr(b, "_vgpa_pad", None) != (t2.shape[-1], width - t2.shape[-1]):
        return None
    if b.shape[-1] != width or not b.is_contiguous() or b.data_ptr() != t2.data_ptr() or b.numel() != t2.shape[0] * width:
        return None
    return b.view(-1, width)


class LoraExt:
    """Operands of one LoRA-carrying projection with the adapters riding the dense GEMM as extra K:

        forward :  y  = [x | T] [W | blockdiag(s_i B_i)]^T + b        T  = x A_cat^T            (lora_down)
        backward:  dx = [dy | dT] [W^T | A_cat^T]^T                    dT_i = dy_i (s_i B_i)     (lora_down)

    so the rank-r update costs (K + R) / K of one hipBLASLt GEMM instead of a read-modify-write pass over the [M, N] output
    (the former lora_up_add kernel: 437 MB per launch at 2.3 TB/s), and it is added in the fp32 accumulator instead of after
    a bf16 rounding of y.  The frozen-reference pass runs the SAME GEMM with a zero tail, so policy and reference share every
    base partial sum bit for bit (loss = ln 2 exactly at B = 0).  bf16 operand copies are cached here and refreshed once per
    optimizer step."""

    def __init__(self):
        self.fkey = self.akey = None

    def refresh(self, W, n_slices, loras):
        act = [i for i, l in enumerate(loras) if l is not None]
        r = loras[act[0]][0].shape[0]
        rp = _pad_rank(r)
        R = len(act) * rp
        N, K = W.shape
        Dn = N // n_slices
        dt, dev = W.dtype, W.device
        fkey = (W.data_ptr(), W._version, tuple(act), rp, N, K)
        if self.fkey != fkey:
            self.W_ext = torch.zeros(N, K + R, dtype=dt, device=dev)
            self.W_ext[:, :K] = W.detach()
            self.Wt_ext = torch.zeros(K, N + R, dtype=dt, device=dev)
            self.Wt_ext[:, :N] = W.detach().t()
            self.A_cat = torch.zeros(R, K, dtype=dt, device=dev)
            self.sBt = torch.zeros(len(act), rp, Dn, dtype=dt, device=dev)
            self.fkey, self.akey = fkey, None
            self.act, self.r, self.rp, self.R, self.N, self.K, self.Dn = act, r, rp, R, N, K, Dn
        akey = (ADAPTER_EPOCH,) + tuple((loras[i][0].data_ptr(), loras[i][0]._version, loras[i][1]._version, float(loras[i][2])) for i in act)
        if self.akey != akey:
            with torch.no_grad():
                for j, i in enumerate(act):
                    A, Bm, sc = loras[i]
                    if A.shape[0] != r:
                        raise RuntimeError("adapters on one projection must share a rank")
                    if dt == torch.bfloat16 and W.is_cuda:
                        # one launch per adapter (csrc/lora.hip lora_ext_refresh_kernel) with PEFT's roundings: a = bf16(A), sB = bf16(float(bf16(B)) * s)
                        # (the adapter is cast to the activation dtype first) -- the eight strided torch copies this replaces were 1300-1900 tiny kernels per step
                        _lib.call("vgpa_lora_ext_refresh", A.detach().float().contiguous(), Bm.detach().float().contiguous(), float(sc), r, K, Dn,
                                  self.A_cat[j * rp:], self.Wt_ext[:, N + j * rp:], N + R, self.W_ext[i * Dn:, K + j * rp:], K + R, self.sBt[j], _stream())
                        continue
                    self.A_cat[j * rp:j * rp + r] = A.to(dt)
                    sB = (Bm.to(dt).float() * sc).to(dt)                     # PEFT casts the adapter to the activation dtype first
                    self.W_ext[i * Dn:(i + 1) * Dn, K + j * rp:K + j * rp + r] = sB
                    self.sBt[j, :r] = sB.t()
                    self.Wt_ext[:, N + j * rp:N + j * rp + r] = self.A_cat[j * rp:j * rp + r].t()
            self.akey = akey
        return self


# ---- fp8 (OCP e4m3) form of a frozen projection: the "fp8 MFMA path" of BASELINE.json configs[4] (Wan2.2-TI2V-5B) ------------------
def quant_fp8_rows(x2):
    """bf16 [M, K] (row stride free) -> (e4m3 [M, K], fp32 scale [M, 1]): per-row dynamic scaling, csrc/fp8.hip"""
    M, K = x2.shape
    q = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=x2.device)
    sc = torch.empty(M, 1, dtype=torch.float32, device=x2.device)
    _timed("quant_fp8_rows", 3.0 * M * K, lambda: _lib.call("vgpa_quant_fp8_rows", x2, x2.stride(0), q, sc, M, K, _stream()), "byte")
    return q, sc


def gelu_tanh_fwd_q8(u2):
    """bf16 [M, K] -> (e4m3(gelu_tanh(u)) [M, K], scale [M, 1]): the activation written as the fp8 GEMM operand by its producer"""
    M, K = u2.shape
    q = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=u2.device)
    sc = torch.empty(M, 1, dtype=torch.float32, device=u2.device)
    _timed("gelu_tanh_fwd_q8", 3.0 * M * K, lambda: _lib.call("vgpa_gelu_tanh_fwd_q8", u2, M, K, q, sc, _stream()), "byte")
    return q, sc


def gelu_tanh_bwd_q8(u2, dy2):
    """(u, dy) bf16 [M, K] -> (e4m3(dy * gelu_tanh'(u)) [M, K], scale [M, 1])"""
    M, K = u2.shape
    q = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=u2.device)
    sc = torch.empty(M, 1, dtype=torch.float32, device=u2.device)
    _timed("gelu_tanh_bwd_q8", 5.0 * M * K, lambda: _lib.call("vgpa_gelu_tanh_bwd_q8", u2, dy2, M, K, q, sc, _stream()), "byte")
    return q, sc


class Fp8Weight:
    """e4m3 copies of a frozen weight W [N, K], one scale per output row, for y = x W^T, and of W^T for dx = dy W; made once
    (the base weights never change in LoRA training) and kept with the weight tensor."""

    def __init__(self, W):
        self.version = W._version
        self.q, self.s = self._rows(W.detach())                 # [N, K], [1, N]
        self.qt, self.st = self._rows(W.detach().t().contiguous())   # [K, N], [1, K]

    @staticmethod
    def _rows(w):
        fmax = torch.finfo(torch.float8_e4m3fn).max
        amax = w.float().abs().amax(dim=1, keepdim=True)
        sc = torch.where(amax > 0, amax / fmax, torch.ones_like(amax))
        return (w.float() / sc).clamp(-fmax, fmax).to(torch.float8_e4m3fn), sc.t().contiguous()

    @staticmethod
    def of(W):
        c = getattr(W, "_vgpa_fp8", None)
        if c is None or c.version != W._version or c.q.device != W.device:
            c = Fp8Weight(W)
            W._vgpa_fp8 = c
        return c


def _fp8_gemm(xq, sx, wq, sw, bias=None):
    """bf16 [M, N] = (xq * sx) (wq * sw)^T through the vendor fp8 GEMM (hipBLASLt); xq [M, K], wq [N, K] e4m3, sx [M, 1], sw [1, N]"""
    M, N, K = xq.shape[0], wq.shape[0], xq.shape[1]
    out = []

    _timed("hipblaslt_gemm_fp8 (vendor)", 2.0 * M * N * K,
           lambda: out.append(torch._scaled_mm(xq, wq.t(), scale_a=sx, scale_b=sw, bias=bias, out_dtype=torch.bfloat16)))
    return out[0]


class _Fp8FrozenLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias):
        w8 = Fp8Weight.of(W)
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        xq, sx = quant_fp8_rows(x2)
        ctx.w8 = w8
        return _fp8_gemm(xq, sx, w8.q, w8.s, bias).view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        w8 = ctx.w8
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        dq, sd = quant_fp8_rows(dy2)
        return _fp8_gemm(dq, sd, w8.qt, w8.st).view(*dy.shape[:-1], w8.qt.shape[0]), None, None


def frozen_linear_fp8(x, W, bias):
    """frozen_linear with e4m3 operands (per-row dynamic activation scales, per-output-row weight scales), bf16 result"""
    if W.requires_grad or (bias is not None and bias.requires_grad):
        raise RuntimeError("videogpa_amd: the fp8 path is for frozen projections only")
    if x.dtype != torch.bfloat16:
        raise TypeError("frozen_linear_fp8: bf16 activations")
    return _Fp8FrozenLinearFn.apply(x, W, bias)


class _LinearLoraExtFn(torch.autograd.Function):
    """y = x W^T + b (+ LoRA when `enabled`), W / b frozen; see LoraExt.  x / dy are used in place when they are the heads of
    padded buffers (produced by ops.residual_ln / qknorm_attention with n_pad / o_pad / grad pads), copied into one otherwise.
    x_recompute: (fn, tensors) with fn(*tensors) returning x [M, K] again (bit-identical) in the backward; the forward then keeps only the [M, R] LoRA
    down-projections instead of the whole [M, K + R] operand ("lean activations": x is a cheap function of a tensor that is saved anyway)."""

    @staticmethod
    def forward(ctx, x, bias, ext, enabled, scalings, x_recompute, *AB):
        K, R, N = ext.K, ext.R, ext.N
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        x_ext = _padded_base(x2, K + R)
        if x_ext is None:
            x_ext = torch.zeros(M, K + R, dtype=x.dtype, device=x.device) if not enabled else torch.empty(M, K + R, dtype=x.dtype, device=x.device)
            x_ext[:, :K].copy_(x2)
        xv, tv = x_ext[:, :K], x_ext[:, K:]
        if enabled:
            lora_down(xv, ext.A_cat, out=tv)       # raw kernel write into the tail (no autograd version bump on the producer's buffer)
        # disabled (reference pass): the tail of a `_padded_empty` buffer is zero since its allocation
        ctx.rows_per_call = current_gemm_rows()
        y = _gemm(x_ext, ext.W_ext, bias, ext=True, split=ctx.rows_per_call)
        if x_recompute is not None and enabled:
            fn, srcs = x_recompute
            ctx.save_for_backward(tv.contiguous(), *srcs)     # the recompute sources go through autograd's saved-tensor checks (in-place version, lifetime)
            ctx.x_fn = fn
        else:
            ctx.save_for_backward(x_ext)
            ctx.x_fn = None
        ctx.ext, ctx.enabled, ctx.xshape, ctx.scalings = ext, enabled, x.shape, scalings
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        ext, enabled = ctx.ext, ctx.enabled
        saved, *srcs = ctx.saved_tensors
        K, R, N, Dn, rp, r, act = ext.K, ext.R, ext.N, ext.Dn, ext.rp, ext.r, ext.act
        dy2 = dy.reshape(-1, N)
        M = dy2.shape[0]
        dy_ext = _padded_base(dy2, N + R) if dy2.stride(1) == 1 else None
        if dy_ext is None:
            dy_ext = torch.zeros(M, N + R, dtype=dy.dtype, device=dy.device) if not enabled else torch.empty(M, N + R, dtype=dy.dtype, device=dy.device)
            dy_ext[:, :N].copy_(dy2)
        if enabled:
            for j, i in enumerate(act):
                lora_down(dy_ext[:, i * Dn:(i + 1) * Dn], ext.sBt[j], out=dy_ext[:, N + j * rp:N + (j + 1) * rp])      # dT_i = dy_i (s B_i)
        dx = _gemm(dy_ext, ext.Wt_ext, ext=True, split=ctx.rows_per_call)                                                       # dy W + dT A
        out_grads = [None] * len(ctx.needs_input_grad[6:])
        if enabled:
            if ctx.x_fn is not None:
                t_all, xh = saved, ctx.x_fn(*srcs).reshape(M, K)     # [M, R] kept, x made again
            else:
                t_all, xh = saved[:, K:], saved[:, :K]
            for j, i in enumerate(act):
                # dB_i = s dy_i^T T_i (T_i = x A_i^T sits in the tail of x_ext); the scale is applied to the fp32 result
                out_grads[2 * i + 1] = lora_grad(dy_ext[:, i * Dn:(i + 1) * Dn], t_all[:, j * rp:(j + 1) * rp], ctx.scalings[j])[:, :r]
            dA = lora_grad(dy_ext[:, N:], xh)                                                                        # dT^T x, all adapters at once
            for j, i in enumerate(act):
                out_grads[2 * i] = dA[j * rp:j * rp + r]
        return (dx.view(ctx.xshape), None, None, None, None, None, *out_grads)


def linear_lora_ext(x, W, bias, ext, loras, enabled=True, x_recompute=None):
    """loras: list (one per equal output slice) of None or (A [r,in] fp32, B [out_i,r] fp32, scaling) -- the adapters that EXIST
    on this projection; `enabled` False = the reference pass (same GEMM, zero LoRA tail).  x_recompute: see _LinearLoraExtFn."""
    if W.requires_grad or (bias is not None and bias.requires_grad):
        raise RuntimeError("videogpa_amd: base weights are frozen on this path (LoRA-only training, as in the reference)")
    ext.refresh(W, len(loras), loras)
    flat = []
    for l in loras:
        flat += [None, None] if l is None else [l[0], l[1]]
    return _LinearLoraExtFn.apply(x, bias, ext, bool(enabled), tuple(float(loras[i][2]) for i in ext.act), x_recompute, *flat)


def linear_lora(x, W, bias, loras, ext=None):
    """loras: list (one per equal output slice) of None or (A [r,in] fp32, B [out_i,r] fp32, scaling)."""
    if all(l is None for l in loras):
        return frozen_linear(x, W, bias)
    if ext is None:
        ext = getattr(W, "_vgpa_lora_ext", None)
        if ext is None:
            ext = LoraExt()
            W._vgpa_lora_ext = ext          # lives with the frozen weight
    return linear_lora_ext(x, W, bias, ext, loras, True)


# --------------------------------------------------------------------------------------------- attention
def _bhs_strides(t):
    """element strides {batch, head, token} of a [B,H,S,64] view"""
    assert t.stride(3) == 1
    return _i64x3(t.stride(0), t.stride(1), t.stride(2))


LOG2E = 1.4426950408889634
# backward attention: "split" (default) = deterministic dK/dV kernel + dQ kernel (7 matrix products per score block);
# "fused" = one kernel with dQ by fp32 atomics (5 products).  Measured on MI355X at the headline shape the fused form
# is SLOWER (46.6 ms vs 26.5 ms per layer): its 60 GB of dQ atomics per launch run at ~2.6 TB/s (23.9 ms without them).
# It lives in tools/variants/ (variant builds export vgpa_attn_bwd_fused; ops.ATTN_BWD_FUSED = True selects it there).
ATTN_BWD_FUSED = False        # True needs a variant build (tools/build_variant.sh: vgpa_attn_bwd_fused lives in tools/variants/)
# tail-round treatment of the attention launches (vgpa_attn_*_ws split_mode): -1 automatic (default), 0 off
ATTN_SPLIT_MODE = -1
# Attention kernels come from the "w1" family (csrc/attention_w1.hip: one wave per SIMD, LDS-DMA rings, generated hand-scheduled
# main loops).  ATTN_W1 = subset of {fwd, dq, dkv} selects which (default all three; empty = the 2-waves-per-SIMD kernels of attention.hip,
# which stay in the library as the online-softmax / redo path of the forward and for A/B runs: tools set ops.ATTN_W1).
ATTN_W1 = {"fwd", "dq", "dkv"}
# "Precise delta": the attention forward also stores what the bf16 rounding of its output dropped, and the backward forms delta = rowsum(dO o O) from the
# completed output.  delta stands for rowsum(P o dP); formed from the bf16 O alone (what every flash-attention backward, torch's included, does) each row's dS
# stops summing to zero and dQ picks up a coherent error -d(delta_i) sum_j P_ij K_j that swamps q / k gradients which are small by cancellation (37-87 % of
# the last block's to_q / to_k adapter gradients at BASELINE configs[0] width: profiles/r04_cfg1_round_diag_*.json).  Two forms of the residual tensor:
#   "int8" (default): one byte per output element = eight further mantissa bits (csrc/common.h res8) -- O to 2^-17; every attention path has it (head_dim 64
#                     and 128, e4m3 forward, short-key kernel, lean activations);
#   "bf16"          : bf16(O_fp32 - bf16(O)), twice the bytes, O to 2^-18 (the round-4 form; head_dim 64 only);
#   None / "off"    : the textbook backward.
# It is NOT process state: models carry `precise_delta` (constructor default = precise_delta_default(), i.e. VGPA_PRECISE_DELTA in the environment: 0 | off |
# bf16 | int8) and hand it to qknorm_attention / attention128 per call; the autograd node keeps what its forward used.
def precise_delta_default():
    v = _os.environ.get("VGPA_PRECISE_DELTA", "int8").lower()
    if v in ("0", "off", "none", "false"):
        return None
    if v in ("1", "int8", "res8", "true"):
        return "int8"
    if v == "bf16":
        return "bf16"
    raise ValueError(f"VGPA_PRECISE_DELTA={v!r}: expected 0 | off | int8 | bf16")


def _res_kind(o_res):
    if o_res is None:
        return 0
    if o_res.dtype == torch.bfloat16:
        return 1
    if o_res.dtype == torch.uint8:
        return 2
    raise TypeError(f"o_res: bf16 (rounding residual) or uint8 (eight further mantissa bits), got {o_res.dtype}")


def res8_decode(o, o_res8):
    """fp32 value of a bf16 tensor `o` completed by its res8 bytes (csrc/common.h): o + (byte - 128) * 2^(E - 142), E = biased exponent of o (tests)"""
    E = (o.contiguous().view(torch.int16).to(torch.int32) >> 7) & 0xFF
    sc = torch.where(E > 15, torch.ldexp(torch.ones((), device=o.device), E - 142), torch.zeros((), device=o.device))
    return o.float() + (o_res8.float() - 128.0) * sc


def prescale_q(q, scale=None):
    """The attention kernels take q * scale * log2(e) (one bf16 rounding).  The fused path gets it from the QK-norm
    kernel; this helper is for callers holding a plain q."""
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    return (q.float() * (scale * LOG2E)).to(q.dtype)


class AttnFwdPolicy:
    """Which forward one attention layer runs, decided from what the kernel itself reports.  "bound": the w1 kernel (softmax shifted per row by
    min(Cauchy-Schwarz bound, sampled row maximum + 64) instead of a running maximum; strips with a row it cannot represent -- the true maximum more than ~176 log2
    units above the maximum over 64 sampled keys -- are flagged and redone inside the same call by the online-softmax kernel: results never depend on it, time does).
    "online": every strip on the online-softmax kernel (vgpa_attn_fwd_online_res).  On the data that gets strips flagged (near one-hot rows with a huge score range)
    the online kernel itself runs 2.3-2.5 x its usual time (its running maximum keeps moving: measured 17.5 ms against 7.2, profiles/r06b_attn_trained_like.txt), so
    the switch only pays once MORE THAN HALF the strips would be swept twice: SWITCH = 0.5.  The flags of a layer's first CHECKS calls (and of every RECHECK-th call
    after) are read back -- one host sync each -- and the layer switches for good above SWITCH.  `fixed` pins the mode (tools, tests)."""
    SWITCH, CHECKS, RECHECK = 0.5, 2, 1024

    def __init__(self, mode="bound", fixed=False):
        self.mode, self.fixed = mode, fixed
        self.calls = 0
        self.redo_fraction = None          # of the most recent observed call
        self.switched_at = None

    def wants_flags(self):
        return self.mode == "bound" and not self.fixed and (self.calls < self.CHECKS or self.calls % self.RECHECK == 0)

    def observe(self, fraction):
        self.redo_fraction = fraction
        if not self.fixed and fraction > self.SWITCH:
            self.mode, self.switched_at = "online", self.calls


def attention_redo_fraction(ws, B, H, S):
    """fraction of the (batch, head, 256-row strip) tasks the last vgpa_attn_fwd_w1* call on workspace `ws` flagged and redid (synchronises)"""
    tasks = B * H * ((S + 255) // 256)
    flags = ws[:4 * (B * H + tasks)].view(torch.int32)[B * H:]
    return float((flags != 0).float().mean())


def attention_fwd_raw(q, k, v, scale=None, q_prescaled=False, split_mode=None, o_pad=0, o_res=None, policy=None):
    """q,k,v: bf16 [B,H,S,64] views (any batch/head/token strides).  -> o [B,S,H*64] bf16, lse2 [B,H,S] fp32.
    o_res: optional [B,S,H*64] buffer, bf16 or uint8, that receives what the output's bf16 rounding dropped (w1 forward only; see "Precise delta").
    split_mode: -1 lets the launcher cut the tasks of a mostly empty last scheduling round into key-range chunks,
    0 forbids it, k >= 2 forces k chunks for every task (tests).  o_pad: o is the head of a [B,S,H*64+o_pad] buffer (the
    output projection's LoRA tail, see LoraExt)."""
    B, H, S, Dh = q.shape
    scale = Dh ** -0.5 if scale is None else scale
    if not q_prescaled:
        q = prescale_q(q, scale)
    o = _padded_empty((B, S), H * Dh, o_pad, torch.bfloat16, q.device) if o_pad else torch.empty(B, S, H * Dh, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=q.device)
    ov = o.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3)
    split_mode = ATTN_SPLIT_MODE if split_mode is None else split_mode
    if "fwd" in ATTN_W1:
        ws_bytes = _lib.query("vgpa_attn_fwd_w1_workspace_bytes", B, H, S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        rv = None if o_res is None else o_res.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3)
        entry = "vgpa_attn_fwd_online_res" if (policy is not None and policy.mode == "online") else "vgpa_attn_fwd_w1_res"
        _timed("attn_fwd_kernel", 4.0 * S * S * Dh * B * H, lambda: _lib.call(
            entry, q, k, v, o, o_res, _res_kind(o_res), lse, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(ov), None if rv is None else _bhs_strides(rv),
            B, H, S, Dh, float(scale), int(split_mode), ws, ws_bytes, _stream()))
        if policy is not None:
            if policy.wants_flags() and not torch.cuda.is_current_stream_capturing():
                policy.observe(attention_redo_fraction(ws, B, H, S))
            policy.calls += 1
        return o, lse
    if o_res is not None:
        raise RuntimeError("attention_fwd_raw: o_res needs the w1 forward (ops.ATTN_W1 includes \"fwd\")")
    ws_bytes = _lib.query("vgpa_attn_fwd_workspace_bytes", B, H, S) if split_mode != 0 else 0
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
    _timed("attn_fwd_kernel", 4.0 * S * S * Dh * B * H, lambda: _lib.call(
        "vgpa_attn_fwd_ws", q, k, v, o, lse, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(ov), B, H, S, Dh,
        float(scale), int(split_mode), ws if ws_bytes else None, ws_bytes, _stream()))
    return o, lse


def attention_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, scale=None, q_prescaled=False, split_mode=None, o_res=None):
    """All [B,H,S,64] bf16 views; writes dq, dk, dv in place.  Three launches: delta, dK/dV, dQ.  o_res ([B,H,S,64] view, bf16 or uint8): what the forward
    kept of the output beyond its bf16 rounding; delta is then formed from the completed output ("Precise delta").
    Algorithmic FLOPs (backward = 2 x forward): dK/dV kernel carries dV, dP, dK = 6 S^2 d; dQ kernel 2 S^2 d
    (the S = QK^T recomputes in both kernels and the second dP are overhead, not counted)."""
    B, H, S, Dh = q.shape
    scale = Dh ** -0.5 if scale is None else scale
    if not q_prescaled:
        q = prescale_q(q, scale)
    delta = torch.empty(B, H, S, dtype=torch.float32, device=q.device)
    st = _stream()
    w1_dkv = "dkv" in ATTN_W1 and not ATTN_BWD_FUSED
    if w1_dkv:     # one pass: delta + the {-lse2, -delta} planes the w1 dK/dV kernel streams
        stats = torch.empty(B, H, 2, S, dtype=torch.float32, device=q.device)
        _timed("attn_delta_kernel", (4.0 + (0 if o_res is None else o_res.element_size())) * B * H * S * Dh, lambda: _lib.call(
            "vgpa_attn_bwd_prep_w1_res", o, o_res, _res_kind(o_res), do, lse, _bhs_strides(o), None if o_res is None else _bhs_strides(o_res), _bhs_strides(do),
            delta, stats, B, H, S, Dh, st), "byte")
    else:
        _timed("attn_delta_kernel", (4.0 + (0 if o_res is None else o_res.element_size())) * B * H * S * Dh, lambda: _lib.call(
            "vgpa_attn_bwd_delta_res", o, o_res, _res_kind(o_res), do, _bhs_strides(o), None if o_res is None else _bhs_strides(o_res), _bhs_strides(do), delta,
            B, H, S, Dh, st), "byte")
    if ATTN_BWD_FUSED:
        dq32 = torch.zeros(B, H, S, Dh, dtype=torch.float32, device=q.device)
        _timed("attn_bwd_fused_kernel", 8.0 * S * S * Dh * B * H, lambda: _lib.call(
            "vgpa_attn_bwd_fused", q, k, v, do, lse, delta, dq32, dk, dv, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(do),
            _bhs_strides(dk), _bhs_strides(dv), B, H, S, Dh, float(scale), st))
        dq.copy_(dq32)
        return
    split_mode = ATTN_SPLIT_MODE if split_mode is None else split_mode
    ws_bytes = _lib.query("vgpa_attn_bwd_split_workspace_bytes", B, H, S) if split_mode != 0 else 0
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
    wsp = ws if ws_bytes else None
    if w1_dkv:
        _timed("attn_bwd_dkv_kernel", 6.0 * S * S * Dh * B * H, lambda: _lib.call(
            "vgpa_attn_bwd_dkv_w1", q, k, v, do, stats, dk, dv, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(do),
            _bhs_strides(dk), _bhs_strides(dv), B, H, S, Dh, float(scale), int(split_mode), wsp, ws_bytes, st))
    else:
        _timed("attn_bwd_dkv_kernel", 6.0 * S * S * Dh * B * H, lambda: _lib.call(
            "vgpa_attn_bwd_dkv_ws", q, k, v, do, lse, delta, dk, dv, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(do),
            _bhs_strides(dk), _bhs_strides(dv), B, H, S, Dh, float(scale), int(split_mode), wsp, ws_bytes, st))
    _timed("attn_bwd_dq_kernel", 2.0 * S * S * Dh * B * H, lambda: _lib.call(
        "vgpa_attn_bwd_dq_w1" if "dq" in ATTN_W1 else "vgpa_attn_bwd_dq_ws", q, k, v, do, lse, delta, dq, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(do),
        _bhs_strides(dq), B, H, S, Dh, float(scale), int(split_mode), wsp, ws_bytes, st))


ATTN128_W1 = True    # False: the compiler-scheduled hd128 forward everywhere (tools)


ATTN128_F8_MIN_KEYS = 1024      # below this the e4m3 forward's prep passes and pipeline fill do not pay (cross-attention over 512 text tokens stays bf16)


def attention128_uses_f8(f8, Skv):
    """does attention128_fwd_raw(f8=f8) run the e4m3 kernel for a sweep of Skv keys?"""
    return bool(f8) and Skv >= ATTN128_F8_MIN_KEYS


class F8AttnPolicy:
    """Whether one head_dim-128 self-attention layer runs its forward on e4m3 operands (vgpa_attn128_fwd_f8) or in bf16, decided ONCE from the data of its first call.
    The e4m3 forward's time no longer depends on the data (sampled shift), its ACCURACY does: q and k each carry a 2^-4 relative rounding per element, so a score
    q.k (log2 units) is off by ~ ERR_PER_BOUND x |q| |k| scale log2(e) rms -- measured 0.08 log2 units at a row bound of 29 (unit QK-norm gains: the ~5 % per weight of
    the kernel's tests) and 0.73 at a bound of 180 (gains of 2.5: every weight off by a factor 1.7 rms; tools/f8_sharp_diag.py).  `threshold` (log2 units of rms score
    error, default 0.5) is where the layer goes to the bf16 forward for good (6.3 ms per launch at the cfg5 shape against 4.1).  fixed=True pins the mode
    (WanModel.enable_fp8(attention=True): the e4m3 kernel whatever the data).  The decision is taken at the layer's FIRST call only -- the frozen-reference pass of a
    DPO step -- so that reference and policy pass always run the same forward (loss = ln 2 exactly at B = 0); reset() re-arms it."""
    ERR_PER_BOUND = 0.004

    def __init__(self, mode="f8", fixed=False, threshold=0.5):
        self.mode, self.fixed, self.threshold = mode, fixed, float(threshold)
        self.decided = False
        self.estimated_score_error = None

    def reset(self):
        self.decided = False

    @staticmethod
    def score_error_estimate(q, k, H, scale):
        """q [B, Lq, H*d], k [B, Lk, H*d] (token-major, as the model holds them) -> estimated rms error of an e4m3 score in log2 units, worst (batch, head)"""
        B, Lq, D = q.shape
        d = D // H
        qn = q.view(B, Lq, H, d).float().pow(2).sum(-1).amax(dim=1).sqrt()          # [B, H]: the longest query row per head
        kn = k.view(B, k.shape[1], H, d).float().pow(2).sum(-1).amax(dim=1).sqrt()
        return float((qn * kn).amax()) * float(scale) * LOG2E * F8AttnPolicy.ERR_PER_BOUND

    def use_f8(self, q, k, H, scale):
        if not self.fixed and not self.decided and self.mode == "f8":
            self.estimated_score_error = self.score_error_estimate(q, k, H, scale)      # one host sync, once per layer
            if self.estimated_score_error > self.threshold:
                self.mode = "bf16"
        self.decided = True
        return self.mode == "f8"


def attention128_f8_redo_fraction(ws, B, H, Sq):
    """fraction of the (batch, head, 256-row strip) tasks the last vgpa_attn128_fwd_f8 call on workspace `ws` flagged and redid in bf16 (synchronises)"""
    tasks = B * H * ((Sq + 255) // 256)
    flags = ws[:4 * (5 * B * H + tasks)].view(torch.int32)[5 * B * H:]
    return float((flags != 0).float().mean())


def attention128_fwd_raw(q, k, v, scale, o_pad=0, f8=False, o_res8=None, deq=None, report=None):
    """q [B,H,Sq,128], k / v [B,H,Skv,128] bf16 views (any batch / head / token strides, last dim contiguous) -> (o, lse2 [B,H,Sq] fp32).
    f8: the e4m3 forward (csrc/attention_hd128.hip, vgpa_attn128_fwd_f8) for sweeps of at least ATTN128_F8_MIN_KEYS keys.
    deq (e4m3 forward only): three bf16 buffers [B, Sq, H*128], [B, Skv, H*128], [B, Skv, H*128] that receive the operands the e4m3 products really ran
    on, dequantised EXACTLY (q: times scale * log2 e) -- what the backward of THIS forward runs on (attention128_bwd_raw(..., q_prescaled=True) on them recomputes
    the forward's own scores bit for bit).
    o_res8: optional uint8 [B, Sq, H*128] buffer that receives eight further mantissa bits of every output value ("Precise delta").
    report: optional dict; the e4m3 forward stores "redo_fraction" (strips it flagged and redid in bf16) and "strip_flags" (bool [B, H, strips of 256 rows])
    there -- one host sync.
    o is a [B,H,Sq,128] view of token-major storage [B, Sq, H*128 (+ o_pad)]: the caller's flatten to [B*Sq, H*128] is free, and with
    o_pad it is the head of a `_padded_empty` buffer (the output projection's LoRA tail, see LoraExt)."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    assert D == 128 and k.shape == (B, H, Skv, 128) and v.shape == k.shape and q.dtype == k.dtype == v.dtype == torch.bfloat16
    o2 = _padded_empty((B, Sq), H * D, o_pad, torch.bfloat16, q.device) if o_pad else torch.empty(B, Sq, H * D, dtype=torch.bfloat16, device=q.device)
    o = o2.unflatten(-1, (H, D)).permute(0, 2, 1, 3)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device)
    rv = rs = None
    if o_res8 is not None:
        if o_res8.dtype != torch.uint8 or o_res8.shape != (B, Sq, H * D) or not o_res8.is_contiguous():
            raise TypeError("attention128_fwd_raw: o_res8 is a contiguous uint8 [B, Sq, H*128] buffer")
        rv = o_res8.unflatten(-1, (H, D)).permute(0, 2, 1, 3)
        rs = _bhs_strides(rv)
    if attention128_uses_f8(f8, Skv):
        ws_bytes = _lib.query("vgpa_attn128_fwd_f8_workspace_bytes", B, H, Sq, Skv)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        dv_ = [None] * 3
        if deq is not None:
            for t, S_ in zip(deq, (Sq, Skv, Skv)):
                if t.dtype != torch.bfloat16 or t.shape != (B, S_, H * D) or not t.is_contiguous():
                    raise TypeError("attention128_fwd_raw: deq = three contiguous bf16 [B, S, H*128] buffers (q, k, v)")
            dv_ = [t.unflatten(-1, (H, D)).permute(0, 2, 1, 3) for t in deq]
        ds_ = [None if t is None else _bhs_strides(t) for t in dv_]
        _timed("attn128_fwd_f8", 4.0 * B * H * Sq * Skv * D, lambda: _lib.call(
            "vgpa_attn128_fwd_f8", q, k, v, o, lse, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(o), rv, rs, dv_[0], dv_[1], dv_[2],
            ds_[0], ds_[1], ds_[2], B, H, Sq, Skv, float(scale), ws, ws_bytes, _stream()))
        if report is not None:          # tools / tests: what fraction of the strips the e4m3 kernel handed to the bf16 redo pass
            report["redo_fraction"] = attention128_f8_redo_fraction(ws, B, H, Sq)
            report["strip_flags"] = (ws[:4 * (5 * B * H + B * H * ((Sq + 255) // 256))].view(torch.int32)[5 * B * H:] != 0).view(B, H, -1).clone()
        return o, lse
    if deq is not None:
        raise ValueError("attention128_fwd_raw: deq buffers are written by the e4m3 forward only (f8=True and at least ATTN128_F8_MIN_KEYS keys)")
    ws_bytes = _lib.query("vgpa_attn128_fwd_workspace_bytes", B, H, Sq) if ATTN128_W1 else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device) if ws_bytes else None
    _timed("attn128_fwd" if Skv >= 1024 else "attn128_fwd (short keys)", 4.0 * B * H * Sq * Skv * D, lambda: _lib.call(
        "vgpa_attn128_fwd", q, k, v, o, lse, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(o), rv, rs, B, H, Sq, Skv, float(scale),
        ws, ws_bytes, _stream()))
    return o, lse


def attention128_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, scale, o_res8=None, q_prescaled=False):
    """all [B,H,S,128] bf16 views; writes dq, dk, dv in place (they may be strided slices of a fused gradient buffer).  o_res8 (uint8 [B, Sq, H*128], as the
    forward wrote it): delta = rowsum(dO o O) is formed from the output completed by those eight further mantissa bits ("Precise delta").
    q_prescaled: q is the e4m3 forward's q_deq (the query times scale * log2 e, exact): vgpa_attn128_bwd_prescaled -- scores bit for bit the forward's, dq still
    the gradient w.r.t. the unscaled query."""
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    rv = None if o_res8 is None else o_res8.unflatten(-1, (H, D)).permute(0, 2, 1, 3)
    ws_bytes = _lib.query("vgpa_attn128_bwd_workspace_bytes", B, H, Sq)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    # algorithmic work as SURVEY 8d counts it: backward = 2 x forward (dV, dP, dK, dQ); the S = QK^T recomputes of the split kernels are overhead
    _timed("attn128_bwd" if Skv >= 1024 else "attn128_bwd (short keys)", 8.0 * B * H * Sq * Skv * D, lambda: _lib.call(
        "vgpa_attn128_bwd_prescaled" if q_prescaled else "vgpa_attn128_bwd", q, k, v, o, do, lse, dq, dk, dv, _bhs_strides(q), _bhs_strides(k), _bhs_strides(v), _bhs_strides(o), _bhs_strides(do),
        _bhs_strides(dq), _bhs_strides(dk), _bhs_strides(dv), rv, None if rv is None else _bhs_strides(rv), B, H, Sq, Skv, float(scale),
        -1 if ATTN128_W1 else 0, ws, ws_bytes, _stream()))


class _Attention128Fn(torch.autograd.Function):
    """softmax(scale q k^T) v for head_dim 128, query and key lengths free (csrc/attention_hd128.hip): Wan2.2's self- and cross-attention"""

    @staticmethod
    def forward(ctx, q, k, v, scale, o_pad, precise_delta):
        q, k, v = (t if t.stride(3) == 1 else t.contiguous() for t in (q, k, v))
        o_res8 = None
        if precise_delta and any(ctx.needs_input_grad[:3]):
            o_res8 = torch.empty(q.shape[0], q.shape[2], q.shape[1] * q.shape[3], dtype=torch.uint8, device=q.device)
        o, lse = attention128_fwd_raw(q, k, v, scale, o_pad, o_res8=o_res8)
        ctx.save_for_backward(q, k, v, o, lse, o_res8)
        ctx.scale = float(scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, o_res8 = ctx.saved_tensors
        B, H, Sq, D = q.shape
        Skv = k.shape[2]
        do = do if do.stride(3) == 1 else do.contiguous()
        dq = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device=q.device).permute(0, 2, 1, 3)      # token-major, like the projections' outputs
        dk = torch.empty(B, Skv, H, D, dtype=torch.bfloat16, device=q.device).permute(0, 2, 1, 3)
        dv = torch.empty(B, Skv, H, D, dtype=torch.bfloat16, device=q.device).permute(0, 2, 1, 3)
        attention128_bwd_raw(q, k, v, o, do, lse, dq, dk, dv, ctx.scale, o_res8=o_res8)
        return dq, dk, dv, None, None, None


def attention128(q, k, v, scale=None, o_pad=0, precise_delta="int8"):
    """q [B, H, Sq, 128], k / v [B, H, Skv, 128] (bf16, last dim contiguous) -> [B, H, Sq, 128].  precise_delta: "int8" (default) / None -- see "Precise delta" above"""
    if precise_delta not in (None, "int8"):
        raise ValueError('attention128: precise_delta is "int8" or None (the bf16 residual form exists at head_dim 64 only)')
    return _Attention128Fn.apply(q, k, v, q.shape[-1] ** -0.5 if scale is None else scale, int(o_pad), precise_delta)


class _QKNormAttentionFn(torch.autograd.Function):
    """qkv [B,S,3*H*64] (fused QKV GEMM output) -> attention output [B,S,H*64].
    QK-norm (+ optional 3D RoPE on tokens >= text_len) -> flash attention; backward returns dqkv in the same layout."""

    @staticmethod
    def forward(ctx, qkv, wq, bq, wk, bk, rope_cos, rope_sin, text_len, H, eps, o_pad, grad_pad, rope_mode=0, recompute_qk=False, precise_delta=None, fwd_policy=None):
        """o_pad / grad_pad: the attention output / the gradient of qkv are returned as heads of buffers that much wider (the
        LoRA tails of the projections on either side, see LoraExt).  recompute_qk: the normalised q / k are not kept for the backward
        but made again from qkv (one more pass of the QK-norm kernel, bit-identical) -- a third of this node's saved bytes."""
        _req(qkv, torch.bfloat16)
        B, S, W = qkv.shape
        Dh = W // (3 * H)
        qkv5 = qkv.view(B, S, 3, H, Dh)
        q_in, k_in, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # [B,H,S,Dh] views
        qn = torch.empty(B, H, S, Dh, dtype=torch.bfloat16, device=qkv.device)
        kn = torch.empty_like(qn)
        _timed("qknorm_rope_fwd", 8.0 * B * H * S * Dh, lambda: _lib.call(
            "vgpa_qknorm_rope_fwd", q_in, k_in, qn, kn, _bhs_strides(q_in), _bhs_strides(k_in), _bhs_strides(qn), _bhs_strides(kn),
            wq, bq, wk, bk, rope_cos, rope_sin, text_len, B, H, S, Dh, float(eps), float(Dh ** -0.5 * LOG2E), int(rope_mode), _stream()), "byte")
        # what the output's bf16 rounding drops, for the backward's delta -- only where a backward will run and on the w1 forward; lean activations keep it too
        # (1 byte per output element next to the 6 that recompute_qk gives back)
        o_res = None
        if precise_delta and "fwd" in ATTN_W1 and ctx.needs_input_grad[0]:
            o_res = torch.empty(B, S, H * Dh, dtype=torch.uint8 if precise_delta == "int8" else torch.bfloat16, device=qkv.device)
        o, lse = attention_fwd_raw(qn, kn, v, q_prescaled=True, o_pad=o_pad, o_res=o_res, policy=fwd_policy)
        if recompute_qk:
            ctx.save_for_backward(qkv, o, lse, wq, wk, rope_cos, rope_sin, bq, bk, o_res)
        else:
            ctx.save_for_backward(qkv, qn, kn, o, lse, wq, wk, rope_cos, rope_sin, o_res)
        ctx.meta = (text_len, H, eps, grad_pad, int(rope_mode), bool(recompute_qk))
        return o

    @staticmethod
    def backward(ctx, do):
        text_len, H, eps, grad_pad, rope_mode, recompute_qk = ctx.meta
        if recompute_qk:
            qkv, o, lse, wq, wk, rope_cos, rope_sin, bq, bk, o_res = ctx.saved_tensors
        else:
            qkv, qn, kn, o, lse, wq, wk, rope_cos, rope_sin, o_res = ctx.saved_tensors
        B, S, W = qkv.shape
        Dh = W // (3 * H)
        do = do.contiguous()
        qkv5 = qkv.view(B, S, 3, H, Dh)
        q_in, k_in, v = (qkv5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        if recompute_qk:
            qn = torch.empty(B, H, S, Dh, dtype=torch.bfloat16, device=qkv.device)
            kn = torch.empty_like(qn)
            _timed("qknorm_rope_fwd", 8.0 * B * H * S * Dh, lambda: _lib.call(
                "vgpa_qknorm_rope_fwd", q_in, k_in, qn, kn, _bhs_strides(q_in), _bhs_strides(k_in), _bhs_strides(qn), _bhs_strides(kn),
                wq, bq, wk, bk, rope_cos, rope_sin, text_len, B, H, S, Dh, float(eps), float(Dh ** -0.5 * LOG2E), int(rope_mode), _stream()), "byte")
        dqkv = _padded_empty((B, S), W, grad_pad, qkv.dtype, qkv.device) if grad_pad else torch.empty_like(qkv)
        d5 = dqkv.unflatten(-1, (3, H, Dh))
        dq_in, dk_in, dv = (d5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dqn = torch.empty_like(qn)
        dkn = torch.empty_like(kn)
        ov = o.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3)
        dov = do.view(B, S, H, Dh).permute(0, 2, 1, 3)
        attention_bwd_raw(qn, kn, v, ov, dov, lse, dqn, dkn, dv, q_prescaled=True,
                          o_res=None if o_res is None else o_res.unflatten(-1, (H, Dh)).permute(0, 2, 1, 3))
        _timed("qknorm_rope_bwd", 12.0 * B * H * S * Dh, lambda: _lib.call(
            "vgpa_qknorm_rope_bwd", dqn, dkn, q_in, k_in, dq_in, dk_in, _bhs_strides(dqn), _bhs_strides(dkn), _bhs_strides(q_in),
            _bhs_strides(k_in), _bhs_strides(dq_in), _bhs_strides(dk_in), wq, wk, rope_cos, rope_sin, text_len, B, H, S, Dh,
            float(eps), rope_mode, _stream()), "byte")
        return dqkv, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def qknorm_attention(qkv, wq, bq, wk, bk, H, text_len=0, rope=None, eps=1e-6, o_pad=0, grad_pad=0, rope_mode=0, recompute_qk=False, precise_delta="int8", fwd_policy=None):
    """rope = (cos, sin) fp32 [S - text_len, 64]; rope_mode 0: interleaved pairs (diffusers' CogVideoX), 1: half-split pairs inside each
    32-feature half (VGGT's RotaryPositionEmbedding2D, tables from `rope2d_tables`).  precise_delta: "int8" | "bf16" | None, see "Precise delta"."""
    cos, sin = (None, None) if rope is None else rope
    if precise_delta not in (None, "int8", "bf16"):
        raise ValueError(f'precise_delta: "int8", "bf16" or None, got {precise_delta!r}')
    return _QKNormAttentionFn.apply(qkv, wq, bq, wk, bk, cos, sin, text_len, H, eps, o_pad, grad_pad, rope_mode, bool(recompute_qk), precise_delta, fwd_policy)


def rope2d_tables(pos, head_dim=64, frequency=100.0):
    """cos / sin rows [N, head_dim] fp32 of vggt/layers/rope.py:103-112,154-188 for integer positions pos [N, 2] = (y, x): features
    [0, d/2) rotate with y, [d/2, d) with x; inside a half, feature i and i + d/4 share the angle pos * frequency^(-2i / (d/2))."""
    half = head_dim // 2
    inv = 1.0 / (frequency ** (torch.arange(0, half, 2, device=pos.device).float() / half))
    ang = pos.float()[:, :, None] * inv[None, None, :]                 # [N, 2, half/2]
    ang = torch.cat([ang, ang], dim=-1).reshape(pos.shape[0], head_dim)  # [y-angles x2 | x-angles x2]
    return ang.cos().contiguous(), ang.sin().contiguous()
