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
  * the four LoRA projections of each attention run ops.linear_lora_ext through lora.LoraLinear once wrapped by get_peft_model.
Call convention as upstream:  model(list of [C,F,H,W], t=[B] or [B, seq_len], context=list of [n, text_dim], seq_len=int) -> list of
[C_out,F,H,W] fp32.  All samples of a call must share one latent shape (the reference's batches do), and seq_len must equal the token
count (it does: 03_train.py:176-179 computes it from the latent)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops

_F32, _BF16 = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr_dtype(x):
    if x.dtype == torch.float32:
        return _F32
    if x.dtype == torch.bfloat16:
        return _BF16
    raise TypeError(f"videogpa_amd.wan_model: rows must be fp32 or bf16, got {x.dtype}")


class _LnModFn(torch.autograd.Function):
    """bf16( LN_eps(x) [rounded to bf16] * ln_w + ln_b, then * (1 + scale[gid]) + shift[gid] );   x [rows, D] fp32 or bf16"""

    @staticmethod
    def forward(ctx, x, gid, ln_w, ln_b, shift, scale, eps, round_xhat):
        rows, D = x.shape
        x = x.contiguous()
        out = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ms = 0 if shift is None else shift.stride(0)
        ops._timed("wan_ln_mod_fwd", (x.element_size() + 2.0) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_fwd", x, _ptr_dtype(x), gid, ln_w, ln_b, shift, scale, ms, rows, D, float(eps), int(round_xhat), out, mean, rstd, _stream()), "byte")
        ctx.save_for_backward(x, mean, rstd, gid, ln_w, scale)
        ctx.ms = ms
        return out

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gid, ln_w, scale = ctx.saved_tensors
        rows, D = x.shape
        dx = torch.empty(rows, D, dtype=torch.float32, device=x.device)
        ops._timed("wan_ln_mod_bwd", (x.element_size() + 6.0) * rows * D, lambda: _lib.call(
            "vgpa_wan_ln_mod_bwd", dy.contiguous(), x, _ptr_dtype(x), mean, rstd, gid, ln_w, scale, ctx.ms, rows, D, None, dx, _stream()), "byte")
        return dx.to(x.dtype), None, None, None, None, None, None, None


class _GateResidualFn(torch.autograd.Function):
    """fp32: x + y(bf16) * gate[gid]      (gate None: 1)"""

    @staticmethod
    def forward(ctx, x, y, gid, gate):
        rows, D = y.shape
        out = torch.empty(rows, D, dtype=torch.float32, device=y.device)
        ms = 0 if gate is None else gate.stride(0)
        ops._timed("wan_gate_residual", 10.0 * rows * D, lambda: _lib.call(
            "vgpa_wan_gate_residual", x.contiguous(), y.contiguous(), gid, gate, ms, rows, D, out, _stream()), "byte")
        ctx.save_for_backward(gid, gate)
        ctx.ms = ms
        return out

    @staticmethod
    def backward(ctx, dout):
        gid, gate = ctx.saved_tensors
        rows, D = dout.shape
        dy = torch.empty(rows, D, dtype=torch.bfloat16, device=dout.device)
        dout = dout.contiguous()
        ops._timed("wan_gate_bwd", 6.0 * rows * D, lambda: _lib.call("vgpa_wan_gate_bwd", dout, gid, gate, ctx.ms, rows, D, dy, _stream()), "byte")
        return dout, dy, None, None


class _RmsRopeFn(torch.autograd.Function):
    """WanRMSNorm over the full row, bf16 weight, RoPE per head:  u [B, L, D] bf16 -> [B, L, D] bf16"""

    @staticmethod
    def forward(ctx, u, w, cos, sin, head_dim, eps):
        B, L, D = u.shape
        u = u.contiguous()
        out = torch.empty_like(u)
        rstd = torch.empty(B * L, dtype=torch.float32, device=u.device)
        ops._timed("wan_rms_rope_fwd", 4.0 * B * L * D, lambda: _lib.call(
            "vgpa_wan_rms_rope_fwd", u, w, cos, sin, L, head_dim, B * L, D, float(eps), out, rstd, _stream()), "byte")
        ctx.save_for_backward(u, rstd, w, cos, sin)
        ctx.head_dim = head_dim
        return out

    @staticmethod
    def backward(ctx, dout):
        u, rstd, w, cos, sin = ctx.saved_tensors
        B, L, D = u.shape
        du = torch.empty_like(u)
        ops._timed("wan_rms_rope_bwd", 6.0 * B * L * D, lambda: _lib.call(
            "vgpa_wan_rms_rope_bwd", dout.contiguous(), u, rstd, w, cos, sin, L, ctx.head_dim, B * L, D, du, _stream()), "byte")
        return du, None, None, None, None, None


def ln_mod(x, gid=None, ln_w=None, ln_b=None, shift=None, scale=None, eps=1e-6, round_xhat=False):
    return _LnModFn.apply(x, gid, ln_w, ln_b, shift, scale, eps, round_xhat)


def gate_residual(x, y, gid=None, gate=None):
    return _GateResidualFn.apply(x, y, gid, gate)


def rms_rope(u, w, cos=None, sin=None, head_dim=128, eps=1e-6):
    return _RmsRopeFn.apply(u, w, cos, sin, head_dim, eps)


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

    def _heads(self, t, B, S):
        return t.view(B, S, self.num_heads, self.head_dim).permute(0, 2, 1, 3)

    def _norm(self, norm, u, rope):
        if norm is None:
            if rope is not None:
                raise NotImplementedError("RoPE without QK-norm")
            return u
        return rms_rope(u, norm.weight, rope[0] if rope else None, rope[1] if rope else None, self.head_dim, norm.eps)

    def forward(self, x, B, L, rope):
        """x [B*L, dim] bf16 -> [B*L, dim] bf16"""
        q = self._norm(self.norm_q, self.q(x).view(B, L, self.dim), rope)
        k = self._norm(self.norm_k, self.k(x).view(B, L, self.dim), rope)
        v = self.v(x).view(B, L, self.dim)
        o = ops.attention128(self._heads(q, B, L), self._heads(k, B, L), self._heads(v, B, L))
        return self.o(o.permute(0, 2, 1, 3).reshape(B * L, self.dim))


class WanCrossAttention(WanSelfAttention):
    def forward(self, x, context, B, L):
        """x [B*L, dim] bf16, context [B, T, dim] bf16 (every one of the T text positions is attended: upstream passes k_lens=None)"""
        T = context.shape[1]
        q = self._norm(self.norm_q, self.q(x).view(B, L, self.dim), None)
        k = self._norm(self.norm_k, self.k(context.reshape(B * T, self.dim)).view(B, T, self.dim), None)
        v = self.v(context.reshape(B * T, self.dim)).view(B, T, self.dim)
        o = ops.attention128(self._heads(q, B, L), self._heads(k, B, T), self._heads(v, B, T))
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
        first = x.dtype == torch.bfloat16                           # norm1(x).type_as(x) rounds only while the stream is still bf16
        h = ln_mod(x, gid, None, None, tab[:, 0], tab[:, 1], self.eps, round_xhat=first)
        x = gate_residual(x.float() if first else x, self.self_attn(h, B, L, rope), gid, tab[:, 2])
        if isinstance(self.norm3, nn.Identity):
            h = x.to(torch.bfloat16)
        else:
            h = ln_mod(x, None, self.norm3.weight.float(), self.norm3.bias.float(), None, None, self.eps)
        x = gate_residual(x, self.cross_attn(h, context, B, L), None, None)
        h = ln_mod(x, gid, None, None, tab[:, 3], tab[:, 4], self.eps)
        lin = ops.frozen_linear_fp8 if self.fp8_ffn else ops.frozen_linear
        y = lin(ops.gelu_tanh(lin(h, self.ffn[0].weight, self.ffn[0].bias)), self.ffn[2].weight, self.ffn[2].bias)
        return gate_residual(x, y, gid, tab[:, 5])


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, out_dim * math.prod(patch_size))
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e, gid):
        """x [B*L, dim] fp32, e [G, dim] fp32 -> [B*L, out] fp32: upstream runs the whole head under autocast(float32); once per forward,
        1.5 % of one block's bytes: torch elementwise + an fp32 library GEMM"""
        tab = self.modulation.float() + e[:, None]                  # [G, 2, dim]
        idx = gid.long()
        h = F.layer_norm(x, (self.dim,), None, None, self.eps) * (1 + tab[:, 1][idx]) + tab[:, 0][idx]
        return F.linear(h, self.head.weight.float(), self.head.bias.float())


class WanModel(nn.Module):
    def __init__(self, model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256, text_dim=4096,
                 out_dim=48, num_heads=24, num_layers=30, window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6):
        super().__init__()
        assert model_type in ("t2v", "i2v", "ti2v", "s2v")
        self.model_type, self.patch_size, self.text_len, self.in_dim, self.dim, self.ffn_dim = model_type, tuple(patch_size), text_len, in_dim, dim, ffn_dim
        self.freq_dim, self.text_dim, self.out_dim, self.num_heads, self.num_layers, self.eps = freq_dim, text_dim, out_dim, num_heads, num_layers, eps
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([WanAttentionBlock(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps) for _ in range(num_layers)])
        self.head = Head(dim, out_dim, self.patch_size, eps)
        self.gradient_checkpointing = False
        self._rope = {}
        self.init_weights()

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

    def enable_gradient_checkpointing(self, enabled=True):
        """the reference wraps every block's forward in torch.utils.checkpoint (03_train.py:150-159)"""
        self.gradient_checkpointing = enabled

    def enable_fp8(self, enabled=True):
        """BASELINE.json configs[4] "fp8 MFMA path": the frozen feed-forward projections (61 % of the linear FLOPs per token) take
        OCP-e4m3 operands -- per-row dynamic activation scales (csrc/fp8.hip), per-output-row weight scales, fp32 accumulation, bf16
        out, forward and dX.  The LoRA-carrying q/k/v/o projections stay in bf16."""
        for blk in self.blocks:
            blk.fp8_ffn = enabled

    def _rope_tables(self, grid, device):
        key = (tuple(grid), str(device))
        if key not in self._rope:
            self._rope = {key: rope_tables(grid, self.dim // self.num_heads, device)}
        return self._rope[key]

    def forward(self, x, t, context, seq_len, y=None):
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        xb = torch.stack(list(x))
        B, C, Fr, H, W = xb.shape
        pt, ph, pw = self.patch_size
        f, h, w = Fr // pt, H // ph, W // pw
        L = f * h * w
        if seq_len != L:
            raise NotImplementedError(f"seq_len {seq_len} != token count {L}: padded sequences are not on the training path (03_train.py:176-179)")
        dev = xb.device
        wdt = self.patch_embedding.weight.dtype
        # patch embedding: Conv3d with kernel = stride = patch  ==  one GEMM over the (c, pt, ph, pw) patch vectors
        patches = xb.view(B, C, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * L, C * pt * ph * pw).to(wdt)
        tok = F.linear(patches, self.patch_embedding.weight.view(self.dim, -1), self.patch_embedding.bias)           # [B*L, dim]
        # time embedding on the distinct timesteps only
        if t.dim() == 1:
            t = t[:, None].expand(B, L)
        tvals, inv = torch.unique(t.reshape(-1).float(), return_inverse=True)
        gid = inv.to(torch.int32).contiguous()
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
        for blk in self.blocks:
            if self.gradient_checkpointing and torch.is_grad_enabled():
                from torch.utils.checkpoint import checkpoint
                xs = checkpoint(blk, xs, e0, gid, B, L, rope, ctx, use_reentrant=False)
            else:
                xs = blk(xs, e0, gid, B, L, rope, ctx)
        out = self.head(xs.float(), e, gid)                                                                             # [B*L, out_dim * prod(patch)]
        # unpatchify: [f, h, w, pt, ph, pw, c] -> [c, f pt, h ph, w pw]
        c = self.out_dim
        out = out.view(B, f, h, w, pt, ph, pw, c).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, c, f * pt, h * ph, w * pw)
        return [u.float() for u in out]
