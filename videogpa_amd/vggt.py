"""The VGGT aggregator's attention path on the HIP kernels (the scorer's front half shares the kernel family of the
denoiser -- QK-norm over head_dim 64, rotary embedding, full attention).  Mirrors the reference-held modules so that their
checkpoints load by name:

    Attention                vggt/layers/attention.py:20-72   (qkv, q_norm, k_norm, proj; forward(x, pos=None))
    RotaryPositionEmbedding2D vggt/layers/rope.py:60-188       (frequency 100; here it only carries the frequency -- the rotation is
                                                                fused into the QK-norm kernel, rope_mode 1)
    Block                    vggt/layers/block.py:30-108      (norm1, attn, ls1, norm2, mlp.fc1 / fc2, ls2; forward(x, pos=None))
    alternating_attention    vggt/models/aggregator.py:236-306 (frame attention on (B*S, P, C), global attention on (B, S*P, C))

Unlike the CogVideoX attention (un-vendored diffusers), this code is IN the reference tree, so the kernels behind it are pinned
against reference outputs: tests/golden/vggt_attention.pt, tests/test_gpu_vggt.py.  The backbone is frozen in the reference's use
(metrics only): gradients flow to the input and to the Linear layers (torch autograd around the kernels); LayerNorm / LayerScale
parameters get none.  head_dim must be 64 and qk_norm on (what VGGT-1B's aggregator uses: dim 1024, 16 heads).  Patch embedding
(DINOv2), camera / depth heads and checkpoint loading stay the caller's (third-party networks, outside the path)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .transformer import _f32


class RotaryPositionEmbedding2D(nn.Module):
    def __init__(self, frequency: float = 100.0, scaling_factor: float = 1.0):
        super().__init__()
        if scaling_factor != 1.0:
            raise NotImplementedError("scaling_factor != 1 is not used by VGGT")
        self.base_frequency = frequency

    def tables(self, pos, head_dim):
        """pos [B, N, 2] (identical for every batch row, as the aggregator builds it) -> (cos, sin) fp32 [N, head_dim]"""
        if pos.ndim == 3:
            if pos.shape[0] > 1 and not bool((pos == pos[:1]).all()):
                raise NotImplementedError("per-sample positions: the aggregator uses one grid for every frame")
            pos = pos[0]
        return ops.rope2d_tables(pos, head_dim, self.base_frequency)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True, proj_bias=True, attn_drop=0.0, proj_drop=0.0, norm_layer=nn.LayerNorm, qk_norm=False,
                 fused_attn=True, rope=None):
        super().__init__()
        if dim % num_heads or dim // num_heads != 64 or not qk_norm or attn_drop or proj_drop:
            raise NotImplementedError("the HIP attention path covers head_dim 64 with QK-norm and no dropout (VGGT's aggregator blocks)")
        self.num_heads, self.head_dim = num_heads, 64
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm, self.k_norm = norm_layer(64), norm_layer(64)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.rope = rope

    def forward(self, x, pos=None):
        B, N, C = x.shape
        qkv = self.qkv(x)                                   # [B, N, 3, H, 64] in memory: the layout the fused kernel reads
        rope = self.rope.tables(pos, 64) if (self.rope is not None and pos is not None) else None
        o = ops.qknorm_attention(qkv.contiguous(), _f32(self.q_norm.weight), _f32(self.q_norm.bias), _f32(self.k_norm.weight), _f32(self.k_norm.bias),
                                 self.num_heads, text_len=0, rope=rope, eps=self.q_norm.eps, rope_mode=1)
        return self.proj(o)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.fc2 = nn.Linear(hidden_features, in_features, bias=bias)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))               # nn.GELU() = the erf form (the tanh kernel of the denoiser is a different function)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=None, norm_layer=nn.LayerNorm,
                 qk_norm=False, fused_attn=True, rope=None):
        super().__init__()
        if not init_values:
            raise NotImplementedError("blocks without LayerScale are not used by VGGT's aggregator")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias, qk_norm=qk_norm, rope=rope)
        self.ls1 = LayerScale(dim, init_values)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values)

    def forward(self, x, pos=None):
        B = x.shape[0]
        n1 = ops.ln_modulate(x, _f32(self.norm1.weight), _f32(self.norm1.bias), None, 0, self.norm1.eps)
        a = self.attn(n1, pos=pos)
        g1 = _f32(self.ls1.gamma)[None, None].expand(B, 2, -1).contiguous()      # LayerScale = the gate of the fused residual + LayerNorm kernel
        x, n2 = ops.residual_ln(x, a, g1, _f32(self.norm2.weight), _f32(self.norm2.bias), None, 0, self.norm2.eps)
        g2 = _f32(self.ls2.gamma)[None, None].expand(B, 2, -1).contiguous()
        return ops.gate_residual(x, self.mlp(n2), g2, 0)


def alternating_attention(tokens, frame_blocks, global_blocks, B, S, pos=None, aa_order=("frame", "global")):
    """The aggregator's loop (vggt/models/aggregator.py:236-306) over already embedded tokens [B*S, P, C]: for every depth, a frame
    block on (B*S, P, C) and a global block on (B, S*P, C); returns the per-depth concatenated intermediates [B, S, P, 2C]."""
    P, C = tokens.shape[1], tokens.shape[2]
    outs = []
    for fb, gb in zip(frame_blocks, global_blocks):
        inter = {}
        for kind in aa_order:
            if kind == "frame":
                tokens = fb(tokens.reshape(B * S, P, C), pos=None if pos is None else pos.reshape(B * S, P, 2))
            elif kind == "global":
                tokens = gb(tokens.reshape(B, S * P, C), pos=None if pos is None else pos.reshape(B, S * P, 2))
            else:
                raise ValueError(f"Unknown attention type: {kind}")
            inter[kind] = tokens.reshape(B, S, P, C)
        outs.append(torch.cat([inter["frame"], inter["global"]], dim=-1))
    return outs, tokens
