"""The VGGT aggregator's attention path on the HIP kernels (the scorer's front half shares the kernel family of the
denoiser -- QK-norm over head_dim 64, rotary embedding, full attention).  Mirrors the reference-held modules so that their
checkpoints load by name:

    Attention                vggt/layers/attention.py:20-72   (qkv, q_norm, k_norm, proj; forward(x, pos=None))
    RotaryPositionEmbedding2D vggt/layers/rope.py:60-188       (frequency 100; here it only carries the frequency -- the rotation is
                                                                fused into the QK-norm kernel, rope_mode 1)
    Block                    vggt/layers/block.py:30-108      (norm1, attn, ls1, norm2, mlp.fc1 / fc2, ls2; forward(x, pos=None))
    alternating_attention    vggt/models/aggregator.py:236-306 (frame attention on (B*S, P, C), global attention on (B, S*P, C))
    Aggregator               vggt/models/aggregator.py:25-258  (camera / register tokens, positions, aa_block_num x aa_order loop, list of concatenated
                                                                intermediates; state-dict names patch_embed.*, frame_blocks.N.*, global_blocks.N.*,
                                                                camera_token, register_token)

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
                 qk_norm=False, fused_attn=True, rope=None, ln_eps=None):
        super().__init__()
        if not init_values:
            raise NotImplementedError("blocks without LayerScale are not used by VGGT's aggregator")
        if ln_eps is not None:      # Depth Anything 3's DINOv2 block (depth_anything_3/model/dinov2/layers/block.py:26-75) passes ln_eps = 1e-6
            base_norm = norm_layer
            norm_layer = lambda d: base_norm(d, eps=ln_eps)
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


class PositionGetter:
    """vggt/layers/rope.py:24-57: (y, x) grid coordinates of the patches, [batch, height * width, 2] integer, cached per grid size"""

    def __init__(self):
        self.position_cache = {}

    def __call__(self, batch_size, height, width, device):
        key = (height, width, str(device))
        if key not in self.position_cache:
            yy, xx = torch.meshgrid(torch.arange(height, device=device), torch.arange(width, device=device), indexing="ij")
            self.position_cache[key] = torch.stack([yy.reshape(-1), xx.reshape(-1)], dim=-1)
        return self.position_cache[key][None].expand(batch_size, -1, -1).clone()


class PatchEmbed(nn.Module):
    """vggt/layers/patch_embed.py:25-78 (the aggregator's patch_embed="conv" form): Conv2d with kernel = stride = patch, flattened row-major"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        ph, pw = self.patch_size
        if x.shape[-2] % ph or x.shape[-1] % pw:
            raise AssertionError(f"input image size {tuple(x.shape[-2:])} is not a multiple of the patch size {self.patch_size}")
        return self.proj(x).flatten(2).transpose(1, 2)


def slice_expand_and_flatten(token_tensor, B, S):
    """vggt/models/aggregator.py:309-331: a (1, 2, X, C) special token -> (B*S, X, C): entry 0 for the first frame of every sequence, entry 1 for the
    other S - 1 frames"""
    first = token_tensor[:, 0:1].expand(B, 1, *token_tensor.shape[2:])
    rest = token_tensor[:, 1:2].expand(B, S - 1, *token_tensor.shape[2:])
    return torch.cat([first, rest], dim=1).reshape(B * S, *token_tensor.shape[2:])


class Aggregator(nn.Module):
    """vggt/models/aggregator.py:25-258 on the HIP attention path: `forward(images [B, S, 3, H, W] in [0, 1]) -> (list of [B, S, P, 2C] per depth,
    patch_start_idx)`.  Constructor arguments, parameter names and the order of operations are the reference's, so an aggregator state dict loads with
    load_state_dict.  patch_embed: "conv" builds the reference's PatchEmbed; the DINOv2 backbones ("dinov2_vitl14_reg", ...) are third-party networks
    outside this path -- pass the constructed module instead (anything mapping [B*S, 3, H, W] to patch tokens [B*S, N, C] or to a dict carrying
    "x_norm_patchtokens"), it is registered under the same name `patch_embed`."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0, num_register_tokens=4, block_fn=Block,
                 qkv_bias=True, proj_bias=True, ffn_bias=True, patch_embed="dinov2_vitl14_reg", aa_order=("frame", "global"), aa_block_size=1,
                 qk_norm=True, rope_freq=100, init_values=0.01):
        super().__init__()
        if isinstance(patch_embed, nn.Module):
            self.patch_embed = patch_embed
        elif "conv" in patch_embed:
            self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=3, embed_dim=embed_dim)
        else:
            raise NotImplementedError(f"patch_embed={patch_embed!r}: the DINOv2 backbone is a third-party network outside this path; construct it and pass "
                                      "the module as patch_embed=")
        self.rope = RotaryPositionEmbedding2D(frequency=rope_freq) if rope_freq > 0 else None
        self.position_getter = PositionGetter() if self.rope is not None else None
        mk = lambda: block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, proj_bias=proj_bias, ffn_bias=ffn_bias,
                              init_values=init_values, qk_norm=qk_norm, rope=self.rope)
        self.frame_blocks = nn.ModuleList([mk() for _ in range(depth)])
        self.global_blocks = nn.ModuleList([mk() for _ in range(depth)])
        self.depth, self.aa_order, self.patch_size, self.aa_block_size = depth, list(aa_order), patch_size, aa_block_size
        if depth % aa_block_size != 0:
            raise ValueError(f"depth ({depth}) must be divisible by aa_block_size ({aa_block_size})")
        self.aa_block_num = depth // aa_block_size
        # two camera tokens and two sets of register tokens: one for the first frame, one for the rest
        self.camera_token = nn.Parameter(torch.randn(1, 2, 1, embed_dim))
        self.register_token = nn.Parameter(torch.randn(1, 2, num_register_tokens, embed_dim))
        self.patch_start_idx = 1 + num_register_tokens
        nn.init.normal_(self.camera_token, std=1e-6)
        nn.init.normal_(self.register_token, std=1e-6)
        self.register_buffer("_resnet_mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 1, 3, 1, 1), persistent=False)
        self.register_buffer("_resnet_std", torch.tensor([0.229, 0.224, 0.225]).view(1, 1, 3, 1, 1), persistent=False)
        self.use_reentrant = False

    def _run(self, blocks, idx, tokens, pos):
        if self.training and torch.is_grad_enabled():      # the reference checkpoints every block in training mode (:268-271, :292-295)
            from torch.utils.checkpoint import checkpoint
            return checkpoint(blocks[idx], tokens, pos, use_reentrant=self.use_reentrant)
        return blocks[idx](tokens, pos=pos)

    def forward(self, images):
        B, S, C_in, H, W = images.shape
        if C_in != 3:
            raise ValueError(f"Expected 3 input channels, got {C_in}")
        images = ((images - self._resnet_mean) / self._resnet_std).reshape(B * S, C_in, H, W)
        patch_tokens = self.patch_embed(images.to(next(self.frame_blocks.parameters()).dtype))
        if isinstance(patch_tokens, dict):
            patch_tokens = patch_tokens["x_norm_patchtokens"]
        tokens = torch.cat([slice_expand_and_flatten(self.camera_token, B, S).to(patch_tokens.dtype),
                            slice_expand_and_flatten(self.register_token, B, S).to(patch_tokens.dtype), patch_tokens], dim=1)
        pos = None
        if self.rope is not None:
            pos = self.position_getter(B * S, H // self.patch_size, W // self.patch_size, device=images.device)
            if self.patch_start_idx > 0:      # special tokens sit at position 0, the patch grid starts at 1 (:215-224)
                pos = torch.cat([pos.new_zeros(B * S, self.patch_start_idx, 2), pos + 1], dim=1)
        _, P, C = tokens.shape
        tokens = tokens.contiguous()
        fi = gi = 0
        out = []
        for _ in range(self.aa_block_num):
            inter = {}
            for kind in self.aa_order:
                if kind not in ("frame", "global"):
                    raise ValueError(f"Unknown attention type: {kind}")
                shape = (B * S, P) if kind == "frame" else (B, S * P)
                tokens = tokens.reshape(*shape, C)
                pk = None if pos is None else pos.reshape(*shape, 2)
                got = []
                for _ in range(self.aa_block_size):
                    if kind == "frame":
                        tokens, fi = self._run(self.frame_blocks, fi, tokens, pk), fi + 1
                    else:
                        tokens, gi = self._run(self.global_blocks, gi, tokens, pk), gi + 1
                    got.append(tokens.reshape(B, S, P, C))
                inter[kind] = got
            out += [torch.cat([f, g], dim=-1) for f, g in zip(inter["frame"], inter["global"])]
        return out, self.patch_start_idx
