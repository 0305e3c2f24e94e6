"""Drop-in for diffusers' `CogVideoXTransformer3DModel` on MI355X (the denoiser the reference trains:
train/CogVideoX-5B/03_train.py:101,110,134-151; generate/CogVideoX-5B.py:17,70-77).

Same constructor config keys, same module tree / state-dict names, same
`forward(hidden_states, encoder_hidden_states, timestep, timestep_cond=None, ofs=None, image_rotary_emb=None,
attention_kwargs=None, return_dict=True) -> .sample`.  The compute path is MI355X-first, not diffusers':

  * one residual stream x[B, S, D] bf16 with the 226 text tokens first -- the text/video split only selects a
    per-range modulation vector inside the kernels (no cat/split);
  * AdaLN-Zero LN+modulate, gated residual, GELU-tanh, QK-norm+RoPE and full 3D attention run hand-written HIP
    kernels (videogpa_amd.ops); the plain dense projections go to hipBLASLt through torch.matmul with a fused
    [3D, D] QKV weight;
  * activations are kept (no recompute) -- 288 GB HBM holds the ~60 GB/sequence a 42-block backward needs;
    `enable_gradient_checkpointing()` is still honoured for API parity.

Only adapter (LoRA) parameters receive gradients: like the reference, the base model is frozen
(train/CogVideoX-5B/03_train.py:102-111); norm / modulation parameters are treated as constants by the kernels.
"""
import json
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _Config(dict):
    """dict with attribute access + assignment (the reference does `model.config.x = ...`,
    train/CogVideoX1.5-5B/03_train.py:95,113)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


DEFAULT_CONFIG = dict(
    num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16, flip_sin_to_cos=True, freq_shift=0,
    time_embed_dim=512, ofs_embed_dim=None, text_embed_dim=4096, num_layers=30, dropout=0.0, attention_bias=True,
    sample_width=90, sample_height=60, sample_frames=49, patch_size=2, patch_size_t=None, temporal_compression_ratio=4,
    max_text_seq_length=226, activation_fn="gelu-approximate", timestep_activation_fn="silu", norm_elementwise_affine=True,
    norm_eps=1e-5, spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False,
    use_learned_positional_embeddings=False, patch_bias=True)

# CogVideoX-5B T2V transformer/config.json
COGVIDEOX_5B = dict(DEFAULT_CONFIG, num_attention_heads=48, num_layers=42, use_rotary_positional_embeddings=True)
COGVIDEOX_5B_I2V = dict(COGVIDEOX_5B, in_channels=32, use_learned_positional_embeddings=True)
COGVIDEOX_1_5_5B = dict(COGVIDEOX_5B, patch_size_t=2, patch_bias=False, sample_height=96, sample_width=170, sample_frames=81)


class Transformer3DModelOutput(SimpleNamespace):
    pass


def _f32(t):
    """fp32 copy of a (frozen) norm / modulation parameter for the kernels, cached on the parameter while it is unchanged."""
    c = getattr(t, "_vgpa_f32", None)
    key = (t._version, t.data_ptr(), t.dtype, t.device)     # data_ptr: `param.data = other` keeps the version number
    if c is None or c[0] != key:
        c = (key, t.detach().float().contiguous())
        t._vgpa_f32 = c
    return c[1]


class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, cfg, dim):
        super().__init__()
        p, pt = cfg.patch_size, cfg.patch_size_t
        self.patch_size, self.patch_size_t = p, pt
        if pt is None:
            self.proj = nn.Conv2d(cfg.in_channels, dim, kernel_size=(p, p), stride=p, bias=cfg.patch_bias)
        else:
            self.proj = nn.Linear(cfg.in_channels * p * p * pt, dim, bias=cfg.patch_bias)
        self.text_proj = nn.Linear(cfg.text_embed_dim, dim)
        self.use_learned = cfg.use_learned_positional_embeddings
        self.use_sincos = not cfg.use_rotary_positional_embeddings and not self.use_learned
        if self.use_learned:
            n_tok = cfg.max_text_seq_length + ((cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1) * \
                (cfg.sample_height // p) * (cfg.sample_width // p)
            self.pos_embedding = nn.Parameter(torch.zeros(1, n_tok, dim))
        if self.use_sincos:
            raise NotImplementedError("sincos positional table (CogVideoX-2B) is outside the 5B hot path")

    def forward(self, text, video):
        B, Fr, C, H, W = video.shape
        p, pt = self.patch_size, self.patch_size_t
        text = F.linear(text, self.text_proj.weight, self.text_proj.bias)
        if pt is None:
            # Conv2d(k=2,s=2) == GEMM over (c, dy, dx) patches; rows ordered frame-major then row-major
            x = video.reshape(B, Fr, C, H // p, p, W // p, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, Fr * (H // p) * (W // p), C * p * p)
            x = F.linear(x, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
        else:
            x = video.permute(0, 1, 3, 4, 2).reshape(B, Fr // pt, pt, H // p, p, W // p, p, C)
            x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)
            x = F.linear(x, self.proj.weight, self.proj.bias)
        emb = torch.cat([text, x], dim=1)
        if self.use_learned:
            emb = emb + self.pos_embedding[:, : emb.shape[1]].to(emb.dtype)
        return emb.contiguous()


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, t_emb):
        return self.linear_2(F.silu(self.linear_1(t_emb)))


class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, cond_dim, dim, affine, eps):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=affine)

    def modulation(self, temb):
        """-> mod fp32 [B,4,D] (shift_v, 1+scale_v, shift_t, 1+scale_t), gates fp32 [B,2,D] (gate_v, gate_t).
        `1 + scale` is formed in the model dtype first, as upstream does."""
        m = F.linear(F.silu(temb), self.linear.weight, self.linear.bias)
        shift, scale, gate, e_shift, e_scale, e_gate = m.chunk(6, dim=1)
        mod = torch.stack([shift, 1 + scale, e_shift, 1 + e_scale], dim=1).float().contiguous()
        gates = torch.stack([gate, e_gate], dim=1).float().contiguous()
        return mod.detach(), gates.detach()


class AdaLayerNorm(nn.Module):
    def __init__(self, cond_dim, dim, affine, eps):
        super().__init__()
        self.linear = nn.Linear(cond_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=affine)


class _GELUProj(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)


class FeedForward(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])


def _parts(mod):
    """(weight, bias, lora) of an nn.Linear or a LoRA-wrapped linear (videogpa_amd.lora.LoraLinear)."""
    if hasattr(mod, "base_layer"):
        return mod.base_layer.weight, mod.base_layer.bias, mod.active_lora()
    return mod.weight, mod.bias, None


def _adapter(mod):
    """((A, B, scaling) | None, enabled): the adapter that EXISTS on a wrapped linear (also while `disable_adapter()` is in
    force -- the frozen-reference pass runs the same extended GEMM with a zero LoRA tail, see ops.LoraExt) and whether it is on."""
    if not hasattr(mod, "base_layer") or mod.merged or not mod.active_adapters:
        return None, False
    if len(mod.active_adapters) != 1:
        raise NotImplementedError("one active adapter at a time")
    n = mod.active_adapters[0]
    return (mod.lora_A[n].weight, mod.lora_B[n].weight, mod.scaling[n]), not mod.disable_adapters


class AttentionCore:
    """The MI355X attention path over any module that carries diffusers' attention attributes (`to_q`, `to_k`, `to_v`, `norm_q`,
    `norm_k`, `to_out[0]`, `heads`): fused-QKV GEMM with the LoRA adapters riding as extra K, QK-norm + RoPE + flash attention
    kernels, output projection.  Used by this package's `Attention` module and by the diffusers-processor seam
    (videogpa_amd/attn_processor.py); holds the caches that belong to one attention layer."""

    def __init__(self, mod, qk_eps=None):
        self.mod = mod
        self.qk_eps = qk_eps if qk_eps is not None else getattr(mod.norm_q, "eps", 1e-6)
        self._fused = None
        self._qkv_ext = self._out_ext = None
        self.precise_delta = ops.precise_delta_default()     # "int8" | "bf16" | None: what the forward keeps of its output for the backward's delta (ops "Precise delta")
        self.fwd_policy = ops.AttnFwdPolicy()                # bound-shifted or online-softmax forward, from the kernel's own redo count (ops.AttnFwdPolicy)

    def fused_qkv(self):
        """[3D, D] weight / [3D] bias, cached while the three base weights are unchanged (frozen base)."""
        m = self.mod
        ws = [_parts(x)[0] for x in (m.to_q, m.to_k, m.to_v)]
        key = tuple((w.data_ptr(), w._version, w.dtype, w.device) for w in ws)
        if self._fused is None or self._fused[0] != key:
            bs = [_parts(x)[1] for x in (m.to_q, m.to_k, m.to_v)]
            W = torch.cat([w.detach() for w in ws], dim=0)
            b = torch.cat([x.detach() for x in bs], dim=0) if bs[0] is not None else None
            self._fused = (key, W, b)
        return self._fused[1], self._fused[2]

    def lora_state(self):
        """(adapters of q/k/v, adapter of to_out, enabled).  All wrapped linears of one attention share the on / off state."""
        m = self.mod
        qa = [_adapter(x) for x in (m.to_q, m.to_k, m.to_v)]
        oa = _adapter(m.to_out[0])
        on = [e for a, e in qa + [oa] if a is not None]
        return [a for a, _ in qa], oa[0], (bool(on) and all(on))

    def pads(self):
        """(in_pad, out_pad): widths of the LoRA tails after this attention's input n (q/k/v adapters) and after its output
        (to_out adapter) -- what the producers of those tensors append so that the projections run as ONE extended GEMM."""
        qa, oa, _ = self.lora_state()
        act = [a for a in qa if a is not None]
        in_pad = len(act) * ops._pad_rank(act[0][0].shape[0]) if act else 0
        out_pad = ops._pad_rank(oa[0].shape[0]) if oa is not None else 0
        return in_pad, out_pad

    def forward(self, n, text_len, rope, lean_src=None):
        """lean_src = (x, ln_w, ln_b, mod, eps) with n == LN-modulate(x): "lean activations" -- the backward makes n and the normalised q / k
        again (two cheap row-kernel passes, bit-identical) instead of keeping them: 1.5 of the 6.6 GB a block saves at S = 41 026."""
        m = self.mod
        W, b = self.fused_qkv()
        qa, oa, on = self.lora_state()
        in_pad, out_pad = self.pads()
        if in_pad:
            if self._qkv_ext is None:
                self._qkv_ext = ops.LoraExt()
            fn = None
            if lean_src is not None:
                xs, lw, lb, lmod, leps = lean_src
                fn = (lambda xs_, lw_, lb_, lmod_: ops.ln_modulate_recompute(xs_, lw_, lb_, lmod_, text_len, leps), (xs, lw, lb, lmod))
            qkv = ops.linear_lora_ext(n, W, b, self._qkv_ext, qa, enabled=on, x_recompute=fn)
        else:
            qkv = ops.frozen_linear(n, W, b)
        a = ops.qknorm_attention(qkv, _f32(m.norm_q.weight), _f32(m.norm_q.bias), _f32(m.norm_k.weight), _f32(m.norm_k.bias),
                                 m.heads, text_len, rope, self.qk_eps, o_pad=out_pad, grad_pad=in_pad, recompute_qk=lean_src is not None,
                                 precise_delta=self.precise_delta, fwd_policy=self.fwd_policy)
        wo, bo, _ = _parts(m.to_out[0])
        if out_pad:
            if self._out_ext is None:
                self._out_ext = ops.LoraExt()
            return ops.linear_lora_ext(a, wo, bo, self._out_ext, [oa], enabled=on)
        return ops.frozen_linear(a, wo, bo)


class Attention(nn.Module):
    def __init__(self, dim, heads, head_dim, bias, qk_eps=1e-6):
        super().__init__()
        self.heads, self.head_dim = heads, head_dim
        self.to_q = nn.Linear(dim, dim, bias=bias)
        self.to_k = nn.Linear(dim, dim, bias=bias)
        self.to_v = nn.Linear(dim, dim, bias=bias)
        self.norm_q = nn.LayerNorm(head_dim, eps=qk_eps)
        self.norm_k = nn.LayerNorm(head_dim, eps=qk_eps)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim, bias=True), nn.Dropout(0.0)])
        self.qk_eps = qk_eps
        self.core = AttentionCore(self, qk_eps)
        self.processor = None

    def set_processor(self, processor):
        """diffusers' plugin point (`attn.set_processor(P)`): kept so that the seam object of videogpa_amd/attn_processor.py can
        be installed on this module too (tests); None restores the built-in path."""
        self.processor = processor

    def fused_qkv(self):
        return self.core.fused_qkv()

    def pads(self):
        return self.core.pads()

    def forward(self, n, text_len, rope, lean_src=None):
        return self.core.forward(n, text_len, rope, lean_src)


class CogVideoXBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, time_embed_dim, bias, affine, eps):
        super().__init__()
        self.norm1 = CogVideoXLayerNormZero(time_embed_dim, dim, affine, eps)
        self.attn1 = Attention(dim, heads, head_dim, bias)
        self.norm2 = CogVideoXLayerNormZero(time_embed_dim, dim, affine, eps)
        self.ff = FeedForward(dim, 4 * dim)
        self.eps = eps

    def modulations(self, temb):
        """((mod1, gates1), (mod2, gates2)) of this block for one conditioning embedding."""
        return self.norm1.modulation(temb), self.norm2.modulation(temb)

    def norm1_params(self):
        return _f32(self.norm1.norm.weight), _f32(self.norm1.norm.bias)

    def forward(self, x, n, gates1, mod2, gates2, text_len, rope, nxt_w, nxt_b, nxt_mod, nxt_eps, nxt_pad=0, lean=None):
        """x: residual stream; n = norm1(x) already modulated (produced by the previous block's fused residual+LN pass).
        Returns (x_out, n_next) where n_next is the NEXT normalisation (next block's norm1, or the model's norm_final)
        applied to x_out -- each gated residual add is fused with the LayerNorm that consumes it.  nxt_pad: LoRA tail width the
        next block's q/k/v projection wants behind n_next (Attention.pads).  lean = (ln_w, ln_b, mod) of THIS block's norm1: n is then not kept
        for the backward but made again from x (AttentionCore.forward)."""
        a = self.attn1(n, text_len, rope, None if lean is None else (x, lean[0], lean[1], lean[2], self.eps))
        x, n2 = ops.residual_ln(x, a, gates1, _f32(self.norm2.norm.weight), _f32(self.norm2.norm.bias), mod2, text_len, self.eps,
                                dy_pad=self.attn1.pads()[1])
        u = ops.frozen_linear(n2, self.ff.net[0].proj.weight, self.ff.net[0].proj.bias)
        g = ops.gelu_tanh(u)
        f = ops.frozen_linear(g, self.ff.net[2].weight, self.ff.net[2].bias)
        return ops.residual_ln(x, f, gates2, nxt_w, nxt_b, nxt_mod, text_len, nxt_eps, n_pad=nxt_pad)


def timestep_sincos(t, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class CogVideoXTransformer3DModel(nn.Module):
    config_name = "config.json"

    def __init__(self, **kwargs):
        super().__init__()
        cfg = _Config(DEFAULT_CONFIG)
        unknown = set(kwargs) - set(cfg) - {"_class_name", "_diffusers_version", "_name_or_path", "attention_out_bias", "norm_num_groups"}
        cfg.update(kwargs)
        self.config = cfg
        self._unknown_config = sorted(unknown)
        D = cfg.num_attention_heads * cfg.attention_head_dim
        if cfg.attention_head_dim != 64:
            raise NotImplementedError("the MI355X attention kernels are written for head_dim = 64 (every CogVideoX size)")
        self.patch_embed = CogVideoXPatchEmbed(cfg, D)
        self.time_embedding = TimestepEmbedding(D, cfg.time_embed_dim)
        if cfg.ofs_embed_dim is not None:
            self.ofs_embedding = TimestepEmbedding(cfg.ofs_embed_dim, cfg.ofs_embed_dim)
        self.transformer_blocks = nn.ModuleList([
            CogVideoXBlock(D, cfg.num_attention_heads, cfg.attention_head_dim, cfg.time_embed_dim, cfg.attention_bias,
                           cfg.norm_elementwise_affine, cfg.norm_eps) for _ in range(cfg.num_layers)])
        self.norm_final = nn.LayerNorm(D, cfg.norm_eps, cfg.norm_elementwise_affine)
        self.norm_out = AdaLayerNorm(cfg.time_embed_dim, D, cfg.norm_elementwise_affine, cfg.norm_eps)
        pt = cfg.patch_size_t or 1
        self.proj_out = nn.Linear(D, cfg.patch_size * cfg.patch_size * pt * cfg.out_channels)
        self.gradient_checkpointing = False
        self.lean_activations = False

    def enable_lean_activations(self, enabled=True):
        """Keep less per block for the backward: the LN output n1 (LoRA dA needs it) and the normalised q / k are made again from tensors that
        are saved anyway (the residual stream, the fused projection output) -- 23 % fewer saved bytes for two more row-kernel passes per block
        (+0.3 % of a step).  What it buys: CogVideoX1.5 at S = 41 026 keeps ALL 42 blocks resident in 288 GB instead of recomputing every
        fourth block (bench.py --config cfg4).  Results are bit-identical either way (the output's res8 bytes for the backward's delta are kept in
        both modes)."""
        self.lean_activations = bool(enabled)

    def set_precise_delta(self, mode="int8"):
        """what every attention of THIS model keeps of its output beyond the bf16 rounding for the backward's delta: "int8" (default: one byte per
        element), "bf16" (the residual as a bf16 tensor) or None (the textbook flash-attention backward) -- ops.py "Precise delta" has the why"""
        if mode not in (None, "int8", "bf16"):
            raise ValueError(f'precise_delta: "int8", "bf16" or None, got {mode!r}')
        for blk in self.transformer_blocks:
            blk.attn1.core.precise_delta = mode

    def attention_forward_report(self):
        """per block: which forward its attention runs and the fraction of strips the bound-shifted kernel last had to redo (ops.AttnFwdPolicy)"""
        return [{"block": i, "mode": b.attn1.core.fwd_policy.mode, "redo_fraction": b.attn1.core.fwd_policy.redo_fraction,
                 "switched_at_call": b.attn1.core.fwd_policy.switched_at} for i, b in enumerate(self.transformer_blocks)]

    # ------------------------------------------------------------------ diffusers-style protocol
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_gradient_checkpointing(self, stride=1):
        """diffusers' switch (train/CogVideoX-5B/03_train.py:107-108).  stride = 1 recomputes every block like the reference;
        stride = k > 1 recomputes only blocks 0, k, 2k, ... and keeps the activations of the others -- 288 GB of HBM3E usually
        has room for most of them (config 4, S = 41 026: every 2nd block recomputed fits in ~170 GB)."""
        self.gradient_checkpointing = True
        self.checkpoint_stride = max(1, int(stride))

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    @classmethod
    def from_config(cls, config, **kw):
        cfg = dict(config)
        cfg.update(kw)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        """Loads diffusers' on-disk layout: <path>/<subfolder>/config.json + diffusion_pytorch_model*.safetensors."""
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        model = cls(**{k: v for k, v in cfg.items()})
        files = sorted(f for f in os.listdir(root) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors weights under {root}")
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(root, f)))
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = "CogVideoXTransformer3DModel"
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "diffusion_pytorch_model.safetensors"))

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, timestep_cond=None, ofs=None,
                image_rotary_emb=None, attention_kwargs=None, return_dict=True):
        cfg = self.config
        if not hidden_states.is_cuda:
            raise RuntimeError("videogpa_amd.CogVideoXTransformer3DModel runs on MI355X only (no CPU fallback)")
        B, Fr, C, H, W = hidden_states.shape
        p = cfg.patch_size
        dt = self.dtype
        if dt != torch.bfloat16:
            raise RuntimeError("the MI355X path computes in bf16: load / cast the model with torch_dtype=torch.bfloat16")
        D = cfg.num_attention_heads * cfg.attention_head_dim
        timestep = torch.as_tensor(timestep, device=hidden_states.device).reshape(-1).expand(B)
        t_emb = timestep_sincos(timestep, D, cfg.flip_sin_to_cos, cfg.freq_shift).to(dt)
        emb = self.time_embedding(t_emb)
        if cfg.ofs_embed_dim is not None and ofs is not None:
            o_emb = timestep_sincos(ofs.reshape(-1).expand(B), cfg.ofs_embed_dim, cfg.flip_sin_to_cos, cfg.freq_shift).to(dt)
            emb = emb + self.ofs_embedding(o_emb)
        emb = emb.detach() if not any(p_.requires_grad for p_ in self.time_embedding.parameters()) else emb

        x = self.patch_embed(encoder_hidden_states.to(dt), hidden_states.to(dt))
        Lt = encoder_hidden_states.shape[1]
        rope = None
        if image_rotary_emb is not None:
            rope = (image_rotary_emb[0].float().contiguous(), image_rotary_emb[1].float().contiguous())

        blocks = self.transformer_blocks
        mods = [blk.modulations(emb) for blk in blocks]
        w0, b0 = blocks[0].norm1_params()
        n = ops.ln_modulate(x, w0, b0, mods[0][0][0], Lt, cfg.norm_eps, n_pad=blocks[0].attn1.pads()[0])
        for i, blk in enumerate(blocks):
            (_, gates1), (mod2, gates2) = mods[i]
            if i + 1 < len(blocks):
                (nw, nb), nmod, npad = blocks[i + 1].norm1_params(), mods[i + 1][0][0], blocks[i + 1].attn1.pads()[0]
            else:
                (nw, nb), nmod, npad = (_f32(self.norm_final.weight), _f32(self.norm_final.bias)), None, 0
            lean = None
            if self.lean_activations and self.training and torch.is_grad_enabled():
                lean = (*blk.norm1_params(), mods[i][0][0])
            args = (x, n, gates1, mod2, gates2, Lt, rope, nw, nb, nmod, cfg.norm_eps, npad, lean)
            if self.gradient_checkpointing and self.training and torch.is_grad_enabled() and i % getattr(self, "checkpoint_stride", 1) == 0:
                x, n = torch.utils.checkpoint.checkpoint(blk, *args, use_reentrant=False)
            else:
                x, n = blk(*args)

        hv = n[:, Lt:].contiguous()          # norm_final was applied to every token by the last fused pass
        m = F.linear(F.silu(emb), self.norm_out.linear.weight, self.norm_out.linear.bias)
        shift, scale = m.chunk(2, dim=1)
        mod = torch.stack([shift, 1 + scale, shift, 1 + scale], dim=1).float().contiguous().detach()
        hv = ops.ln_modulate(hv, _f32(self.norm_out.norm.weight), _f32(self.norm_out.norm.bias), mod, 0, cfg.norm_eps)
        hv = F.linear(hv, self.proj_out.weight, self.proj_out.bias)
        if cfg.patch_size_t is None:
            out = hv.reshape(B, Fr, H // p, W // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        else:
            pt = cfg.patch_size_t
            out = hv.reshape(B, (Fr + pt - 1) // pt, H // p, W // p, -1, pt, p, p)
            out = out.permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)
        if not return_dict:
            return (out,)
        return Transformer3DModelOutput(sample=out)
