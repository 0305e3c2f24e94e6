"""CogVideoXTransformer3DModel + PEFT-LoRA forward, CPU oracle (plain torch, any float dtype).

PARITY UNPINNED: diffusers (requirements.txt:20) and peft (requirements.txt:24) are not
vendored in /root/reference and not installed; this restates their published algorithm
(diffusers models/transformers/cogvideox_transformer_3d.py, models/embeddings.py,
models/normalization.py, models/attention_processor.py::CogVideoXAttnProcessor2_0;
peft tuners/lora/layer.py::Linear.forward) and is anchored on the reference call sites
train/CogVideoX-5B/03_train.py:101-111,134-151 (no image_rotary_emb is passed there, so
RoPE is applied only when `image_rotary_emb` is given -- generate/CogVideoX-5B.py:72-77).

Functional style: `forward(sd, cfg, hidden_states, encoder_hidden_states, timestep, ...)`
where `sd` is a diffusers-named state dict and `lora` an optional
PEFT-named adapter dict (`base_model.model.transformer_blocks.N.attn1.to_q.lora_A.weight`).
Autograd works through it, so it is also the backward oracle.
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class CogVideoXConfig:
    num_attention_heads: int = 48
    attention_head_dim: int = 64
    in_channels: int = 16
    out_channels: int = 16
    num_layers: int = 42
    time_embed_dim: int = 512
    text_embed_dim: int = 4096
    patch_size: int = 2
    patch_size_t: Optional[int] = None
    patch_bias: bool = True
    sample_width: int = 90
    sample_height: int = 60
    sample_frames: int = 49
    temporal_compression_ratio: int = 4
    max_text_seq_length: int = 226
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    norm_eps: float = 1e-5
    use_rotary_positional_embeddings: bool = True
    use_learned_positional_embeddings: bool = False
    ff_mult: int = 4
    qk_norm_eps: float = 1e-6

    @property
    def inner_dim(self):
        return self.num_attention_heads * self.attention_head_dim


def init_state_dict(cfg: CogVideoXConfig, seed=0, dtype=torch.float32, std=0.02, mod_std=None):
    """Random-init weights with diffusers names/shapes."""
    g = torch.Generator().manual_seed(seed)
    D, Td = cfg.inner_dim, cfg.time_embed_dim
    mod_std = std if mod_std is None else mod_std

    def w(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    sd = {}
    p = cfg.patch_size
    if cfg.patch_size_t is None:
        sd["patch_embed.proj.weight"] = w(D, cfg.in_channels, p, p)
    else:
        sd["patch_embed.proj.weight"] = w(D, cfg.in_channels * p * p * cfg.patch_size_t)
    if cfg.patch_bias:
        sd["patch_embed.proj.bias"] = w(D)
    sd["patch_embed.text_proj.weight"] = w(D, cfg.text_embed_dim)
    sd["patch_embed.text_proj.bias"] = w(D)
    sd["time_embedding.linear_1.weight"] = w(Td, D)
    sd["time_embedding.linear_1.bias"] = w(Td)
    sd["time_embedding.linear_2.weight"] = w(Td, Td)
    sd["time_embedding.linear_2.bias"] = w(Td)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[b + n + ".linear.weight"] = w(6 * D, Td, s=mod_std)
            sd[b + n + ".linear.bias"] = w(6 * D, s=mod_std)
            sd[b + n + ".norm.weight"] = 1 + w(D)
            sd[b + n + ".norm.bias"] = w(D)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sd[b + f"attn1.{n}.weight"] = w(D, D)
            sd[b + f"attn1.{n}.bias"] = w(D)
        for n in ("norm_q", "norm_k"):
            sd[b + f"attn1.{n}.weight"] = 1 + w(cfg.attention_head_dim)
            sd[b + f"attn1.{n}.bias"] = w(cfg.attention_head_dim)
        sd[b + "ff.net.0.proj.weight"] = w(cfg.ff_mult * D, D)
        sd[b + "ff.net.0.proj.bias"] = w(cfg.ff_mult * D)
        sd[b + "ff.net.2.weight"] = w(D, cfg.ff_mult * D)
        sd[b + "ff.net.2.bias"] = w(D)
    sd["norm_final.weight"] = 1 + w(D)
    sd["norm_final.bias"] = w(D)
    sd["norm_out.linear.weight"] = w(2 * D, Td, s=mod_std)
    sd["norm_out.linear.bias"] = w(2 * D, s=mod_std)
    sd["norm_out.norm.weight"] = 1 + w(D)
    sd["norm_out.norm.bias"] = w(D)
    pt = cfg.patch_size_t or 1
    sd["proj_out.weight"] = w(p * p * pt * cfg.out_channels, D)
    sd["proj_out.bias"] = w(p * p * pt * cfg.out_channels)
    if cfg.use_learned_positional_embeddings:
        n_tok = cfg.max_text_seq_length + ((cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1) * \
            (cfg.sample_height // p) * (cfg.sample_width // p)
        sd["patch_embed.pos_embedding"] = w(1, n_tok, D)
    return sd


LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


def init_lora(cfg: CogVideoXConfig, r=64, seed=1, b_std=0.0, dtype=torch.float32, targets=LORA_TARGETS):
    """PEFT init: A ~ kaiming_uniform(a=sqrt(5)), B = 0 (b_std>0 gives N(0,b_std) for tests beyond step 0)."""
    g = torch.Generator().manual_seed(seed)
    D = cfg.inner_dim
    out = {}
    bound = 1.0 / math.sqrt(D)  # kaiming_uniform(a=sqrt5) on [r, D]: gain*sqrt(3/fan_in) = 1/sqrt(fan_in)
    for i in range(cfg.num_layers):
        for t in targets:
            k = f"base_model.model.transformer_blocks.{i}.attn1.{t}"
            out[k + ".lora_A.weight"] = ((torch.rand(r, D, generator=g) * 2 - 1) * bound).to(dtype)
            out[k + ".lora_B.weight"] = (torch.randn(D, r, generator=g) * b_std).to(dtype)
    return out


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def rope_3d_tables(num_frames, grid_h, grid_w, head_dim=64, theta=10000.0):
    """get_3d_rotary_pos_embed at native resolution: (cos, sin) each [F*h*w, head_dim] (SURVEY A-2)."""
    dim_t, dim_h, dim_w = head_dim // 4, head_dim // 8 * 3, head_dim // 8 * 3

    def axis(dim, n):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(n, dtype=torch.float32), freqs)
        return ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)

    ct, st = axis(dim_t, num_frames)
    ch, sh = axis(dim_h, grid_h)
    cw, sw = axis(dim_w, grid_w)

    def combine(t, h, w):
        t = t[:, None, None, :].expand(-1, grid_h, grid_w, -1)
        h = h[None, :, None, :].expand(num_frames, -1, grid_w, -1)
        w = w[None, None, :, :].expand(num_frames, grid_h, -1, -1)
        return torch.cat([t, h, w], dim=-1).reshape(num_frames * grid_h * grid_w, head_dim)

    return combine(ct, ch, cw), combine(st, sh, sw)


def apply_rotary_emb(x, cos, sin):
    """x [B,H,S,64]; interleaved-pair rotation (use_real_unbind_dim=-1), computed in fp32 like upstream."""
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos.float() + x_rot * sin.float()).to(x.dtype)


def _lora_linear(x, sd, lora, name, lora_scale):
    y = F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))
    if lora is not None:
        ka = "base_model.model." + name + ".lora_A.weight"
        if ka in lora:
            A = lora[ka].to(x.dtype)
            B = lora["base_model.model." + name + ".lora_B.weight"].to(x.dtype)
            y = y + F.linear(F.linear(x, A), B) * lora_scale
    return y


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def patch_embed(sd, cfg, text, video):
    B, Fr, C, H, W = video.shape
    p = cfg.patch_size
    text = F.linear(text, sd["patch_embed.text_proj.weight"], sd["patch_embed.text_proj.bias"])
    if cfg.patch_size_t is None:
        x = F.conv2d(video.reshape(B * Fr, C, H, W), sd["patch_embed.proj.weight"], sd.get("patch_embed.proj.bias"), stride=p)
        x = x.view(B, Fr, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
    else:
        pt = cfg.patch_size_t
        x = video.permute(0, 1, 3, 4, 2)
        x = x.reshape(B, Fr // pt, pt, H // p, p, W // p, p, C)
        x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)
        x = F.linear(x, sd["patch_embed.proj.weight"], sd.get("patch_embed.proj.bias"))
    emb = torch.cat([text, x], dim=1)
    if cfg.use_learned_positional_embeddings:
        emb = emb + sd["patch_embed.pos_embedding"][:, : emb.shape[1]].to(emb.dtype)
    return emb


class _RoundedSDPA(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v in the oracle's own precision EXCEPT for the two roundings every bf16 flash attention makes:
    the weights P are rounded to bf16 where they multiply V (forward) and dO (dV), and dS = P o (dP - delta) is rounded to bf16 where
    it multiplies K (dQ) and Q (dK); the softmax normaliser and P inside dS stay unrounded.  This is the "rounding-injected oracle"
    of tests/test_gpu_cfg1.py: what is left between it and the HIP path is the kernels' arithmetic, not their arithmetic TYPE.
    Heads are processed in chunks (the S x S matrix is recomputed in the backward, never stored)."""

    CHUNK = 4

    @staticmethod
    def forward(ctx, q, k, v):
        scale = q.shape[-1] ** -0.5
        o = torch.empty_like(q)
        for h0 in range(0, q.shape[1], _RoundedSDPA.CHUNK):
            sl = slice(h0, h0 + _RoundedSDPA.CHUNK)
            p = torch.softmax((q[:, sl] @ k[:, sl].transpose(-1, -2)) * scale, dim=-1)
            o[:, sl] = p.bfloat16().to(p.dtype) @ v[:, sl]
        ctx.save_for_backward(q, k, v, o)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        scale = q.shape[-1] ** -0.5
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for h0 in range(0, q.shape[1], _RoundedSDPA.CHUNK):
            sl = slice(h0, h0 + _RoundedSDPA.CHUNK)
            p = torch.softmax((q[:, sl] @ k[:, sl].transpose(-1, -2)) * scale, dim=-1)
            delta = (do[:, sl] * o[:, sl]).sum(-1, keepdim=True)
            ds = (p * (do[:, sl] @ v[:, sl].transpose(-1, -2) - delta)).bfloat16().to(p.dtype)
            dv[:, sl] = p.bfloat16().to(p.dtype).transpose(-1, -2) @ do[:, sl]
            dq[:, sl] = (ds @ k[:, sl]) * scale
            dk[:, sl] = (ds.transpose(-1, -2) @ q[:, sl]) * scale
        return dq, dk, dv


def block_forward(sd, cfg, i, hid, enc, temb, lora=None, lora_scale=2.0, image_rotary_emb=None, capture=None, round_p_ds=False):
    b = f"transformer_blocks.{i}."
    Lt = enc.shape[1]
    H, hd = cfg.num_attention_heads, cfg.attention_head_dim
    B = hid.shape[0]

    def ln_zero(name, hid, enc):
        m = F.linear(F.silu(temb), sd[b + name + ".linear.weight"], sd[b + name + ".linear.bias"])
        shift, scale, gate, e_shift, e_scale, e_gate = m.chunk(6, dim=1)
        nw, nb = sd[b + name + ".norm.weight"], sd[b + name + ".norm.bias"]
        n_h = layer_norm(hid, nw, nb, cfg.norm_eps) * (1 + scale)[:, None] + shift[:, None]
        n_e = layer_norm(enc, nw, nb, cfg.norm_eps) * (1 + e_scale)[:, None] + e_shift[:, None]
        return n_h, n_e, gate[:, None], e_gate[:, None]

    n_h, n_e, gate, e_gate = ln_zero("norm1", hid, enc)
    x = torch.cat([n_e, n_h], dim=1)
    q = _lora_linear(x, sd, lora, b + "attn1.to_q", lora_scale)
    k = _lora_linear(x, sd, lora, b + "attn1.to_k", lora_scale)
    v = _lora_linear(x, sd, lora, b + "attn1.to_v", lora_scale)
    q, k, v = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    q = layer_norm(q, sd[b + "attn1.norm_q.weight"], sd[b + "attn1.norm_q.bias"], cfg.qk_norm_eps)
    k = layer_norm(k, sd[b + "attn1.norm_k.weight"], sd[b + "attn1.norm_k.bias"], cfg.qk_norm_eps)
    if image_rotary_emb is not None:
        cos, sin = image_rotary_emb
        q = torch.cat([q[:, :, :Lt], apply_rotary_emb(q[:, :, Lt:], cos, sin)], dim=2)
        k = torch.cat([k[:, :, :Lt], apply_rotary_emb(k[:, :, Lt:], cos, sin)], dim=2)
    o = _RoundedSDPA.apply(q, k, v) if round_p_ds else F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, -1, H * hd)
    if capture is not None:
        capture.update(q=q, k=k, v=v, attn=o)
    o = _lora_linear(o, sd, lora, b + "attn1.to_out.0", lora_scale)
    hid = hid + gate * o[:, Lt:]
    enc = enc + e_gate * o[:, :Lt]

    n_h, n_e, gate, e_gate = ln_zero("norm2", hid, enc)
    x = torch.cat([n_e, n_h], dim=1)
    x = F.gelu(F.linear(x, sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"]), approximate="tanh")
    x = F.linear(x, sd[b + "ff.net.2.weight"], sd[b + "ff.net.2.bias"])
    hid = hid + gate * x[:, Lt:]
    enc = enc + e_gate * x[:, :Lt]
    return hid, enc


def forward(sd, cfg, hidden_states, encoder_hidden_states, timestep, lora=None, lora_scale=2.0,
            image_rotary_emb=None, round_p_ds=False):
    """hidden_states [B,F,C,H,W], encoder_hidden_states [B,L,4096], timestep [B] -> sample [B,F,C_out,H,W]."""
    B, Fr, C, H, W = hidden_states.shape
    p = cfg.patch_size
    D = cfg.inner_dim
    dt = hidden_states.dtype
    t_emb = timestep_embedding(timestep, D, cfg.flip_sin_to_cos, cfg.freq_shift).to(dt)
    emb = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])

    x = patch_embed(sd, cfg, encoder_hidden_states, hidden_states)
    Lt = encoder_hidden_states.shape[1]
    enc, hid = x[:, :Lt], x[:, Lt:]
    for i in range(cfg.num_layers):
        hid, enc = block_forward(sd, cfg, i, hid, enc, emb, lora, lora_scale, image_rotary_emb, round_p_ds=round_p_ds)

    hid = layer_norm(torch.cat([enc, hid], dim=1), sd["norm_final.weight"], sd["norm_final.bias"], cfg.norm_eps)[:, Lt:]
    m = F.linear(F.silu(emb), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
    shift, scale = m.chunk(2, dim=1)
    hid = layer_norm(hid, sd["norm_out.norm.weight"], sd["norm_out.norm.bias"], cfg.norm_eps) * (1 + scale)[:, None] + shift[:, None]
    hid = F.linear(hid, sd["proj_out.weight"], sd["proj_out.bias"])
    if cfg.patch_size_t is None:
        out = hid.reshape(B, Fr, H // p, W // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    else:
        pt = cfg.patch_size_t
        out = hid.reshape(B, (Fr + pt - 1) // pt, H // p, W // p, -1, pt, p, p)
        out = out.permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)
    return out


def dpo_pair_step(sd, cfg, lora, abar, x_win, x_lose, prompt_emb, t, noise, beta=1.0, lora_scale=2.0, round_p_ds=False):
    """One preference-pair step as train/CogVideoX-5B/03_train.py:116-157 does it.

    x_win/x_lose arrive as the dataset stores them, [B,C,F,H,W] (train/dataset.py:228-229), and are
    permuted to [B,F,C,H,W] (:120-121); win and lose share (t, noise) (:125-130); ref = same base
    weights without the adapter (:110-111,149-151)."""
    from . import dpo, scheduler
    xw = x_win.permute(0, 2, 1, 3, 4)
    xl = x_lose.permute(0, 2, 1, 3, 4)
    xw_n = scheduler.add_noise(abar, xw, noise, t)
    xl_n = scheduler.add_noise(abar, xl, noise, t)
    v_w = forward(sd, cfg, xw_n, prompt_emb, t, lora, lora_scale, round_p_ds=round_p_ds)
    v_l = forward(sd, cfg, xl_n, prompt_emb, t, lora, lora_scale, round_p_ds=round_p_ds)
    with torch.no_grad():
        v_wr = forward(sd, cfg, xw_n, prompt_emb, t, None, round_p_ds=round_p_ds)
        v_lr = forward(sd, cfg, xl_n, prompt_emb, t, None, round_p_ds=round_p_ds)
    tw = scheduler.get_velocity(abar, xw, noise, t)
    tl = scheduler.get_velocity(abar, xl, noise, t)
    out = dpo.dpo_loss(v_w, v_l, v_wr, v_lr, tw, tl, beta=beta)
    out.update(v_win=v_w, v_lose=v_l, v_win_ref=v_wr, v_lose_ref=v_lr)
    return out
