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


# ---------------------------------------------------------------------------------------------------------------- activation-rounded mode
# `round_activations=True` ("rnd" below): the oracle keeps computing in its own precision (fp32 / fp64) but rounds to bf16 EXACTLY the tensors the
# HIP path stores in bf16, in the forward and -- for their incoming gradients -- in the backward; everything between two such points (GEMM
# accumulation, LayerNorm statistics, softmax, the loss) stays unrounded, as in the kernels (fp32 accumulators).  The rounding points follow
# videogpa_amd/transformer.py + ops.py + csrc/{residual_ln,qknorm,lora,attention_w1}.hip:
#   forward : time / text / patch embeddings and the modulation vectors as torch bf16 makes them (each linear / silu / `1 + scale` result rounded);
#             LN+modulate output n; LoRA down-projection T = n A^T (A, and s*B, cast to bf16 first); the fused projection output (base + adapter in ONE
#             fp32 accumulation, one rounding); normalised q (rounded AFTER the scale * log2(e) fold) and k; P where it multiplies V; the attention
#             output; gate * y and the residual sum (two roundings, csrc/residual_ln.hip); FF pre-activation, GELU output, FF output; norm_final /
#             norm_out / proj_out outputs; (pred - target) before the square (train/loss.py:73-77 under bf16 autocast).
#   backward: the gradient of every tensor above is rounded where the HIP path stores it (dy of each GEMM, dT, dq / dk / dv, dS inside attention,
#             d(qkv) behind the QK-norm backward, dx' of the fused residual+LN backward and gate * dx', dGELU); adapter gradients dA, dB are fp32
#             results of bf16 operands and are NOT rounded.
# What such an oracle can and cannot pin is measured in tests/test_gpu_cfg1.py (see its header).
class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def _rv(x):
    """value rounded to bf16, gradient passed straight through (weights cast to the activation dtype: their fp32 gradients are not rounded)"""
    return x + (x.detach().bfloat16().to(x.dtype) - x.detach())


def _rg(x):
    """identity whose incoming gradient is rounded to bf16"""
    return _RoundGrad.apply(x) if x.requires_grad else x


def _r(x, on=True):
    """a tensor the HIP path stores in bf16: value and incoming gradient rounded"""
    return _rg(_rv(x)) if on else x


def _lora_linear(x, sd, lora, name, lora_scale, rnd=False):
    if rnd:
        # ops.LoraExt: y = [x | T] [W | s B]^T + b in one fp32 accumulation, T = bf16(x A^T), A and s*B as bf16 copies of the fp32 adapters
        y = F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))
        ka = "base_model.model." + name + ".lora_A.weight"
        if lora is not None and ka in lora:
            A = _rv(lora[ka].to(x.dtype))
            sB = _rv(_rv(lora["base_model.model." + name + ".lora_B.weight"].to(x.dtype)) * lora_scale)
            y = y + F.linear(_r(F.linear(x, A)), sB)
        return _r(y)
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


def patch_embed(sd, cfg, text, video, rnd=False):
    if rnd:
        emb = patch_embed(sd, cfg, text, video)          # each of the two projections is one bf16 GEMM; the positional table is added in bf16
        if cfg.use_learned_positional_embeddings:
            pos = sd["patch_embed.pos_embedding"][:, : emb.shape[1]].to(emb.dtype)
            return _r(_r(emb - pos) + pos)
        return _r(emb)
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
    def forward(ctx, q, k, v, round_o=False, round_pds=True):
        """round_pds=False: NO rounding at all -- exact softmax(q k^T / sqrt(d)) v in the oracle's precision, head-chunked and recomputed in the
        backward so that the S x S matrix of a 17 776-token sequence (60 GB in fp32 over 48 heads) never exists (`chunked_attention` of forward();
        the full-depth parity tests).  Checked against F.scaled_dot_product_attention in tests/test_oracle_kat.py.
        round_o: the output is rounded to bf16 HERE, so that the backward's delta = rowsum(dO o O) is formed from the STORED bf16 output, as every
        flash-attention backward does (csrc/attention*.hip `attn_delta` / `w1_bwd_prep`, and torch's own bf16 kernels).  That matters: the identity
        rowsum(P o dP) = rowsum(dO o O) behind delta holds for the unrounded O = P V only; with the rounded O each row's dS no longer sums to zero,
        and dQ_i picks up -d(delta_i) * sum_j P_ij K_j -- a COHERENT term, not a random one, which dominates the heavily cancelling q / k gradients of
        the last block (measured: tools/cfg1_round_diag.py, profiles/r04_cfg1_round_diag_*.json)."""
        scale = q.shape[-1] ** -0.5
        o = torch.empty_like(q)
        rp = (lambda x: x.bfloat16().to(x.dtype)) if round_pds else (lambda x: x)
        for h0 in range(0, q.shape[1], _RoundedSDPA.CHUNK):
            sl = slice(h0, h0 + _RoundedSDPA.CHUNK)
            p = torch.softmax((q[:, sl] @ k[:, sl].transpose(-1, -2)) * scale, dim=-1)
            o[:, sl] = rp(p) @ v[:, sl]
            del p
        if round_o:
            o = o.bfloat16().to(o.dtype)
        ctx.round_pds = round_pds
        ctx.save_for_backward(q, k, v, o)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        scale = q.shape[-1] ** -0.5
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        rp = (lambda x: x.bfloat16().to(x.dtype)) if ctx.round_pds else (lambda x: x)
        for h0 in range(0, q.shape[1], _RoundedSDPA.CHUNK):
            sl = slice(h0, h0 + _RoundedSDPA.CHUNK)
            p = torch.softmax((q[:, sl] @ k[:, sl].transpose(-1, -2)) * scale, dim=-1)
            delta = (do[:, sl] * o[:, sl]).sum(-1, keepdim=True)
            dv[:, sl] = rp(p).transpose(-1, -2) @ do[:, sl]
            ds = (do[:, sl] @ v[:, sl].transpose(-1, -2)).sub_(delta).mul_(p)
            del p
            ds = rp(ds)
            dq[:, sl] = (ds @ k[:, sl]) * scale
            dk[:, sl] = (ds.transpose(-1, -2) @ q[:, sl]) * scale
            del ds
        return dq, dk, dv, None, None


def block_forward(sd, cfg, i, hid, enc, temb, lora=None, lora_scale=2.0, image_rotary_emb=None, capture=None, round_p_ds=False, rnd=False,
                  exact_delta=False, chunked=False):
    b = f"transformer_blocks.{i}."
    Lt = enc.shape[1]
    H, hd = cfg.num_attention_heads, cfg.attention_head_dim
    B = hid.shape[0]

    def ln_zero(name, hid, enc):
        m = _r(F.linear(_r(F.silu(temb), rnd), sd[b + name + ".linear.weight"], sd[b + name + ".linear.bias"]), rnd)
        shift, scale, gate, e_shift, e_scale, e_gate = m.chunk(6, dim=1)
        nw, nb = sd[b + name + ".norm.weight"], sd[b + name + ".norm.bias"]
        n_h = layer_norm(hid, nw, nb, cfg.norm_eps) * _r(1 + scale, rnd)[:, None] + shift[:, None]      # `1 + scale` is formed in the model dtype
        n_e = layer_norm(enc, nw, nb, cfg.norm_eps) * _r(1 + e_scale, rnd)[:, None] + e_shift[:, None]
        return _r(n_h, rnd), _r(n_e, rnd), gate[:, None], e_gate[:, None]

    def add(x, g, y):
        # csrc/residual_ln.hip: x' = bf16(x + bf16(gate * y)); backward dx' rounded once, dy = bf16(gate * bf16(dx'))
        return _r(x + _r(g * y)) if rnd else x + g * y

    n_h, n_e, gate, e_gate = ln_zero("norm1", hid, enc)
    x = torch.cat([n_e, n_h], dim=1)
    q = _lora_linear(x, sd, lora, b + "attn1.to_q", lora_scale, rnd)
    k = _lora_linear(x, sd, lora, b + "attn1.to_k", lora_scale, rnd)
    v = _lora_linear(x, sd, lora, b + "attn1.to_v", lora_scale, rnd)
    q, k, v = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    q = layer_norm(q, sd[b + "attn1.norm_q.weight"], sd[b + "attn1.norm_q.bias"], cfg.qk_norm_eps)
    k = layer_norm(k, sd[b + "attn1.norm_k.weight"], sd[b + "attn1.norm_k.bias"], cfg.qk_norm_eps)
    if image_rotary_emb is not None:
        cos, sin = image_rotary_emb
        q = torch.cat([q[:, :, :Lt], apply_rotary_emb(q[:, :, Lt:], cos, sin)], dim=2)
        k = torch.cat([k[:, :, :Lt], apply_rotary_emb(k[:, :, Lt:], cos, sin)], dim=2)
    if rnd:
        # csrc/qknorm.hip folds scale * log2(e) into q before its one rounding; the attention kernels return dq for the UNSCALED q, as bf16
        c = hd ** -0.5 * 1.4426950408889634
        q, k = _rg(_rv(q * c) / c), _r(k)
    if round_p_ds or rnd:
        o = _RoundedSDPA.apply(q, k, v, rnd and not exact_delta)
    elif chunked:
        o = _RoundedSDPA.apply(q, k, v, False, False)        # exact arithmetic, memory-bounded
    else:
        o = F.scaled_dot_product_attention(q, k, v)
    o = _r(o.transpose(1, 2).reshape(B, -1, H * hd), rnd)
    if capture is not None:
        capture.update(q=q, k=k, v=v, attn=o)
    o = _lora_linear(o, sd, lora, b + "attn1.to_out.0", lora_scale, rnd)
    hid = add(hid, gate, o[:, Lt:])
    enc = add(enc, e_gate, o[:, :Lt])

    n_h, n_e, gate, e_gate = ln_zero("norm2", hid, enc)
    x = torch.cat([n_e, n_h], dim=1)
    x = _r(F.gelu(_r(F.linear(x, sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"]), rnd), approximate="tanh"), rnd)
    x = _r(F.linear(x, sd[b + "ff.net.2.weight"], sd[b + "ff.net.2.bias"]), rnd)
    hid = add(hid, gate, x[:, Lt:])
    enc = add(enc, e_gate, x[:, :Lt])
    return hid, enc


def forward(sd, cfg, hidden_states, encoder_hidden_states, timestep, lora=None, lora_scale=2.0,
            image_rotary_emb=None, round_p_ds=False, round_activations=False, exact_delta=False, checkpoint_blocks=False, chunked_attention=False):
    """hidden_states [B,F,C,H,W], encoder_hidden_states [B,L,4096], timestep [B] -> sample [B,F,C_out,H,W].
    checkpoint_blocks: each transformer block under torch.utils.checkpoint (non-reentrant), as the reference trains
    (train/CogVideoX-5B/03_train.py:107-108) -- same values, one block's intermediates alive at a time; chunked_attention: the exact attention without
    the S x S matrix (see _RoundedSDPA round_pds=False).  Together they let the fp32 oracle run all 42 blocks at S = 17 776 on one 288 GB GPU.
    round_activations: see "activation-rounded mode" above (implies round_p_ds).  exact_delta (with round_activations): the attention backward's
    delta is formed from the UNROUNDED attention output -- the one place where the stored bf16 tensor is not what a precise backward wants."""
    B, Fr, C, H, W = hidden_states.shape
    p = cfg.patch_size
    D = cfg.inner_dim
    dt = hidden_states.dtype
    rnd = bool(round_activations)
    t_emb = _r(timestep_embedding(timestep, D, cfg.flip_sin_to_cos, cfg.freq_shift).to(dt), rnd)
    emb = _r(F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"]), rnd)
    emb = _r(F.linear(_r(F.silu(emb), rnd), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"]), rnd)

    x = patch_embed(sd, cfg, encoder_hidden_states, hidden_states, rnd)
    Lt = encoder_hidden_states.shape[1]
    enc, hid = x[:, :Lt], x[:, Lt:]
    for i in range(cfg.num_layers):
        def blk(hid, enc, i=i):
            return block_forward(sd, cfg, i, hid, enc, emb, lora, lora_scale, image_rotary_emb, round_p_ds=round_p_ds, rnd=rnd, exact_delta=exact_delta,
                                 chunked=chunked_attention)
        if checkpoint_blocks and torch.is_grad_enabled() and lora is not None:
            from torch.utils.checkpoint import checkpoint
            hid, enc = checkpoint(blk, hid, enc, use_reentrant=False)
        else:
            hid, enc = blk(hid, enc)

    hid = _r(layer_norm(torch.cat([enc, hid], dim=1), sd["norm_final.weight"], sd["norm_final.bias"], cfg.norm_eps), rnd)[:, Lt:]
    m = _r(F.linear(_r(F.silu(emb), rnd), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"]), rnd)
    shift, scale = m.chunk(2, dim=1)
    hid = _r(layer_norm(hid, sd["norm_out.norm.weight"], sd["norm_out.norm.bias"], cfg.norm_eps) * _r(1 + scale, rnd)[:, None] + shift[:, None], rnd)
    hid = _r(F.linear(hid, sd["proj_out.weight"], sd["proj_out.bias"]), rnd)
    if cfg.patch_size_t is None:
        out = hid.reshape(B, Fr, H // p, W // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
    else:
        pt = cfg.patch_size_t
        out = hid.reshape(B, (Fr + pt - 1) // pt, H // p, W // p, -1, pt, p, p)
        out = out.permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)
    return out


def dpo_pair_step(sd, cfg, lora, abar, x_win, x_lose, prompt_emb, t, noise, beta=1.0, lora_scale=2.0, round_p_ds=False, round_activations=False,
                  exact_delta=False, cond=None, checkpoint_blocks=False, chunked_attention=False):
    """One preference-pair step as train/CogVideoX-5B/03_train.py:116-157 does it.

    x_win/x_lose arrive as the dataset stores them, [B,C,F,H,W] (train/dataset.py:228-229), and are
    permuted to [B,F,C,H,W] (:120-121); win and lose share (t, noise) (:125-130); ref = same base
    weights without the adapter (:110-111,149-151).
    cond [B,F,Cc,H,W] (I2V, train/CogVideoX-I2V-5B/03_train.py:127-136): conditioning channels concatenated to BOTH noised latents after the
    noising (the zero-padded first-frame latent); the v-targets stay those of the 16 video channels.  The CogVideoX1.5 step (:118-186) is this
    function on a cfg with patch_size_t set and latents already even-cropped."""
    from . import dpo, scheduler
    xw = x_win.permute(0, 2, 1, 3, 4)
    xl = x_lose.permute(0, 2, 1, 3, 4)
    xw_n = scheduler.add_noise(abar, xw, noise, t)
    xl_n = scheduler.add_noise(abar, xl, noise, t)
    kw = dict(round_p_ds=round_p_ds, round_activations=round_activations, exact_delta=exact_delta, checkpoint_blocks=checkpoint_blocks,
              chunked_attention=chunked_attention)
    if round_activations:       # csrc/noise.hip: x_t and the v-target are bf16 tensors
        xw_n, xl_n = _rv(xw_n), _rv(xl_n)
    if cond is not None:
        xw_n, xl_n = torch.cat([xw_n, cond.to(xw_n.dtype)], dim=2), torch.cat([xl_n, cond.to(xl_n.dtype)], dim=2)
    v_w = forward(sd, cfg, xw_n, prompt_emb, t, lora, lora_scale, **kw)
    v_l = forward(sd, cfg, xl_n, prompt_emb, t, lora, lora_scale, **kw)
    with torch.no_grad():
        v_wr = forward(sd, cfg, xw_n, prompt_emb, t, None, **kw)
        v_lr = forward(sd, cfg, xl_n, prompt_emb, t, None, **kw)
    tw = scheduler.get_velocity(abar, xw, noise, t)
    tl = scheduler.get_velocity(abar, xl, noise, t)
    if round_activations:
        # bf16 predictions and targets; (pred - target) is formed in bf16 before the fp32 square (train/loss.py:73-77 under bf16 autocast): the
        # rounded difference replaces the prediction, the target becomes zero
        tw, tl = _rv(tw), _rv(tl)
        v_w_, v_l_, v_wr_, v_lr_ = (_r(a - b) for a, b in ((v_w, tw), (v_l, tl), (v_wr, tw), (v_lr, tl)))
        out = dpo.dpo_loss(v_w_, v_l_, v_wr_, v_lr_, torch.zeros_like(tw), torch.zeros_like(tl), beta=beta)
    else:
        out = dpo.dpo_loss(v_w, v_l, v_wr, v_lr, tw, tl, beta=beta)
    out.update(v_win=v_w, v_lose=v_l, v_win_ref=v_wr, v_lose_ref=v_lr)
    return out
