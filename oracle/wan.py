"""TEST INFRASTRUCTURE -- plain-torch restatement of the Wan2.2 denoiser (`wan.modules.model.WanModel`), only imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline.

PARITY UNPINNED: the reference imports WanModel from a sibling checkout of the Wan2.2 repository (train/Wan2.2-TI2V-5B/03_train.py:43-48,
`sys.path.insert(0, '../../Wan2.2')`) that is not vendored under /root/reference and is not installed here, so there are no golden
vectors and nothing to import.  What follows restates the PUBLISHED architecture (Wan2.2 wan/modules/model.py, as released with the
TI2V-5B checkpoint: dim 3072, ffn 14336, 24 heads, 30 layers, in/out 48, patch (1,2,2), text_len 512, per-token timesteps), written
the straightforward way -- per-token modulation tensors [B, L, 6, C], complex RoPE in float64, dense softmax -- so that it shares no
structure with videogpa_amd/wan_model.py.  Everything runs in the dtype of the parameters handed in (tests use fp32 / fp64 copies).

  sinusoidal_embedding_1d, rope_params, rope_apply      upstream helpers of the same names
  rms_norm / layer_norm                                  WanRMSNorm (eps inside the sqrt, weight after the cast) / WanLayerNorm
  self_attention, cross_attention, block, head, forward  WanSelfAttention, WanCrossAttention (t2v form), WanAttentionBlock, Head, WanModel.forward
LoRA (PEFT Linear: y = W x + b + (alpha / r) B A x) is given as a dict  "<module path>" -> (A, B, scaling)."""
import math

import torch
import torch.nn.functional as F


def sinusoidal_embedding_1d(dim, position):
    half = dim // 2
    position = position.to(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, dtype=torch.float64, device=position.device).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_params(max_seq_len, dim, theta=10000, device=None):
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len, dtype=torch.float64, device=device),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64, device=device).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_apply(x, grid, freqs):
    """x [B, L, n, d]; every sample on the same (f, h, w) grid with L = f h w"""
    B, L, n, d = x.shape
    c = d // 2
    fr = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    f, h, w = grid
    xc = torch.view_as_complex(x.to(torch.float64).reshape(B, L, n, c, 2))
    fi = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                    fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
    return torch.view_as_real(xc * fi).flatten(3).to(x.dtype)


def rms_norm(x, weight, eps):
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def layer_norm(x, eps, weight=None, bias=None):
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def _q8_rows(x):
    """per-row dynamic OCP e4m3 quantisation as csrc/fp8.hip does it (scale = amax / 448, round to nearest even, saturating), returned DEquantised in
    x's dtype: the value the e4m3 GEMM operand stands for"""
    fmax = 448.0
    amax = x.abs().amax(dim=-1, keepdim=True)
    sc = torch.where(amax > 0, amax / fmax, torch.ones_like(amax))
    return (x / sc).clamp(-fmax, fmax).to(torch.float32).to(torch.float8_e4m3fn).to(x.dtype) * sc


def _bf(x):
    return x.bfloat16().to(x.dtype)


class _Fp8Ffn(torch.autograd.Function):
    """The feed-forward branch y = W2 gelu(W1 h + b1) + b2 with the roundings of videogpa_amd/wan_model.py::_FfnFp8Fn INJECTED into the oracle's own
    precision (the "fp8 MFMA path" of BASELINE configs[4]; frozen weights, so the backward is dX only):
      forward : h -> bf16 -> e4m3 rows; W1, W2 e4m3 per output row; u = bf16(h8 W1_8^T + b1); g = bf16(gelu(u)) -> e4m3 rows; y = bf16(g8 W2_8^T + b2)
      backward: dy -> bf16 -> e4m3 rows; dg = bf16(dy8 (W2^T)_8^T) with W2^T quantised per ITS rows; du = bf16(dg gelu'(u)) -> e4m3 rows; dh = bf16(du8 (W1^T)_8^T)
    GEMM accumulation itself stays in the oracle's precision (the fp8 MFMAs accumulate in fp32)."""

    @staticmethod
    def forward(ctx, h, W1, b1, W2, b2):
        h8 = _q8_rows(_bf(h))
        u = _bf(F.linear(h8, _q8_rows(W1), b1))
        g8 = _q8_rows(_bf(F.gelu(u, approximate="tanh")))
        y = _bf(F.linear(g8, _q8_rows(W2), b2))
        ctx.save_for_backward(u, W1, W2)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, W1, W2 = ctx.saved_tensors
        dg = _bf(_q8_rows(_bf(dy)) @ _q8_rows(W2.t().contiguous()).t())
        with torch.enable_grad():
            uu = u.detach().requires_grad_(True)
            (gp,) = torch.autograd.grad(F.gelu(uu, approximate="tanh").sum(), uu)
        du8 = _q8_rows(_bf(dg * gp))
        dh = _bf(du8 @ _q8_rows(W1.t().contiguous()).t())
        return dh, None, None, None, None


class Params:
    """state dict + LoRA lookup.  fp8_ffn: the feed-forward of every block with e4m3 GEMM operands (see _Fp8Ffn)."""

    def __init__(self, state, lora=None, dtype=torch.float32, fp8_ffn=False):
        self.s = {k: v.detach().to(dtype) for k, v in state.items()}
        self.lora = lora or {}
        self.dtype = dtype
        self.fp8_ffn = fp8_ffn

    def linear(self, name, x):
        y = F.linear(x, self.s[name + ".weight"], self.s.get(name + ".bias"))
        if name in self.lora:
            A, Bm, sc = self.lora[name]
            y = y + F.linear(F.linear(x, A.to(x.dtype)), Bm.to(x.dtype)) * sc
        return y

    def __getitem__(self, k):
        return self.s[k]


def _attend(q, k, v):
    """[B, Lq, n, d] x [B, Lk, n, d] -> [B, Lq, n d]; flash_attention's default scale d^-0.5, no mask"""
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), dim=-1)
    return (p @ v).transpose(1, 2).flatten(2)


def self_attention(P, pre, x, n, grid, freqs, eps):
    B, L, C = x.shape
    d = C // n
    q = rms_norm(P.linear(pre + ".q", x), P[pre + ".norm_q.weight"], eps).view(B, L, n, d)
    k = rms_norm(P.linear(pre + ".k", x), P[pre + ".norm_k.weight"], eps).view(B, L, n, d)
    v = P.linear(pre + ".v", x).view(B, L, n, d)
    return P.linear(pre + ".o", _attend(rope_apply(q, grid, freqs), rope_apply(k, grid, freqs), v))


def cross_attention(P, pre, x, context, n, eps):
    B, L, C = x.shape
    d = C // n
    q = rms_norm(P.linear(pre + ".q", x), P[pre + ".norm_q.weight"], eps).view(B, L, n, d)
    k = rms_norm(P.linear(pre + ".k", context), P[pre + ".norm_k.weight"], eps).view(B, -1, n, d)
    v = P.linear(pre + ".v", context).view(B, -1, n, d)
    return P.linear(pre + ".o", _attend(q, k, v))


def block(P, pre, x, e, n, grid, freqs, context, eps, cross_attn_norm=True):
    """e [B, L, 6, C]"""
    e = (P[pre + ".modulation"].unsqueeze(0) + e).chunk(6, dim=2)
    y = self_attention(P, pre + ".self_attn", layer_norm(x, eps) * (1 + e[1].squeeze(2)) + e[0].squeeze(2), n, grid, freqs, eps)
    x = x + y * e[2].squeeze(2)
    h = layer_norm(x, eps, P[pre + ".norm3.weight"], P[pre + ".norm3.bias"]) if cross_attn_norm else x
    x = x + cross_attention(P, pre + ".cross_attn", h, context, n, eps)
    y = layer_norm(x, eps) * (1 + e[4].squeeze(2)) + e[3].squeeze(2)
    if P.fp8_ffn:
        y = _Fp8Ffn.apply(y, P[pre + ".ffn.0.weight"], P[pre + ".ffn.0.bias"], P[pre + ".ffn.2.weight"], P[pre + ".ffn.2.bias"])
        # the gate's backward writes bf16(gate * dout) straight as the e4m3 operand: _Fp8Ffn.backward rounds its incoming gradient, which is that product
    else:
        y = P.linear(pre + ".ffn.2", F.gelu(P.linear(pre + ".ffn.0", y), approximate="tanh"))
    return x + y * e[5].squeeze(2)


def forward(P, cfg, x_list, t, context_list, seq_len):
    """cfg: dict(patch_size, text_len, dim, freq_dim, out_dim, num_heads, num_layers, eps, cross_attn_norm) -> list of [C_out, F, H, W]"""
    dt = P.dtype
    dim, n, eps = cfg["dim"], cfg["num_heads"], cfg["eps"]
    pt, ph, pw = cfg["patch_size"]
    xb = torch.stack(list(x_list)).to(dt)
    B = xb.shape[0]
    x = F.conv3d(xb, P["patch_embedding.weight"], P["patch_embedding.bias"], stride=(pt, ph, pw))
    f, h, w = x.shape[2:]
    grid = (f, h, w)
    x = x.flatten(2).transpose(1, 2)                                      # [B, L, dim]
    L = x.shape[1]
    assert L == seq_len
    if t.dim() == 1:
        t = t[:, None].expand(B, seq_len)
    e = sinusoidal_embedding_1d(cfg["freq_dim"], t.reshape(-1)).to(dt).view(B, seq_len, -1)
    e = P.linear("time_embedding.2", F.silu(P.linear("time_embedding.0", e)))          # [B, L, dim]
    e0 = P.linear("time_projection.1", F.silu(e)).unflatten(2, (6, dim))               # [B, L, 6, dim]
    ctx = torch.stack([torch.cat([u, u.new_zeros(cfg["text_len"] - u.size(0), u.size(1))]) for u in context_list]).to(dt)
    ctx = P.linear("text_embedding.2", F.gelu(P.linear("text_embedding.0", ctx), approximate="tanh"))
    d = dim // n
    dev = xb.device
    freqs = torch.cat([rope_params(1024, d - 4 * (d // 6), device=dev), rope_params(1024, 2 * (d // 6), device=dev), rope_params(1024, 2 * (d // 6), device=dev)], dim=1)
    for i in range(cfg["num_layers"]):
        x = block(P, f"blocks.{i}", x, e0, n, grid, freqs, ctx, eps, cfg.get("cross_attn_norm", True))
    em = (P["head.modulation"].unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
    x = P.linear("head.head", layer_norm(x, eps) * (1 + em[1].squeeze(2)) + em[0].squeeze(2))
    c = cfg["out_dim"]
    out = []
    for u in x:
        u = u[: f * h * w].view(f, h, w, pt, ph, pw, c)
        out.append(torch.einsum("fhwpqrc->cfphqwr", u).reshape(c, f * pt, h * ph, w * pw))
    return out
