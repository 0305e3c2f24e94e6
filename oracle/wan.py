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
LoRA (PEFT Linear: y = W x + b + (alpha / r) B A x) is given as a dict  "<module path>" -> (A, B, scaling).

Activation-rounded mode (Params(round_activations=True), the Wan counterpart of oracle/cogvideox.py's): the arithmetic stays in the oracle's own precision but
EXACTLY the tensors videogpa_amd/wan_model.py stores in bf16 are rounded to bf16 in the forward, and their incoming gradients in the backward:
  forward : patch embedding output; text embedding (both linears and the GELU between them); block 0's normalised input (round_xhat: upstream's
            norm1(x).type_as(x) on the still-bf16 stream) and every LN + modulation output h; the LoRA down-projection T = h A^T (A and s B as bf16 copies) and the
            K-extended projection output (base + adapter, one accumulation, one rounding); RMS-norm: n = bf16(u rs), y = bf16(n w), RoPE result bf16; the softmax
            weights where they multiply V and the attention output; norm3 output; feed-forward pre-activation, GELU output, output.  The residual stream, the
            modulation table, gate * y and the whole output head stay unrounded (fp32 on the device).
  backward: the gradient of each of those tensors where the HIP path stores it in bf16 (dy of every GEMM incl. bf16(dx' * gate), dT, dq / dk / dv, dS inside
            attention, du behind the RMS-norm backward, dGELU); LayerNorm backward results and the residual-stream gradient stay unrounded; dA / dB are fp32.
  exact_delta: the attention backward's delta = rowsum(dO o O) from the UNROUNDED output (the HIP path's "Precise delta": ops.py) instead of the stored bf16 one.
  f8_attn    : the self-attention with the roundings of the hand-written e4m3 forward and of the backward that follows it (_F8Attn below)."""
import math

import torch
import torch.nn.functional as F

from .cogvideox import _RoundedSDPA, _r, _rv


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


def _f8_exp_of(amax):
    """csrc/attention_hd128.hip f8_exp_of: the smallest e with amax 2^-e <= 448 (frexp of amax / 448); 0 for amax = 0"""
    a32 = amax.to(torch.float32) * torch.tensor(1.0 / 448.0, dtype=torch.float32, device=amax.device)
    return torch.where(amax > 0, torch.frexp(a32)[1], torch.zeros_like(a32, dtype=torch.int32)).to(torch.int32)


def _e4m3(x):
    """round to nearest even onto the OCP e4m3 grid (|x| <= 448 by construction), in x's dtype"""
    return x.to(torch.float32).to(torch.float8_e4m3fn).to(x.dtype)


def f8_operands(q, k, v):
    """the operands the e4m3 forward really multiplies (attn128_f8_amax_kernel / attn128_f8_quant_kernel), DEquantised, [B, n, L, d] each:
         q8 = e4m3(q c 2^-eq) 2^eq  with c = d^-1/2 log2(e) folded in (scores in log2 units),   k8 = e4m3(k 2^-ek) 2^ek,   v8 = e4m3(v 2^-ev) 2^ev,
       one power of two per (batch, head) and tensor: e = f8_exp_of(max |.|)  (for q: of max |q| * c, both in fp32 as on the device)."""
    c = q.shape[-1] ** -0.5 * 1.4426950408889634
    c32 = torch.tensor(c, dtype=torch.float32, device=q.device)
    out = []
    for t, fold in ((q, True), (k, False), (v, False)):
        amax = t.detach().abs().amax(dim=(2, 3), keepdim=True).to(torch.float32)
        e = _f8_exp_of(amax * c32 if fold else amax)
        if fold:
            mul = torch.ldexp(c32.expand_as(amax).clone(), -e)          # ldexpf(c, -eq): one fp32 multiplier
        else:
            mul = torch.ldexp(torch.ones_like(amax), -e)
        t8 = _e4m3(t.to(torch.float32) * mul).to(t.dtype) * torch.ldexp(torch.ones_like(amax), e).to(t.dtype)
        out.append(t8)
    return out[0], out[1], out[2], c


class _F8Attn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v as videogpa_amd computes it with enable_fp8(attention=True), roundings INJECTED into the oracle's own precision.
    forward (csrc/attention_hd128.hip attn128_fwd_f8_kernel, tools/gen_w1_asm.py::Fwd128F8Loop):
      s = q8 k8^T (log2 units, exact products, wide accumulation);  b[i] = |q8 row i| max_j |k8 row j| * (1 + 2^-10) >= every score of the row (no running maximum);
      M[i] = b - floor(max(0, b - (m_s + 64))), m_s = the row's maximum over 64 keys spread evenly over the sweep (keys 0, L // 64, ...; sweeps of >= 128 keys) -- an
      INTEGER step off the bound (round 6: rows far under their bound no longer underflow; the e4m3 bits of the weights do not depend on the step);
      p = exp2(s - M);  l = sum_j p (unquantised);  per 64-key tile and row: x = frexp-exponent(tile sum) - 8 (E8M0 floor 2^-126), P8 = e4m3(p / 2^x) 2^x;
      O = (sum_j P8 v8) / l;  lse2 = M + log2(l).
    backward (vgpa_attn128_bwd_prescaled on the DEquantised operands the forward saved, all three exact in bf16 -- q8 itself (the query pre-scaled by
      c = d^-1/2 log2 e), k8, v8 -- so that its recomputed P = exp2(q8 k8 - lse2) IS the forward's p / l and every row sums to one; round 5 handed over
      q8 / c rounded to bf16, whose 2^-9 per element tilts a score of +-100 log2 units by several per cent of a weight): dV = bf16(P)^T dO, dP = dO v8^T, delta = rowsum(dO o O) with O the forward's
      own output (unrounded when exact_delta: "Precise delta"), dS = bf16(P o (dP - delta)), dq = d^-1/2 dS k8, dk = d^-1/2 dS^T q' -- the straight-through
      gradient of the quantised forward.  What the injected model leaves between itself and the device: fp32 accumulation order, exp2 / log2 in fp32, and the
      independent realisation of the P8 / bf16 rounding noise.
    consistent=False restates the ROUND-4 device backward instead (bf16 q, k, v with the e4m3 forward's lse2 and output): P no longer sums to one and dP is
      formed from values the forward did not use -- kept to measure what the change bought (tests/test_gpu_wan_cfg1.py)."""

    CHUNK = 2

    @staticmethod
    def _fwd_head(q8, k8, v8, lq_norm=False):
        """[.., L, d] -> O, lse2 (log2 units), for a chunk of heads.  lq_norm (an experiment, NOT what the device does): normalise by the sum of the
        QUANTISED weights, which makes O an exact convex combination of the v8 rows"""
        s = q8 @ k8.transpose(-1, -2)
        qn = q8.pow(2).sum(-1, keepdim=True).sqrt()
        kmax = k8.pow(2).sum(-1).amax(dim=-1, keepdim=True).sqrt().unsqueeze(-1)
        M = qn * kmax * 1.0009765625
        Lk = s.shape[-1]
        if Lk >= 128:                                                            # W1H_SAMPLE_KEYS = 64, W1H_SAMPLE_UP = 64 (csrc/attention_hd128.hip)
            ms = s[..., torch.arange(64, device=s.device) * (Lk // 64)].amax(dim=-1, keepdim=True)
            M = M - torch.floor(torch.clamp(M - (ms + 64.0), min=0.0))
        p = torch.exp2(s - M)
        l = p.sum(-1, keepdim=True)
        Lk = p.shape[-1]
        pad = (-Lk) % 64
        pt = F.pad(p, (0, pad)).unflatten(-1, (-1, 64))                          # [.., L, tiles, 64]
        ts = pt.sum(-1, keepdim=True)
        e = torch.frexp(ts.to(torch.float32))[1]                                 # tile sum in [2^(e-1), 2^e)
        x = torch.clamp(e - 8, min=-126).to(p.dtype)
        sc = torch.exp2(x)
        p8 = (_e4m3((pt / sc).clamp(max=448.0)) * sc).flatten(-2)[..., :Lk]
        o = (p8 @ v8) / (p8.sum(-1, keepdim=True) if lq_norm else l)
        return o, M + torch.log2(l)

    @staticmethod
    def forward(ctx, q, k, v, exact_delta=True, consistent=True, lq_norm=False):
        """q, k, v [B, n, L, d] (bf16-valued).  -> O [B, n, L, d], unrounded; the caller rounds it to bf16 like every stored activation"""
        q8, k8, v8, c = f8_operands(q, k, v)
        o = torch.empty_like(q)
        lse2 = q.new_empty(q.shape[:-1] + (1,))
        for h0 in range(0, q.shape[1], _F8Attn.CHUNK):
            sl = slice(h0, h0 + _F8Attn.CHUNK)
            o[:, sl], lse2[:, sl] = _F8Attn._fwd_head(q8[:, sl], k8[:, sl], v8[:, sl], lq_norm)
        o_delta = o if exact_delta else o.bfloat16().to(o.dtype)
        if consistent:
            ctx.save_for_backward(q8 / c, k8, v8, o_delta, lse2)          # q8 / c UNrounded: the device keeps q8 and multiplies by 1 where this multiplies by c
        else:
            ctx.save_for_backward(q, k, v, o_delta, lse2)
        ctx.c = c
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse2 = ctx.saved_tensors
        c, scale = ctx.c, q.shape[-1] ** -0.5
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for h0 in range(0, q.shape[1], _F8Attn.CHUNK):
            sl = slice(h0, h0 + _F8Attn.CHUNK)
            p = torch.exp2((q[:, sl] @ k[:, sl].transpose(-1, -2)) * c - lse2[:, sl])
            delta = (do[:, sl] * o[:, sl]).sum(-1, keepdim=True)
            ds = (p * (do[:, sl] @ v[:, sl].transpose(-1, -2) - delta)).bfloat16().to(p.dtype)
            dv[:, sl] = p.bfloat16().to(p.dtype).transpose(-1, -2) @ do[:, sl]
            dq[:, sl] = (ds @ k[:, sl]) * scale
            dk[:, sl] = (ds.transpose(-1, -2) @ q[:, sl]) * scale
        return dq, dk, dv, None, None, None


class Params:
    """state dict + LoRA lookup.  fp8_ffn: the feed-forward of every block with e4m3 GEMM operands (see _Fp8Ffn).  round_activations / exact_delta / f8_attn:
    the activation-rounded mode of the module docstring (f8_attn: False, True = the device's forward + consistent backward, "r4" = the round-4 device backward,
    "lq" = an experiment: output normalised by the sum of the quantised weights)."""

    def __init__(self, state, lora=None, dtype=torch.float32, fp8_ffn=False, round_activations=False, exact_delta=True, f8_attn=False, f8_min_keys=1024,
                 checkpoint_blocks=False, chunked_attention=False):
        """checkpoint_blocks: every block under torch.utils.checkpoint, as the reference trains (train/Wan2.2-TI2V-5B/03_train.py:151-160) -- same values, one
        block's intermediates alive at a time; chunked_attention: the exact attention head-chunked and recomputed in the backward (oracle/cogvideox.py
        _RoundedSDPA with round_pds=False) so that the L x L matrix of an 18 480-token sample never exists.  Together: the full 30-block model in fp32 on one GPU
        (tests/test_gpu_depth_wan.py; pinned to the plain form in tests/test_oracle_kat.py)."""
        self.checkpoint_blocks, self.chunked_attention = bool(checkpoint_blocks), bool(chunked_attention)
        self.s = {k: (v if v.dtype == dtype else v.detach().to(dtype)) for k, v in state.items()}
        self.lora = lora or {}
        self.dtype = dtype
        self.fp8_ffn = fp8_ffn
        self.rnd = bool(round_activations)
        self.exact_delta = bool(exact_delta)
        self.f8_attn = f8_attn
        self.f8_min_keys = f8_min_keys        # ops.ATTN128_F8_MIN_KEYS: shorter key sweeps (the cross-attention) stay on the bf16 kernels

    def linear(self, name, x, f32=False):
        """f32: a projection the HIP path runs in fp32 (time embedding / projection, output head) -- never rounded"""
        y = F.linear(x, self.s[name + ".weight"], self.s.get(name + ".bias"))
        if self.rnd and not f32:
            # ops.LoraExt: y = [x | T] [W | s B]^T + b in ONE accumulation, T = bf16(x A^T), A and s B as bf16 copies of the fp32 adapters
            if name in self.lora:
                A, Bm, sc = self.lora[name]
                y = y + F.linear(_r(F.linear(x, _rv(A.to(x.dtype)))), _rv(_rv(Bm.to(x.dtype)) * sc))
            return _r(y)
        if name in self.lora:
            A, Bm, sc = self.lora[name]
            y = y + F.linear(F.linear(x, A.to(x.dtype)), Bm.to(x.dtype)) * sc
        return y

    def r(self, x):
        """a tensor the HIP path stores in bf16 (value and incoming gradient rounded) -- identity outside the activation-rounded mode"""
        return _r(x) if self.rnd else x

    def rv(self, x):
        return _rv(x) if self.rnd else x

    def __getitem__(self, k):
        return self.s[k]


def _attend(q, k, v, P=None, self_attn=False):
    """[B, Lq, n, d] x [B, Lk, n, d] -> [B, Lq, n d]; flash_attention's default scale d^-0.5, no mask"""
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if P is not None and P.rnd:
        if self_attn and P.f8_attn and k.shape[2] >= P.f8_min_keys:
            o = _F8Attn.apply(q, k, v, P.exact_delta, P.f8_attn != "r4", P.f8_attn == "lq")
        else:
            o = _RoundedSDPA.apply(q, k, v, not P.exact_delta)
        return _r(o.transpose(1, 2).flatten(2))
    if P is not None and P.chunked_attention:
        return _RoundedSDPA.apply(q, k, v, False, False).transpose(1, 2).flatten(2)
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), dim=-1)
    return (p @ v).transpose(1, 2).flatten(2)


def _qk_norm(P, u, w, eps):
    """WanRMSNorm; rounded mode = csrc/wan.hip wan_rms_rope_fwd_kernel: n = bf16(u rs), y = bf16(n w) (the RoPE that may follow is rounded by its caller)"""
    if not P.rnd:
        return rms_norm(u, w, eps)
    return _rv(_rv(u * torch.rsqrt(u.pow(2).mean(dim=-1, keepdim=True) + eps)) * w)


def self_attention(P, pre, x, n, grid, freqs, eps):
    B, L, C = x.shape
    d = C // n
    q = _qk_norm(P, P.linear(pre + ".q", x), P[pre + ".norm_q.weight"], eps).view(B, L, n, d)
    k = _qk_norm(P, P.linear(pre + ".k", x), P[pre + ".norm_k.weight"], eps).view(B, L, n, d)
    v = P.linear(pre + ".v", x).view(B, L, n, d)
    return P.linear(pre + ".o", _attend(P.r(rope_apply(q, grid, freqs)), P.r(rope_apply(k, grid, freqs)), v, P, self_attn=True))


def cross_attention(P, pre, x, context, n, eps):
    B, L, C = x.shape
    d = C // n
    q = P.r(_qk_norm(P, P.linear(pre + ".q", x), P[pre + ".norm_q.weight"], eps)).view(B, L, n, d)
    k = P.r(_qk_norm(P, P.linear(pre + ".k", context), P[pre + ".norm_k.weight"], eps)).view(B, -1, n, d)
    v = P.linear(pre + ".v", context).view(B, -1, n, d)
    return P.linear(pre + ".o", _attend(q, k, v, P))


def block(P, pre, x, e, n, grid, freqs, context, eps, cross_attn_norm=True, first=False):
    """e [B, L, 6, C].  first (rounded mode only): block 0 normalises the still-bf16 patch embedding, upstream's norm1(x).type_as(x) rounds the normalised value"""
    e = (P[pre + ".modulation"].unsqueeze(0) + e).chunk(6, dim=2)
    xh = layer_norm(x, eps)
    if first and P.rnd:
        xh = _rv(xh)
    y = self_attention(P, pre + ".self_attn", P.r(xh * (1 + e[1].squeeze(2)) + e[0].squeeze(2)), n, grid, freqs, eps)
    x = x + y * e[2].squeeze(2)
    h = P.r(layer_norm(x, eps, P[pre + ".norm3.weight"], P[pre + ".norm3.bias"])) if cross_attn_norm else P.r(x)
    x = x + cross_attention(P, pre + ".cross_attn", h, context, n, eps)
    y = layer_norm(x, eps) * (1 + e[4].squeeze(2)) + e[3].squeeze(2)
    if P.fp8_ffn:
        y = _Fp8Ffn.apply(y, P[pre + ".ffn.0.weight"], P[pre + ".ffn.0.bias"], P[pre + ".ffn.2.weight"], P[pre + ".ffn.2.bias"])
        # the gate's backward writes bf16(gate * dout) straight as the e4m3 operand: _Fp8Ffn.backward rounds its incoming gradient, which is that product
    else:
        y = P.linear(pre + ".ffn.2", P.r(F.gelu(P.linear(pre + ".ffn.0", P.r(y)), approximate="tanh")))
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
    x = P.r(x.flatten(2).transpose(1, 2))                                 # [B, L, dim]
    L = x.shape[1]
    assert L == seq_len
    if t.dim() == 1:
        t = t[:, None].expand(B, seq_len)
    e = sinusoidal_embedding_1d(cfg["freq_dim"], t.reshape(-1)).to(dt).view(B, seq_len, -1)
    e = P.linear("time_embedding.2", F.silu(P.linear("time_embedding.0", e, f32=True)), f32=True)          # [B, L, dim]
    e0 = P.linear("time_projection.1", F.silu(e), f32=True).unflatten(2, (6, dim))                          # [B, L, 6, dim]
    ctx = torch.stack([torch.cat([u, u.new_zeros(cfg["text_len"] - u.size(0), u.size(1))]) for u in context_list]).to(dt)
    ctx = P.linear("text_embedding.2", P.r(F.gelu(P.linear("text_embedding.0", ctx), approximate="tanh")))
    d = dim // n
    dev = xb.device
    freqs = torch.cat([rope_params(1024, d - 4 * (d // 6), device=dev), rope_params(1024, 2 * (d // 6), device=dev), rope_params(1024, 2 * (d // 6), device=dev)], dim=1)
    for i in range(cfg["num_layers"]):
        def blk(x, i=i):
            return block(P, f"blocks.{i}", x, e0, n, grid, freqs, ctx, eps, cfg.get("cross_attn_norm", True), first=(i == 0))
        if P.checkpoint_blocks and torch.is_grad_enabled() and P.lora:
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(blk, x, use_reentrant=False)
        else:
            x = blk(x)
    em = (P["head.modulation"].unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
    x = P.linear("head.head", layer_norm(x, eps) * (1 + em[1].squeeze(2)) + em[0].squeeze(2), f32=True)
    c = cfg["out_dim"]
    out = []
    for u in x:
        u = u[: f * h * w].view(f, h, w, pt, ph, pw, c)
        out.append(torch.einsum("fhwpqrc->cfphqwr", u).reshape(c, f * pt, h * ph, w * pw))
    return out
