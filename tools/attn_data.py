"""Attention operands with the structure a TRAINED QK-normed denoiser produces, for timing the forward away from the friendliest case (bench.py's
random-init weights give QK-norm gains of 1 and nearly flat scores; the w1 forward shifts by a row BOUND, so its speed depends on how far below |q| max|k| the row
maxima lie).  trained_like_qkv(): q, k as LayerNorm(head_dim) outputs with a per-channel affine of mean `gain` (a few channels 3 x larger, as trained norms have),
every query correlated with ONE key at cosine `peak` (low-entropy rows), `sinks` keys per head scaled to `sink_norm` x the typical key norm (attention sinks).
Returns bf16 [B,H,S,64] q (NOT pre-scaled), k, v and a dict of statistics of the score rows (log2 units, a sample of rows of head 0)."""
import math

import torch


def trained_like_qkv(B, H, S, gain=3.0, peak=0.6, sinks=4, sink_norm=1.0, seed=0, device="cuda", d=64):
    g = torch.Generator(device=device).manual_seed(seed)
    ln = lambda x: torch.nn.functional.layer_norm(x, (d,))
    wq = gain * (1 + 0.2 * torch.randn(d, generator=g, device=device))
    wk = gain * (1 + 0.2 * torch.randn(d, generator=g, device=device))
    wq[:3] *= 3.0
    wk[:3] *= 3.0                                      # a few outlier channels
    bq, bk = 0.1 * gain * torch.randn(d, generator=g, device=device), 0.1 * gain * torch.randn(d, generator=g, device=device)
    xk = torch.randn(B, H, S, d, generator=g, device=device)
    perm = torch.randint(0, S, (S,), generator=g, device=device)
    xq = peak * xk[:, :, perm] + math.sqrt(max(0.0, 1 - peak * peak)) * torch.randn(B, H, S, d, generator=g, device=device)
    q = ln(xq) * wq + bq
    k = ln(xk) * wk + bk
    if sinks and sink_norm != 1.0:
        idx = torch.randint(0, S, (sinks,), generator=g, device=device)
        k[:, :, idx] *= sink_norm
    v = torch.randn(B, H, S, d, generator=g, device=device)
    q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    rows = torch.randint(0, S, (256,), generator=g, device=device)
    c = d ** -0.5 * 1.4426950408889634
    s = (q[0, 0, rows].float() @ k[0, 0].float().t()) * c                      # log2 units
    bound = q[0, 0, rows].float().norm(dim=-1) * k[0, 0].float().norm(dim=-1).max() * c
    p = torch.softmax(s * math.log(2.0), dim=-1)
    ent = -(p * torch.log2(p.clamp_min(1e-30))).sum(-1)
    stats = {"bound_log2_mean": float(bound.mean()), "bound_log2_max": float(bound.max()), "gap_bound_minus_rowmax_mean": float((bound - s.max(-1).values).mean()),
             "gap_bound_minus_rowmax_max": float((bound - s.max(-1).values).max()), "row_entropy_bits_mean": float(ent.mean()), "uniform_entropy_bits": math.log2(S)}
    return q, k, v, stats
