"""Tolerance of the forward attention's lse2 (log2 of a row's softmax denominator) -- the contract since round 6.

The w1 forward accumulates the row sums on the matrix pipe from the bf16-ROUNDED weights, the same registers the PV product reads (`mfsum`):
    l~ = sum_j bf16(p_j)        p_j = exp2(s_j - M')
so that O = sum_j bf16(p_j) v_j / l~ is an exact convex combination of V rows.  Round-to-nearest-even to 8 significant bits gives bf16(p_j) = p_j (1 + d_j),
|d_j| <= 2^-8 (reached when the mantissa of p_j is just above 1.0), independent across keys, hence
    l~ / l - 1 = sum_j w_j d_j        w_j = p_j / l  (the exact softmax weights)
    worst case  |.| <= 2^-8                     (a one-hot row: 5.6e-3 in log2 units)
    typical     std <= 2^-8 / sqrt(3) * sqrt(sum_j w_j^2)   (a row spread over n keys: ~ 2.3e-3 / sqrt(n))
Rows redone by the online-softmax kernel (flagged strips) sum the unrounded weights in fp32 and sit far inside this.  The tolerance below is the smaller of the
worst case and six standard deviations, plus the fp32 floor the old fp32 row sums were held to."""
import math

import torch

LOG2E = 1.4426950408889634
RND = 2.0 ** -8


def lse2_tol(w, lse_ref, floor=3e-4, rel=2e-5, sigmas=6.0):
    """w: exact softmax weights (fp64, [..., Sq, Skv], rows summing to one); lse_ref [..., Sq] in log2 units -> per-row tolerance on |lse2 - lse_ref|"""
    conc = (w * w).sum(-1).sqrt()
    stat = sigmas * RND / math.sqrt(3.0) * conc
    worst = torch.full_like(conc, 1.02 * RND)
    return LOG2E * torch.minimum(stat, worst) + floor + rel * lse_ref.abs()
