"""Parity + timing of the w1 attention kernels against the 2-waves-per-SIMD kernels of attention.hip (same C ABI).
    python tools/w1_check.py [--which dq] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="dq")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--big", type=int, default=1)
a = ap.parse_args()
which = set(a.which.split(","))


def make(B, H, S, seed=0, model_layout=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
    q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
    k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
    v = qkv[:, :, 2].permute(0, 2, 1, 3) if model_layout else qkv[:, :, 2].permute(0, 2, 1, 3).contiguous()
    do = torch.randn(B, S, H * 64, generator=g, device="cuda").to(torch.bfloat16)
    return q, k, v, do.view(B, S, H, 64).permute(0, 2, 1, 3)


def run(B, H, S, w1, seed=0):
    q, k, v, dov = make(B, H, S, seed)
    ops.ATTN_W1 = set()
    o, lse = ops.attention_fwd_raw(q, k, v, split_mode=0)
    ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
    dq, dk = torch.full_like(q, float("nan")), torch.full_like(k, float("nan"))
    dv = torch.full((B, S, H, 64), float("nan"), dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
    ops.ATTN_W1 = set(w1)
    ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv, split_mode=0)
    torch.cuda.synchronize()
    ops.ATTN_W1 = set()
    return dq, dk, dv


bad = 0
for (B, H, S) in [(1, 2, 128), (1, 1, 64), (2, 3, 100), (1, 2, 200), (1, 2, 777), (1, 3, 1024), (2, 2, 2500)]:
    ref = run(B, H, S, [])
    new = run(B, H, S, which)
    for name, r, n in zip(("dq", "dk", "dv"), ref, new):
        same = torch.equal(r, n)
        err = (r.float() - n.float()).abs().max().item()
        fin = bool(torch.isfinite(n.float()).all())
        nd = int((r != n).sum())
        print(f"B{B} H{H} S{S:5d} {name}: bitwise {same}  differing {nd}/{r.numel()}  max|diff| {err:.3e} finite {fin}  ref absmax {r.float().abs().max().item():.3e}")
        if not fin or err > 2e-2 * max(1.0, r.float().abs().max().item()):
            bad += 1
print("PARITY", "FAIL" if bad else "OK")

if a.big:
    B, H, S = 2, 48, 17776
    q, k, v, dov = make(B, H, S)
    o, lse = ops.attention_fwd_raw(q, k, v)
    ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    dv = torch.empty(B, S, H, 64, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
    res = {}
    for tag, w1 in (("old", set()), ("w1", which), ("old2", set()), ("w1b", which)):
        ops.ATTN_W1 = w1
        ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        for _ in range(a.iters):
            ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
        torch.cuda.synchronize()
        for name, s in ops.TIMER.summary().items():
            print(f"{tag:5s} {name:22s} avg {s['avg_ms']:8.3f} ms")
        ops.TIMER = None
        res[tag] = (dq.clone(), dk.clone(), dv.clone())
    for i, name in enumerate(("dq", "dk", "dv")):
        r, n = res["old"][i], res["w1"][i]
        print(f"headline {name}: bitwise {torch.equal(r, n)} max|diff| {(r.float() - n.float()).abs().max().item():.3e}")
