"""Time one attention kernel family at the headline shape with whatever library VGPA_LIB selects (timing only).
    VGPA_LIB=var/lib_w1_X.so python tools/w1_time.py --which dq [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="dq")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--tag", default="")
a = ap.parse_args()
B, H, S = 2, 48, 17776
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
v = qkv[:, :, 2].permute(0, 2, 1, 3)
do = torch.randn(B, S, H * 64, generator=g, device="cuda").to(torch.bfloat16)
dov = do.view(B, S, H, 64).permute(0, 2, 1, 3)
ops.ATTN_W1 = set(a.which.split(","))
o, lse = ops.attention_fwd_raw(q, k, v)
ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
dq, dk = torch.empty_like(q), torch.empty_like(k)
dv = torch.empty(B, S, H, 64, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
for _ in range(a.iters):
    if "fwd" in a.which:
        ops.attention_fwd_raw(q, k, v)
    if "dq" in a.which or "dkv" in a.which:
        ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
torch.cuda.synchronize()
tag = a.tag or os.path.basename(os.environ.get("VGPA_LIB", "product"))
for name, s in ops.TIMER.summary().items():
    if name.startswith("attn_") and name != "attn_delta_kernel":
        print(f"{tag:28s} {name:22s} avg {s['avg_ms']:8.3f} ms")
