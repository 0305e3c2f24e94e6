"""Stand-alone timing of the attention kernels at the headline shape (2 sequences x 48 heads x S=17776 x 64).
    python tools/attn_bench.py [--iters 5] [--S 17776] [--B 2] [--H 48] [--which fwd,dkv,dq]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--S", type=int, default=17776)
ap.add_argument("--B", type=int, default=2)
ap.add_argument("--H", type=int, default=48)
ap.add_argument("--which", default="fwd,dkv,dq")
ap.add_argument("--split", type=int, default=-1, help="forward split_mode: -1 automatic, 0 never")
ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const"], help="operand values: zeros / one constant toggle almost no datapath bits -> the time of the "
                "instruction stream without the power throttle that random data brings (DESIGN section 4.0)")
a = ap.parse_args()
B, H, S = a.B, a.H, a.S
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
if a.data == "zeros":
    qkv.zero_()
elif a.data == "const":
    qkv.fill_(0.125)
q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
v = qkv[:, :, 2].permute(0, 2, 1, 3)           # token-major view, as in the model
do = torch.randn(B, S, H * 64, generator=g, device="cuda").to(torch.bfloat16)
if a.data != "randn":
    do.fill_(0.0 if a.data == "zeros" else 0.125)
dov = do.view(B, S, H, 64).permute(0, 2, 1, 3)
dq, dk = torch.empty_like(q), torch.empty_like(k)
dv = torch.empty(B, S, H, 64, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
o, lse = ops.attention_fwd_raw(q, k, v)
ov = o.view(B, S, H, 64).permute(0, 2, 1, 3)
ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv)
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
for _ in range(a.iters):
    if "fwd" in a.which:
        ops.attention_fwd_raw(q, k, v, split_mode=a.split)
    if "dkv" in a.which or "dq" in a.which:
        ops.attention_bwd_raw(q, k, v, ov, dov, lse, dq, dk, dv, split_mode=a.split)
torch.cuda.synchronize()
unit = 2.0 * S * S * 64 * B * H
hw_units = {"attn_fwd_kernel": 2, "attn_bwd_dkv_kernel": 4, "attn_bwd_dq_kernel": 3, "attn_bwd_fused_kernel": 5}
for name, s in ops.TIMER.summary().items():
    if name not in hw_units:          # HBM-bound helpers (attn_delta_kernel): bytes, not FLOPs
        print(f"{name:22s} avg {s['avg_ms']:8.3f} ms  {s['work_per_launch'] / s['avg_ms'] / 1e6:7.1f} GB/s")
        continue
    print(f"{name:22s} avg {s['avg_ms']:8.3f} ms  algorithmic {s['work_per_launch'] / s['avg_ms'] / 1e9:7.1f} TF/s  "
          f"executed-MFMA {hw_units[name] * unit / s['avg_ms'] / 1e9:7.1f} TF/s")
