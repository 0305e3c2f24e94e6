"""Micro-benchmark of the HBM-bound transformer kernels at the headline shape (2 sequences x 17776 tokens x 3072)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

B, S, D, Lt = 2, 17776, 3072, 226
x = torch.randn(B, S, D, device="cuda").bfloat16()
y = torch.randn(B, S, D, device="cuda").bfloat16()
g = torch.randn(B, 2, D, device="cuda")
mod = torch.randn(B, 4, D, device="cuda")
w, bb = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
u = torch.randn(B, S, 4 * D, device="cuda").bfloat16()
unit = B * S * D * 2 / 1e9   # GB per pass over one [B,S,D] bf16 tensor


def t(f, n=20):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def show(name, us, passes):
    print(f"{name:34s} {us:8.1f} us   {passes * unit / us * 1e6 / 1e3:6.2f} TB/s ({passes} passes)")


show("gate_residual fwd", t(lambda: ops.gate_residual(x, y, g, Lt)), 3)
show("ln_modulate v1 fwd", t(lambda: ops.ln_modulate_v1(x, w, bb, mod, Lt, 1e-5)), 2)
show("ln_modulate (fused kernel, no y)", t(lambda: ops.ln_modulate(x, w, bb, mod, Lt, 1e-5)), 2)
show("residual_ln fwd", t(lambda: ops.residual_ln(x, y, g, w, bb, mod, Lt, 1e-5)), 4)
xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
xn, n = ops.residual_ln(xr, yr, g, w, bb, mod, Lt, 1e-5)
show("residual_ln bwd", t(lambda: torch.autograd.grad([xn, n], [xr, yr], [x, y], retain_graph=True)), 5)
xr2 = x.clone().requires_grad_(True)
n1 = ops.ln_modulate_v1(xr2, w, bb, mod, Lt, 1e-5)
show("ln_modulate v1 bwd", t(lambda: torch.autograd.grad(n1, xr2, y, retain_graph=True)), 3)
show("gelu fwd", t(lambda: ops.gelu_tanh(u)), 8)
ur = u.clone().requires_grad_(True)
gu = ops.gelu_tanh(ur)
show("gelu bwd", t(lambda: torch.autograd.grad(gu, ur, u, retain_graph=True)), 12)

# LoRA kernels at the shapes the model uses (rank 64; fused q,k,v = three adapters side by side)
M = B * S
x2 = x.view(M, D)
a1 = torch.randn(64, D, device="cuda").bfloat16()
a3 = torch.randn(192, D, device="cuda").bfloat16()
b1 = (0.01 * torch.randn(D, 64, device="cuda")).bfloat16()
t1 = torch.randn(M, 64, device="cuda").bfloat16()
t3 = torch.randn(M, 192, device="cuda").bfloat16()
y1 = torch.randn(M, D, device="cuda").bfloat16()
qkv = torch.randn(M, 3 * D, device="cuda").bfloat16()
show("lora_down r=64", t(lambda: ops.lora_down(x2, a1)), 1)
show("lora_down r=192 (q,k,v)", t(lambda: ops.lora_down(x2, a3)), 1)
show("lora_up_add [M,3072] r=64 (RMW)", t(lambda: ops.lora_up_add(y1, t1, b1, 2.0)), 2)
show("lora_up_add column slice of [M,9216]", t(lambda: ops.lora_up_add(qkv[:, D:2 * D], t3[:, 64:128], b1, 2.0)), 2)
show("lora_grad dA = t^T x  [64 x 3072]", t(lambda: ops.lora_grad(t1, x2)), 1)
show("lora_grad dB = dy^T t [3072 x 64]", t(lambda: ops.lora_grad(y1, t1)), 1)
