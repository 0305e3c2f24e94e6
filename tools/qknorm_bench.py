"""Stand-alone timing of the QK-norm + RoPE kernels at the cfg2 shape (2 x 17 776 tokens x 48 heads x 64; 874 MB moved forward, 1.3 GB backward: larger than
the infinity cache, so a loop over one operand set is cache-cold)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

B, S, H, D = 2, 17776, 48, 3072
qkv = [torch.randn(B, S, 3 * D, device="cuda").bfloat16().requires_grad_(True) for _ in range(2)]
wq, bq, wk, bk = (torch.randn(64, device="cuda") for _ in range(4))
do = torch.randn(B, S, D, device="cuda").bfloat16()


def t(f, n=16):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ops.TIMER = ops.KernelTimer()
for i in range(6):
    x = qkv[i % 2]
    x.grad = None
    ops.qknorm_attention(x, wq, bq, wk, bk, H).backward(do)
torch.cuda.synchronize()
for name, s in ops.TIMER.summary().items():
    if "qknorm" in name:
        print(f"{name:24s} {s['avg_ms'] * 1e3:8.1f} us  {s['work_per_launch'] / s['avg_ms'] / 1e9:6.2f} TB/s")
