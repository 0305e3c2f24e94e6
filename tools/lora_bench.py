"""LoRA kernel timings at the cfg2 (M = 35 552) and cfg5 (M = 36 960) row counts: down-projection r = 64 / r = 192, the two gradient products."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

D = 3072


def t(f, n=30):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M in (35552, 36960):
    x = torch.randn(M, D, device="cuda").bfloat16()
    a1 = torch.randn(64, D, device="cuda").bfloat16()
    a3 = torch.randn(192, D, device="cuda").bfloat16()
    t1 = torch.randn(M, 64, device="cuda").bfloat16()
    gb = M * D * 2 / 1e9
    for name, f in (("lora_down r=64", lambda: ops.lora_down(x, a1)), ("lora_down r=192", lambda: ops.lora_down(x, a3)),
                    ("lora_grad dA [64 x 3072]", lambda: ops.lora_grad(t1, x)), ("lora_grad dB [3072 x 64]", lambda: ops.lora_grad(x, t1))):
        us = t(f)
        print(f"M={M} {name:28s} {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s", flush=True)
