"""LoRA kernel timings at the cfg2 (M = 35 552) and cfg5 (M = 36 960) row counts: down-projection r = 64 / r = 192, the two gradient products.
Every call takes the NEXT of six different [M, 3072] operands (1.3 GB in rotation): a loop over one 218 MB tensor is served by the 256 MB infinity cache
and reports 5 TB/s where the step, whose operand was written a layer ago, sees 3.4."""
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
    xs = [torch.randn(M, D, device="cuda").bfloat16() for _ in range(6)]
    a1 = torch.randn(64, D, device="cuda").bfloat16()
    a3 = torch.randn(192, D, device="cuda").bfloat16()
    t1 = torch.randn(M, 64, device="cuda").bfloat16()
    gb = M * D * 2 / 1e9
    turn = [0]

    def x():
        turn[0] += 1
        return xs[turn[0] % len(xs)]
    for name, f in (("lora_down r=64", lambda: ops.lora_down(x(), a1)), ("lora_down r=192", lambda: ops.lora_down(x(), a3)),
                    ("lora_grad dA [64 x 3072]", lambda: ops.lora_grad(t1, x())), ("lora_grad dB [3072 x 64]", lambda: ops.lora_grad(x(), t1))):
        us = t(f)
        print(f"M={M} {name:28s} {us:8.1f} us  {gb / us * 1e3:6.2f} TB/s", flush=True)
