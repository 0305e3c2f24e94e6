"""Cross-attention shape of the Wan2.2 step (Sq = 18480 video tokens, Skv = 512 text tokens, B*H = 48) through ops.attention128, forward + backward,
timed per entry; run once per sweep-length setting, which is a compile-time constant of a variant build since round 6 (the library reads no environment):
    tools/build_variant.sh sweep512 -DATTN128_W1_MIN_KEYS=512 && VGPA_LIB=$PWD/var/lib_sweep512.so PYTHONPATH=. python tools/attn128_cross_time.py"""
import os

import torch

from videogpa_amd import ops

B, H, Sq, Skv, D = 2, 24, 18480, 512, 128
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B, H, Sq, D, device="cuda", generator=g).bfloat16().requires_grad_(True)
k = torch.randn(B, H, Skv, D, device="cuda", generator=g).bfloat16().requires_grad_(True)
v = torch.randn(B, H, Skv, D, device="cuda", generator=g).bfloat16().requires_grad_(True)
do = torch.randn(B, H, Sq, D, device="cuda", generator=g).bfloat16()
for _ in range(2):
    ops.attention128(q, k, v).backward(do)
# reference values from the default dispatch are compared by the caller through the checksum below
ops.TIMER = ops.KernelTimer()
for _ in range(5):
    q.grad = k.grad = v.grad = None
    o = ops.attention128(q, k, v)
    o.backward(do)
torch.cuda.synchronize()
for name, s in ops.TIMER.summary().items():
    print(f"lib={os.path.basename(os.environ.get('VGPA_LIB', 'product')):16s} {name:28s} {s['avg_ms']:7.3f} ms")
print("checksums", float(o.float().abs().sum()), float(q.grad.float().abs().sum()), float(k.grad.float().abs().sum()), float(v.grad.float().abs().sum()))
