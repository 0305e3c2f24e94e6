"""Timing of the head_dim-128 attention entry points at the Wan2.2 self-attention shape (B*H = 24, S = 18480), per kernel family:
forward (w1 / compiler-scheduled), backward with the dK/dV kernel in either form.   gpurun -- 'PYTHONPATH=. python tools/attn128_time.py'"""
import argparse

import torch
from videogpa_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1, help="2 = win and lose as one batch, as the cfg5 step runs them")
ap.add_argument("--product-only", action="store_true", help="only the kernels the cfg5 step launches (for rocprofv3 --pmc passes)")
ap.add_argument("--iters", type=int, default=5)
a_ = ap.parse_args()
B, H, S, D = a_.B, 24, 18480, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", generator=g).bfloat16() for _ in range(4))
st = lambda t: ops._bhs_strides(t)
stream = torch.cuda.current_stream().cuda_stream
o = torch.empty_like(q); lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
wsf = torch.empty(_lib.query("vgpa_attn128_fwd_workspace_bytes", B, H, S), dtype=torch.uint8, device="cuda")
wsb = torch.empty(_lib.query("vgpa_attn128_bwd_workspace_bytes", B, H, S), dtype=torch.uint8, device="cuda")
scale = D ** -0.5


def fwd(w1):
    _lib.call("vgpa_attn128_fwd", q, k, v, o, lse, st(q), st(k), st(v), st(o), None, None, B, H, S, S, scale, wsf if w1 else None, wsf.numel() if w1 else 0, stream)


wsf8 = torch.empty(_lib.query("vgpa_attn128_fwd_f8_workspace_bytes", B, H, S, S), dtype=torch.uint8, device="cuda")


def fwd_f8():
    _lib.call("vgpa_attn128_fwd_f8", q, k, v, o, lse, st(q), st(k), st(v), st(o), None, None, None, None, None, None, None, None, B, H, S, S, scale, wsf8, wsf8.numel(), stream)


def bwd(mode):
    _lib.call("vgpa_attn128_bwd", q, k, v, o, do, lse, dq, dk, dv, st(q), st(k), st(v), st(o), st(do), st(dq), st(dk), st(dv), None, None, B, H, S, S, scale, mode, wsb, wsb.numel(), stream)


def timeit(fn, n=a_.iters):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ff = 4.0 * B * H * S * S * D
cases = [("fwd w1", lambda: fwd(True), ff), ("fwd simple", lambda: fwd(False), ff), ("bwd (w1 dq + w1 dkv)", lambda: bwd(1), 2.0 * ff),
         ("bwd (compiler-scheduled)", lambda: bwd(0), 2.0 * ff)]
cases.insert(1, ("fwd e4m3 (amax + quantise + MFMA f8f6f4)", fwd_f8, ff))
if a_.product_only:
    cases = [cases[0], cases[1], cases[3]]
for name, fn, fl in cases:
    t = timeit(fn)
    print(f"{name:32s} {t:8.3f} ms   {fl / t / 1e9:7.0f} TFLOP/s algorithmic")
