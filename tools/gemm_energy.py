"""Joules per launch of the hand-written GEMM probe (vgpa_gemm_bf16, variant builds) against hipBLASLt at the FF1 shape of cfg2: is the probe short of the
vendor's kernel because it needs more joules per FLOP (operand feed) or because it idles (per-tile prologue / epilogue bubbles, below the power cap)?
   tools/build_variant.sh gemm && VGPA_LIB=$PWD/var/lib_gemm.so PYTHONPATH=. python tools/gemm_energy.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy import read_joules  # noqa: E402
from videogpa_amd import _lib  # noqa: E402

dev = "cuda"
M = 35552
for (N, K) in [(12288, 3072), (3072, 12288)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    pre = torch.empty_like(out)
    st = torch.cuda.current_stream().cuda_stream

    def w1(epi, aux=None, bias=b):
        _lib.call("vgpa_gemm_bf16", x, x.stride(0), w, w.stride(0), bias, out, out.stride(0), aux, 0 if aux is None else aux.stride(0), M, N, K, epi, st)
    cases = {"hipBLASLt (F.linear)": lambda: F.linear(x, w, b), "w1 gemm identity": lambda: w1(0), "w1 gemm gelu + pre-activation": lambda: w1(1, pre),
             "w1 gemm dgelu": lambda: w1(2, pre, None)}
    fl = 2.0 * M * N * K
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        n = max(8, int(1.5 / max(time.perf_counter() - t0, 1e-4)))
        e0, t0 = read_joules(), time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        t1, e1 = time.perf_counter(), read_joules()
        ms, J = (t1 - t0) / n * 1e3, (e1 - e0) / n
        print(f"N={N:5d} K={K:5d} {name:32s} {ms:7.3f} ms  {J:6.3f} J  {J / ms * 1e3:7.1f} W  {fl / ms / 1e9:7.1f} TF/s  {fl / J / 1e12:6.3f} TFLOP/J")
