"""hipBLASLt bf16 GEMM time and joules per launch against the row count M around the cfg2 value (35 552 = 2 x 17 776, not a multiple of the 256-row macro tile):
would padding the token dimension pay?   python tools/gemm_m_probe.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy import read_joules  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--base", type=int, default=35552, help="the real row count")
ap.add_argument("--ms", default="35552,35840,36096,36352,36608,36864,37120", help="row counts to run the GEMM over")
ap.add_argument("--shapes", default="12288x3072,3072x12288,9216x3264,3072x3136,3072x9408")
a = ap.parse_args()
BASE = a.base
for (N, K) in [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",")]:
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for M in [int(v) for v in a.ms.split(",")]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        fn = lambda: F.linear(x, w, b)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        n = max(8, int(1.0 / max(time.perf_counter() - t0, 1e-4)))
        e0, t0 = read_joules(), time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        t1, e1 = time.perf_counter(), read_joules()
        ms, J = (t1 - t0) / n * 1e3, (e1 - e0) / n
        fl = 2.0 * M * N * K
        print(f"N={N:5d} K={K:5d} M={M:5d}: {ms:6.3f} ms  ({2.0 * BASE * N * K / ms / 1e9:7.1f} useful TF/s for {BASE} rows)  {fl / ms / 1e9:7.1f} TF/s  {fl / J / 1e12:5.3f} TFLOP/J")
