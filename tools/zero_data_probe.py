"""Random operands against all-zero operands for the MFMA-bound launches of the step: the same instruction streams, with and without the
datapath toggling that drives the part into its 1400 W cap.  hipBLASLt's bf16 GEMM at the cfg2 feed-forward shape and the
head_dim-128 attention entry points at the cfg5 pair-batch shape; tools/attn_bench.py --data does the head_dim-64 kernels.
    gpurun -- 'PYTHONPATH=. python tools/zero_data_probe.py'"""
import torch
import torch.nn.functional as F

from videogpa_amd import ops


def timeit(fn, n=6):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


g = torch.Generator(device="cuda").manual_seed(0)
M, K, N = 35552, 3072, 12288
for data in ("randn", "zeros"):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    if data == "zeros":
        x.zero_(); W.zero_()
    t = timeit(lambda: F.linear(x, W))
    print(f"hipBLASLt bf16 GEMM {M}x{N}x{K}  {data:6s} {t:7.3f} ms  {2.0 * M * N * K / t / 1e9:7.0f} TFLOP/s")
    del x, W
B, H, S, D = 2, 24, 18480, 128
for data in ("randn", "zeros"):
    q, k, v, do = (torch.randn(B, H, S, D, device="cuda", generator=g).bfloat16() for _ in range(4))
    if data == "zeros":
        for t_ in (q, k, v, do):
            t_.zero_()
    qg, kg, vg = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
    ops.TIMER = ops.KernelTimer()
    for _ in range(5):
        qg.grad = kg.grad = vg.grad = None
        ops.attention128(qg, kg, vg).backward(do)
    torch.cuda.synchronize()
    unit = 2.0 * B * H * S * S * D
    for name, s in ops.TIMER.summary().items():
        prods = 2 if "fwd" in name else 7
        print(f"{name:14s} {data:6s} {s['avg_ms']:7.3f} ms  executed MFMA {prods * unit / s['avg_ms'] / 1e9:7.0f} TFLOP/s")
    ops.TIMER = None
