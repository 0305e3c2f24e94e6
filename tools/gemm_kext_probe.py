"""hipBLASLt cost of letting the LoRA adapters ride the dense projection as extra K: forward
[M,3072+192] x [9216,3264]^T vs K = 3072; backward dX [M,9216+192] x [3072,9408]^T vs K = 9216; to_out likewise (+64)."""
import torch
import torch.nn.functional as F

M = 2 * 17776


def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, N, K, ext in (("qkv fwd", 9216, 3072, 192), ("qkv dX", 3072, 9216, 192), ("out fwd", 3072, 3072, 64), ("out dX", 3072, 3072, 64),
                        ("ff1 fwd", 12288, 3072, 0), ("ff2 fwd", 3072, 12288, 0)):
    for k in sorted({K, K + ext, K + 256 if ext else K}):
        x = torch.randn(M, k, device="cuda").bfloat16()
        W = (0.02 * torch.randn(N, k, device="cuda")).bfloat16()
        ms = t(lambda: F.linear(x, W))
        print(f"{name:8s} N={N:5d} K={k:5d}: {ms:6.3f} ms  {2.0 * M * N * k / ms / 1e9:7.1f} TF/s  ({2.0 * M * N * K / ms / 1e9:7.1f} TF/s counting K={K} only)")
    # strided-row input (a [M, K] view of a wider buffer), as the K-extension layout produces for the plain consumers
    if ext:
        big = torch.randn(M, K + ext, device="cuda").bfloat16()
        W = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
        ms = t(lambda: F.linear(big[:, :K], W))
        print(f"{name:8s} N={N:5d} K={K:5d} on a row-strided view (lda={K + ext}): {ms:6.3f} ms")
