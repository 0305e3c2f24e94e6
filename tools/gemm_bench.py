"""hipBLASLt timings of the dense projections at the headline shape (M = 2 x 17776 tokens): forward x W^T ("TN"), backward
dX = dY W as torch.autograd issues it ("NN"), and the same product through a cached transposed weight (TN again)."""
import torch
import torch.nn.functional as F

M = 2 * 17776
shapes = {"qkv": (9216, 3072), "to_out": (3072, 3072), "ff1": (12288, 3072), "ff2": (3072, 12288)}   # (out, in)


def t(f, n=10):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    Wt = W.t().contiguous()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    fl = 2.0 * M * N * K / 1e9
    a, b, c = t(lambda: F.linear(x, W)), t(lambda: dy @ W), t(lambda: F.linear(dy, Wt))
    print(f"{name:7s} fwd x W^T {a:6.3f} ms {fl / a:7.1f} TF/s | dX = dY W (NN) {b:6.3f} ms {fl / b:7.1f} TF/s | dX via cached W^T (TN) {c:6.3f} ms {fl / c:7.1f} TF/s")
