"""Does the row stride of the activation operand matter to hipBLASLt at M = 35 552?  x[M,K] contiguous vs x as the head of a
wider [M, K+pad] buffer, interleaved repeats (box clocks drift)."""
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


for name, N, K in (("ff1", 12288, 3072), ("qkv", 9216, 3072), ("out", 3072, 3072), ("ff2", 3072, 12288)):
    W = (0.02 * torch.randn(N, K, device="cuda")).bfloat16()
    xs = {0: torch.randn(M, K, device="cuda").bfloat16()}
    for pad in (64, 192, 256):
        xs[pad] = torch.randn(M, K + pad, device="cuda").bfloat16()[:, :K]
    res = {p: [] for p in xs}
    for rep in range(4):
        for p, x in xs.items():
            res[p].append(t(lambda: F.linear(x, W)))
    print(name, f"N={N} K={K}", "  ".join(f"lda=K+{p}: {min(v):.3f} ms (med {sorted(v)[len(v)//2]:.3f})" for p, v in res.items()), flush=True)
    # output row stride: y as the head of a wider buffer is not expressible through F.linear (it allocates); skipped
