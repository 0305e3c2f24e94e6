import torch, torch.nn.functional as F
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for M in (35552, 36960):
    for (N, K) in [(9216, 3264), (3072, 3136), (12288, 3072), (3072, 12288), (3072, 3136), (14336, 3072)]:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        xa, xb = x[: M // 2], x[M // 2:]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t1 = timeit(lambda: F.linear(x, w, b))
        t2 = timeit(lambda: (F.linear(xa, w, b), F.linear(xb, w, b)))
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: one call {t1:.3f} ms ({fl/t1/1e9:.0f} TF)   two halves {t2:.3f} ms ({fl/t2/1e9:.0f} TF)")
