"""the vendor e4m3 GEMM (torch._scaled_mm, per-row scales) against the row count around cfg5's 36 960: would row padding pay there too?"""
import time
import torch
for (N, K) in [(14336, 3072), (3072, 14336)]:
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.float8_e4m3fn)
    sw = torch.ones(1, N, device="cuda")
    b = torch.randn(N, device="cuda").bfloat16()
    for M in (36960, 37120, 37376, 37888, 38912, 40960, 18480, 18688, 19456, 20480):
        x = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
        sx = torch.ones(M, 1, device="cuda")
        fn = lambda: torch._scaled_mm(x, w.t(), scale_a=sx, scale_b=sw, bias=b, out_dtype=torch.bfloat16)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        base = 36960 if M > 30000 else 18480
        print(f"N={N:5d} K={K:5d} M={M:5d}: {ms:6.3f} ms  {2.0 * base * N * K / ms / 1e9:7.1f} useful TF/s for {base} rows")
