"""hipBLASLt vs rocBLAS (torch.backends.cuda.preferred_blas_library) for the dense projections at M = 35 552, interleaved."""
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


shapes = (("qkv+ext", 9216, 3264), ("qkv", 9216, 3072), ("out+ext", 3072, 3136), ("ff1", 12288, 3072), ("ff2", 3072, 12288), ("qkv dX+ext", 3072, 9408))
ops = {n: ((torch.randn(M, K, device="cuda").bfloat16()), (0.02 * torch.randn(N, K, device="cuda")).bfloat16()) for n, N, K in shapes}
res = {}
for rep in range(3):
    for lib in ("cublaslt", "cublas"):
        torch.backends.cuda.preferred_blas_library(lib)
        for n, N, K in shapes:
            x, W = ops[n]
            res.setdefault((n, lib), []).append(t(lambda: F.linear(x, W)))
for n, N, K in shapes:
    a, b = min(res[(n, "cublaslt")]), min(res[(n, "cublas")])
    print(f"{n:12s} N={N:5d} K={K:5d}  hipBLASLt {a:6.3f} ms {2.0 * M * N * K / a / 1e9:7.1f} TF/s   rocBLAS {b:6.3f} ms {2.0 * M * N * K / b / 1e9:7.1f} TF/s")
