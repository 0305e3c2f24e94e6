"""Correctness and timing of vgpa_gemm_bf16 (gemm_w1.hip) against hipBLASLt (torch F.linear) at the feed-forward shapes of cfg2.
   The kernel lives in variant builds only:   tools/build_variant.sh gemm   (and un-ignore var/ in .gpurunignore), then
   gpurun -- 'VGPA_LIB=$PWD/var/lib_gemm.so PYTHONPATH=. python tools/gemm_check.py --time'      (profiles/r03_gemm_probe.txt)"""
import sys, torch, torch.nn.functional as F
from videogpa_amd import _lib
dev = "cuda"


def run(x, w, bias, epi, aux=None, out=None):
    M, K = x.shape; N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if out is None else out
    _lib.call("vgpa_gemm_bf16", x, x.stride(0), w, w.stride(0), bias, out, out.stride(0), aux, 0 if aux is None else aux.stride(0),
                   M, N, K, epi, torch.cuda.current_stream().cuda_stream)
    return out


def check(M, N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g).bfloat16()
    ref = x.float() @ w.float().T + b.float()
    y0 = run(x, w, b, 0)
    e0 = (y0.float() - ref).abs().max().item() / ref.abs().max().item()
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    y1 = run(x, w, b, 1, aux=pre)
    r1 = F.gelu(ref.bfloat16().float(), approximate="tanh")
    e1 = (y1.float() - r1).abs().max().item() / r1.abs().max().item()
    ep = (pre.float() - ref).abs().max().item() / ref.abs().max().item()
    y1b = run(x, w, b, 1)
    e1b = (y1b.float() - F.gelu(ref, approximate="tanh")).abs().max().item() / r1.abs().max().item()
    u = torch.randn(M, N, device=dev, generator=g).bfloat16()
    y2 = run(x, w, None, 2, aux=u)
    uf = u.float().requires_grad_(True)
    F.gelu(uf, approximate="tanh").backward(x.float() @ w.float().T)
    e2 = (y2.float() - uf.grad).abs().max().item() / uf.grad.abs().max().item()
    print(f"M={M} N={N} K={K}: id {e0:.2e}  gelu+pre {e1:.2e} (pre {ep:.2e})  gelu {e1b:.2e}  dgelu {e2:.2e}")
    assert max(e0, e1, ep, e1b, e2) < 1e-2


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if __name__ == "__main__":
    for shp in [(256, 128, 64), (256, 128, 192), (512, 256, 256), (1000, 384, 448), (3000, 1024, 3072)]:
        check(*shp)
    if "--time" in sys.argv:
        M = 35552
        for (N, K) in [(12288, 3072), (3072, 12288), (3072, 3072), (9216, 3264)]:
            x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); pre = torch.empty_like(out)
            fl = 2.0 * M * N * K
            t_ref = timeit(lambda: F.linear(x, w, b))
            t0 = timeit(lambda: run(x, w, b, 0, out=out))
            t1 = timeit(lambda: run(x, w, b, 1, aux=pre, out=out))
            t2 = timeit(lambda: run(x, w, None, 2, aux=pre, out=out))
            print(f"N={N} K={K}: hipBLASLt {t_ref:.3f} ms ({fl / t_ref / 1e9:.0f} TF)   w1 id {t0:.3f} ms ({fl / t0 / 1e9:.0f} TF)  gelu+pre {t1:.3f}  dgelu {t2:.3f}")
