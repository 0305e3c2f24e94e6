"""Is an OCP-e4m3 fp8 GEMM available through the vendor library on this box, how fast is it at the Wan2.2 feed-forward shapes, and what does
per-row / per-tensor dynamic scaling cost in accuracy?   gpurun -- 'PYTHONPATH=. python tools/fp8_probe.py'   (profiles/r03_fp8_probe.txt)"""
import torch, torch.nn.functional as F
dev = "cuda"


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print("torch", torch.__version__, "device", torch.cuda.get_device_name(0))
for name in ("float8_e4m3fn", "float8_e4m3fnuz", "float8_e5m2"):
    print(name, hasattr(torch, name))
M = 18480
for (N, K) in [(14336, 3072), (3072, 14336), (3072, 3072)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    ref = x.float() @ w.float().T
    t_bf = timeit(lambda: F.linear(x, w))
    fl = 2.0 * M * N * K
    line = f"M={M} N={N} K={K}: bf16 {t_bf:.3f} ms ({fl / t_bf / 1e9:.0f} TF)"
    for dt in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
        try:
            fmax = torch.finfo(dt).max
            # per-tensor scales
            sx = x.float().abs().max() / fmax; sw = w.float().abs().max() / fmax
            xq = (x.float() / sx).to(dt); wq = (w.float() / sw).to(dt)
            f = lambda: torch._scaled_mm(xq, wq.t(), scale_a=sx.reshape(()).float(), scale_b=sw.reshape(()).float(), out_dtype=torch.bfloat16)
            y = f()
            err = (y.float() - ref).norm() / ref.norm()
            t = timeit(f)
            line += f" | {str(dt)[6:]} tensor-scale {t:.3f} ms ({fl / t / 1e9:.0f} TF) rel-err {err:.3e}"
            # per-row scales
            try:
                sxr = (x.float().abs().amax(dim=1, keepdim=True) / fmax).clamp_min(1e-12); swr = (w.float().abs().amax(dim=1, keepdim=True) / fmax).clamp_min(1e-12)
                xqr = (x.float() / sxr).to(dt); wqr = (w.float() / swr).to(dt)
                fr = lambda: torch._scaled_mm(xqr, wqr.t(), scale_a=sxr.float(), scale_b=swr.t().contiguous().float(), out_dtype=torch.bfloat16)
                yr = fr()
                errr = (yr.float() - ref).norm() / ref.norm()
                tr = timeit(fr)
                line += f" ; row-scale {tr:.3f} ms ({fl / tr / 1e9:.0f} TF) rel-err {errr:.3e}"
            except Exception as e:
                line += f" ; row-scale unavailable ({type(e).__name__}: {str(e)[:80]})"
        except Exception as e:
            line += f" | {str(dt)[6:]} unavailable ({type(e).__name__}: {str(e)[:100]})"
    yb = F.linear(x, w)
    line += f" | bf16 rel-err {((yb.float() - ref).norm() / ref.norm()).item():.3e}"
    print(line)
